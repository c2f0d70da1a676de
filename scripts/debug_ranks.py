#!/usr/bin/env python3
"""`world` ranks sharing the one GPU on a case of tests/test_hip_parity.py::_two_rank_spec, with timing and the
engine's log (dev tool).  usage: python scripts/debug_ranks.py <case> <world>"""
import logging
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
logging.basicConfig(level=logging.INFO)
import test_hip_parity as t

if __name__ == '__main__':  # (the ranks are spawned: they import this file again)
    case, world = sys.argv[1], int(sys.argv[2])
    t0 = time.time()
    out = t._run_ranks(world, case, env={'KH_COOP_XCD': os.environ.get('KH_COOP_XCD', '0')}, timeout=400)
    print('%s world %d: %.1f s; p2p per rank %r; kernels %r' % (case, world, time.time() - t0, [o[3] for o in out], sorted({o[4] for o in out})))
