#!/usr/bin/env python3
"""Experiment: the sparse kernels' series cap / sub-step bound (KH_ELL_CAP) on the three-states problem
(tests/golden/dump_3states.npz): terms per step, time per iteration, distance of tau to the tight-tolerance reference and
to the run with the smallest cap (theta <= 1 per sub-step: the most conservative series)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch

import krotov_amd
import test_hip_parity as t
from krotov_amd.engine import LAST_ENGINE

g, objs, opts = t._three_states_problem()
scale = float(os.environ.get('DT_SCALE', '1'))  # (a longer grid step: larger theta per step; the controls keep their samples)
tlist = g['tlist'] * scale
base = None
for cap in sys.argv[1:] or ['1', '2', '4', '5', '6']:
    os.environ['KH_ELL_CAP'] = cap
    t0 = time.time()
    res = krotov_amd.optimize_pulses(objs, opts, tlist, propagator=krotov_amd.propagators.DensityMatrixODEPropagator(),
                                     chi_constructor=krotov_amd.functionals.chis_re, iter_stop=3)
    torch.cuda.synchronize()
    dt = time.time() - t0
    eng = LAST_ENGINE()
    tau = np.array(res.tau_vals)
    pulses = np.array(res.optimized_controls)
    if base is None:
        base = (tau, pulses)
    print('cap %s  %s  %.2f s  terms/step %.1f  |tau - tight| %.2e  |tau - cap %s| %.2e  |pulses - cap %s| %.2e (max |pulse| %.2e)' % (
        cap, eng.kernel, dt, eng.stats()['matvecs'] / (3 * 1999), np.abs(tau[0] - g['tau_tight_it3']).max(),
        (sys.argv[1:] or ['1'])[0], np.abs(tau - base[0]).max(), (sys.argv[1:] or ['1'])[0], np.abs(pulses - base[1]).max(),
        np.abs(pulses).max()))
