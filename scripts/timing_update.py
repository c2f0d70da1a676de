#!/usr/bin/env python3
"""Per-interval breakdown of the update sweep (needs gpurun_out/libkrotov_hip_timing.so built with -DKH_TIMING; gpurun_out/ is not shipped, so build it on the box: see the command in DESIGN.md section 6)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from krotov_amd import _lib
_lib.LIB_PATH = os.environ.get('KH_TIMING_LIB', os.path.join(ROOT, 'gpurun_out', 'libkrotov_hip_timing.so'))
from krotov_amd import configs
from krotov_amd.engine import HipKrotovEngine
import torch
K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, nt = 64, 4001
spec = configs.config_c5(K=K, N=N, nt=nt)
ops = [[spec.H0[k], spec.Hc[k][0]] for k in range(K)]
eng = HipKrotovEngine(ops, np.diff(spec.tlist))
eng.profile = True
tl = spec.tlist
pulses = np.array([[0.5 * np.sin(np.pi * (t + 0.5 * (tl[1] - tl[0])) / tl[-1]) for t in tl[:-1]]])
chi_T = spec.target / np.linalg.norm(spec.target, axis=1)[:, None]
chi = eng.backward(chi_T, pulses)
norms = np.full(K, 1.0 / (2 * K))
for _ in range(2):
    out = eng.forward_update(chi, norms, spec.init, pulses, np.ones((1, nt - 1)), np.array([50.0]))
eng.check()
buf = (ctypes.c_double * 4)()
_lib.check(eng._lib.kh_last_stats(eng._handle, buf))
ms = min(eng.kernel_times_ms()['update'])
n = nt - 1
print('%s K=%d update %.2f ms (%.2f us/interval); ticks/interval: exchange %.0f  build+phases %.0f  partial %.0f' % (
    eng.kernel, K, ms, ms * 1e3 / n, buf[1] / n, buf[2] / n, buf[3] / n))
if buf[0] < 0:  # matrix-core kernel (kh_tile64mm.h): its own four stamps
    print('  mm kernel, cycles/interval: rebuild %.0f | tiles -> partial sum %.0f | tiles -> last product %.0f | exchange wait %.0f' % (
        -buf[0] / n, buf[1] / n, buf[2] / n, buf[3] / n))
