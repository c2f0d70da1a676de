#!/usr/bin/env python3
"""Per-step breakdown of the q2 backward sweep (needs gpurun_out/libkrotov_hip_timing.so built with -DKH_TIMING on the box)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from krotov_amd import _lib
_lib.LIB_PATH = os.environ.get('KH_TIMING_LIB', os.path.join(ROOT, 'gpurun_out', 'libkrotov_hip_timing.so'))
from krotov_amd import configs
from krotov_amd.engine import HipKrotovEngine
K, N, nt = 256, 64, 4001
spec = configs.config_c5(K=K, N=N, nt=nt)
ops = [[spec.H0[k], spec.Hc[k][0]] for k in range(K)]
eng = HipKrotovEngine(ops, np.diff(spec.tlist))
eng.profile = True
tl = spec.tlist
pulses = np.array([[0.5 * np.sin(np.pi * (t + 0.5 * (tl[1] - tl[0])) / tl[-1]) for t in tl[:-1]]])
chi_T = spec.target / np.linalg.norm(spec.target, axis=1)[:, None]
for _ in range(2):
    chi = eng.backward(chi_T, pulses)
buf = (ctypes.c_double * 4)()
_lib.check(eng._lib.kh_last_stats(eng._handle, buf))
ms = min(eng.kernel_times_ms()['backward'])
n = nt - 1
print('%s backward %.2f ms (%.2f us/step); ticks/step: scalars+degree+rebuild %.0f  phases %.0f' % (
    eng.kernel, ms, ms * 1e3 / n, buf[1] / n, buf[2] / n))
