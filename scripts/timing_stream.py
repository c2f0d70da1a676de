#!/usr/bin/env python3
"""Where a workgroup of the streaming update kernel (kh_tile64s.h) spends an interval: exchange | wait for the operator
tiles | products.  Needs gpurun_out/libkrotov_hip_timing.so built with -DKH_TIMING on the box:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DKH_TIMING -Iinclude krotov_amd/csrc/krotov_hip.hip -o gpurun_out/libkrotov_hip_timing.so
usage: python scripts/timing_stream.py [K] [L]   (KH_STREAM_TWO / KH_STREAM_G / KH_STREAM_VAR as in exp_stream.sh)"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from krotov_amd import _lib
_lib.LIB_PATH = os.environ.get('KH_TIMING_LIB', os.path.join(ROOT, 'gpurun_out', 'libkrotov_hip_timing.so'))
from krotov_amd import configs
from krotov_amd.engine import HipKrotovEngine
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N, nt = 64, 1001
spec = configs.config_c5(K=K, N=N, nt=nt, L=L, distinct='--distinct' in sys.argv)
ops = [[spec.H0[k]] + list(spec.Hc[k]) for k in range(K)]
eng = HipKrotovEngine(ops, np.diff(spec.tlist))
eng.profile = True
tl = spec.tlist
pulses = np.array([[0.5 * np.sin((l + 1) * np.pi * (t + 0.5 * (tl[1] - tl[0])) / tl[-1]) for t in tl[:-1]] for l in range(L)])
S = np.ones((L, nt - 1))
lam = np.full(L, 1e3)
chi_T = spec.target / np.linalg.norm(spec.target, axis=1)[:, None]
chi = eng.backward(chi_T, pulses)
norms = np.full(K, 1.0 / (2 * K))
for _ in range(2):
    eng.forward_update(chi, norms, spec.init, pulses, S, lam)
buf = (ctypes.c_double * 4)()
_lib.check(eng._lib.kh_last_stats(eng._handle, buf))
ms = min(eng.kernel_times_ms()['update'])
n = nt - 1
print('%s K %d L %d  update %.2f ms (%.2f us/interval); workgroup 0, us per interval: publish + gather %.2f  scalars, barrier, pulse update, tiles %.2f  products %.2f' % (
    eng.kernel, K, L, ms, ms * 1e3 / n, buf[1] / n / 100, buf[2] / n / 100, buf[3] / n / 100))
