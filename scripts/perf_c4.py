#!/usr/bin/env python3
"""Time the sweeps of BASELINE config 4 (transmon Liouvillian, N = d^2, K = n_logical^2 density
matrices sharing one operator list) at engine level (dev tool, GPU only).
usage: python scripts/perf_c4.py [d] [nt] [n_logical]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from krotov_amd import _lib
if os.environ.get('KH_LIB'):  # (an experiment build of the library, e.g. -DKH_COOP_X_NOEXCH)
    _lib.LIB_PATH = os.path.abspath(os.environ['KH_LIB'])
from krotov_amd import configs
from krotov_amd.engine import HipKrotovEngine

d = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 1001
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 4
spec = configs.config_c4(d=d, nt=nt, n_logical=nl)
K, N, L = spec.K, spec.N, spec.L
ops = [[spec.H0[k]] + [spec.Hc[k][l] for l in range(L)] for k in range(K)]
eng = HipKrotovEngine(ops, np.diff(spec.tlist), is_super=True)
eng.profile = True
tl = spec.tlist
pulses = np.array([[spec.controls[l](t + 0.5 * (tl[1] - tl[0]), None) for t in tl[:-1]] for l in range(L)])
S = np.ones((L, nt - 1))
lam = np.full(L, 1.0)
chi_T = spec.target / np.linalg.norm(spec.target, axis=1)[:, None]
norms = np.full(K, 1.0 / (2 * K))
chi = None
for _ in range(2):
    chi = eng.backward(chi_T, pulses, out=chi)
    out = eng.forward_update(chi, norms, spec.init, pulses, S, lam)
eng.check()
t = eng.kernel_times_ms()
bw, up = min(t['backward']), min(t['update'])
mv = eng.stats()['matvecs'] / (K * (nt - 1))
print('%s K=%d N=%d nt=%d L=%d  backward %.1f ms  update %.1f ms  matvecs/step/objective (update sweep) %.1f' % (
    eng.kernel, K, N, nt, L, bw, up, mv))
print('  issued flops (update sweep): %.2f TFLOP -> %.2f TFLOP/s' % (
    mv * K * (nt - 1) * 8.0 * N * N / 1e12, mv * K * (nt - 1) * 8.0 * N * N / up / 1e9))
