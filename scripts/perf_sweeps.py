#!/usr/bin/env python3
"""Time the backward and update sweeps at engine level (dev tool, GPU only).
usage: python scripts/perf_sweeps.py [K] [N] [nt] [L] [distinct] ; KH_KERNEL selects the family."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from krotov_amd import _lib
if os.environ.get('KH_LIB'):  # (an experiment build of the library)
    _lib.LIB_PATH = os.path.abspath(os.environ['KH_LIB'])
from krotov_amd import configs
from krotov_amd.engine import HipKrotovEngine

K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 4001
L = int(sys.argv[4]) if len(sys.argv) > 4 else 1
distinct = len(sys.argv) > 5 and sys.argv[5] == 'distinct'  # every objective its own random drift
reps = 3
spec = configs.config_c5(K=K, N=N, nt=nt, L=L, distinct=distinct)
ops = [[spec.H0[k]] + [spec.Hc[k][l] for l in range(L)] for k in range(K)]
eng = HipKrotovEngine(ops, np.diff(spec.tlist))
eng.profile = True
tl = spec.tlist
pulses = np.array([[0.5 * np.sin((l + 1) * np.pi * (t + 0.5 * (tl[1] - tl[0])) / tl[-1]) for t in tl[:-1]] for l in range(L)])
S = np.ones((L, nt - 1))
lam = np.full(L, 50.0)
chi_T = spec.target / np.linalg.norm(spec.target, axis=1)[:, None]
norms = np.full(K, 1.0 / (2 * K))
chi = None
for _ in range(reps):
    chi = eng.backward(chi_T, pulses, out=chi)
    out = eng.forward_update(chi, norms, spec.init, pulses, S, lam)
eng.check()
t = eng.kernel_times_ms()
f_prop = 8.0 * N * N * 14
bw, up = min(t['backward']), min(t['update'])
print('%s%s K=%d N=%d nt=%d L=%d  backward %.2f ms (%.2f us/step, %.1f TF)  update %.2f ms (%.2f us/step, %.1f TF)' % (
    eng.kernel, ' distinct' if distinct else '', K, N, nt, L, bw, bw * 1e3 / (nt - 1), K * (nt - 1) * f_prop / bw / 1e9,
    up, up * 1e3 / (nt - 1), K * (nt - 1) * (f_prop + L * (8.0 * N * N + 8 * N)) / up / 1e9))
print('  matvecs/step/objective:', eng.stats()['matvecs'] / (K * (nt - 1)))
