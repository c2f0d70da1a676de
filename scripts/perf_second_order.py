#!/usr/bin/env python3
"""Update sweep of the config-5 shape, first order vs second order (kh_set_second_order), engine level (dev tool, GPU).
usage: python scripts/perf_second_order.py [K] [N] [nt] [L]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from krotov_amd import configs
from krotov_amd.engine import HipKrotovEngine

K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 1001
L = int(sys.argv[4]) if len(sys.argv) > 4 else 1
spec = configs.config_c5(K=K, N=N, nt=nt, L=L)
ops = [[spec.H0[k]] + [spec.Hc[k][l] for l in range(L)] for k in range(K)]
eng = HipKrotovEngine(ops, np.diff(spec.tlist))
eng.profile = True
tl = spec.tlist
pulses = np.array([[0.5 * np.sin((l + 1) * np.pi * (t + 0.5 * (tl[1] - tl[0])) / tl[-1]) for t in tl[:-1]] for l in range(L)])
S, lam = np.ones((L, nt - 1)), np.full(L, 50.0)
chi_T = spec.target / np.linalg.norm(spec.target, axis=1)[:, None]
norms = np.full(K, 1.0 / (2 * K))
chi = eng.backward(chi_T, pulses)
_, prev = eng.forward(pulses, spec.init, store=True)
for order in (1, 2, 1, 2):
    if order == 2:
        store = torch.empty_like(prev)
        eng.set_second_order(prev, store, float(os.environ.get('SIGMA', '-1')) * np.ones(nt - 1))
    else:
        eng.set_second_order()
    eng.kernel_times_ms(reset=True)
    for _ in range(3):
        eng.forward_update(chi, norms, spec.init, pulses, S, lam)
    eng.check()
    t = min(eng.kernel_times_ms()['update'])
    print('%s K=%d N=%d nt=%d L=%d  order %d: update sweep %.2f ms (%.2f us per interval), %.1f products per step' % (
        eng.kernel, K, N, nt, L, order, t, t * 1e3 / (nt - 1), eng.stats()['matvecs'] / (K * (nt - 1))))
