#!/usr/bin/env python3
"""Backward sweep only, config-5 shape (dev tool; used for timing experiments that may break the results)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
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
chi = None
for _ in range(3):
    chi = eng.backward(chi_T, pulses, out=chi)
ms = min(eng.kernel_times_ms()['backward'])
print('%s backward %.2f ms (%.2f us/step), %.1f products per step' % (eng.kernel, ms, ms * 1e3 / (nt - 1), eng.stats()['matvecs'] / (K * (nt - 1))))
