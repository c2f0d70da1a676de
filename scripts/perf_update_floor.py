#!/usr/bin/env python3
"""Update sweep of the config-5 shape with an experiment build of the library (KH_LIB=<path>): used with
-DKH_Q2_X_NOEXCH (the phases alone: every workgroup uses its own partial sum, nothing is published or polled) and
-DKH_Q2_X_NOPHASES (the exchange alone: no propagation) to split the per-interval time of kh_q2_forward_update into
its two serial parts.  Results of those builds are wrong by design: timing only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from krotov_amd import _lib
if os.environ.get('KH_LIB'):
    _lib.LIB_PATH = os.environ['KH_LIB']
from krotov_amd import configs
from krotov_amd.engine import HipKrotovEngine
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
for _ in range(3):
    out = eng.forward_update(chi, norms, spec.init, pulses, np.ones((1, nt - 1)), np.array([50.0]))
eng.check()
ms = min(eng.kernel_times_ms()['update'])
print('%s K=%d lib=%s: update sweep %.2f ms = %.2f us per interval' % (
    eng.kernel, K, os.path.basename(_lib.LIB_PATH), ms, ms * 1e3 / (nt - 1)))
