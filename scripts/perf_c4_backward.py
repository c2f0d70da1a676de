"""Config 4, backward sweep only (timing experiments whose results are wrong must not reach the update sweep:
garbage pulses mean an unbounded number of sub-steps).  KH_LIB selects the library."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from krotov_amd import _lib
if os.environ.get('KH_LIB'):
    _lib.LIB_PATH = os.environ['KH_LIB']
from krotov_amd import configs
from krotov_amd.engine import HipKrotovEngine
spec = configs.config_c4(nt=1001)
K, N, L = spec.K, spec.N, spec.L
ops = [[spec.H0[k]] + [spec.Hc[k][l] for l in range(L)] for k in range(K)]
eng = HipKrotovEngine(ops, np.diff(spec.tlist), is_super=True, theta_max=float(os.environ.get('KH_THETA_MAX', '0')))
tl = spec.tlist
pulses = np.array([[spec.controls[l](t + 0.5 * (tl[1] - tl[0]), None) for t in tl[:-1]] for l in range(L)])
chi_T = spec.target / np.linalg.norm(spec.target, axis=1)[:, None]
chi = eng.backward(chi_T, pulses)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    eng.backward(chi_T, pulses)
torch.cuda.synchronize()
print('%s backward %.1f ms, %.1f rounds per interval' % (eng.kernel, (time.perf_counter() - t0) / 3 * 1e3,
                                                        eng.stats()['matvecs'] / (K * (len(tl) - 1))))
if os.environ.get('KH_WITH_UPDATE'):  # (only with builds whose pulses stay bounded)
    S = np.ones((L, len(tl) - 1)); lam = np.full(L, 1e3); norms = np.full(K, 1.0 / (2 * K))
    eng.forward_update(chi, norms, spec.init, pulses, S, lam); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.forward_update(chi, norms, spec.init, pulses, S, lam)
    torch.cuda.synchronize()
    print('update %.1f ms' % ((time.perf_counter() - t0) / 3 * 1e3))
