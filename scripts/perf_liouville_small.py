import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch, time
from krotov_amd import configs
from krotov_amd.engine import HipKrotovEngine
for name, spec in (('c4_d5 N=25 K=4', configs.config_c4(d=5, nt=1001, n_logical=2)), ('c4_d8 N=64 K=16', configs.config_c4(d=8, nt=1001, n_logical=4)), ('c2l', configs.config_c2_liouville(nt=500))):
    K, L = spec.K, spec.L
    ops = [[spec.H0[k]] + [spec.Hc[k][l] for l in range(L)] for k in range(K)]
    eng = HipKrotovEngine(ops, np.diff(spec.tlist), is_super=True)
    tl = spec.tlist
    pulses = np.array([[spec.controls[l](t + 0.5 * (tl[1] - tl[0]), None) for t in tl[:-1]] for l in range(L)])
    chi_T = spec.target / np.linalg.norm(spec.target, axis=1)[:, None]
    eng.backward(chi_T, pulses); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): eng.backward(chi_T, pulses)
    torch.cuda.synchronize()
    print('%-18s %-14s backward %.2f ms  matvecs/step/objective %.2f' % (name, eng.kernel, (time.perf_counter() - t0) / 3 * 1e3, eng.stats()['matvecs'] / (K * (len(tl) - 1))))
