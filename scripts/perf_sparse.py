#!/usr/bin/env python3
"""Sparse (CSR) vs dense operators on a damped d-level ladder in Liouville space (N = d^2), engine level
(dev tool, GPU only).  usage: python scripts/perf_sparse.py [d] [nt] [K] [csr]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from krotov_amd import configs
from krotov_amd.engine import HipKrotovEngine

d = int(sys.argv[1]) if len(sys.argv) > 1 else 25
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 501
K = int(sys.argv[3]) if len(sys.argv) > 3 else 16
spec = configs.config_sparse_lindblad(d=d, nt=nt, K=K)
tl = spec.tlist
pulses = np.array([[spec.controls[0](t + 0.5 * (tl[1] - tl[0]), None) for t in tl[:-1]]])
S, lam = np.ones((1, nt - 1)), np.full(1, 2.0)
chi_T = spec.target / np.linalg.norm(spec.target, axis=1)[:, None]
norms = np.full(K, 1.0 / (2 * K))
results = {}
only_csr = len(sys.argv) > 4 and sys.argv[4] == 'csr'  # (profiling runs: no dense comparison)
cases = [('csr', configs.sparse_ops(spec))]
if not only_csr:
    cases.append(('dense', [[spec.H0[k], spec.Hc[k][0]] for k in range(K)]))
for label, ops in cases:
    eng = HipKrotovEngine(ops, np.diff(tl), is_super=True)
    eng.profile = True
    for _ in range(2):
        chi = eng.backward(chi_T, pulses)
        opt, psi, _ = eng.forward_update(chi, norms, spec.init, pulses, S, lam)
    eng.check()
    t = eng.kernel_times_ms()
    results[label] = (opt.cpu().numpy(), psi.cpu().numpy())
    nnz = ops[0][0].nnz if label == 'csr' else spec.N**2
    print('%-6s %-12s N=%d K=%d nt=%d  entries/row %.1f  backward %.1f ms  update %.1f ms  (%.1f terms/step)' % (
        label, eng.kernel, spec.N, K, nt, nnz / spec.N, min(t['backward']), min(t['update']),
        eng.stats()['matvecs'] / (K * (nt - 1))))
    eng.close()
if not only_csr:
    print('max |pulse difference| csr vs dense: %.2e' % np.abs(results['csr'][0] - results['dense'][0]).max())
