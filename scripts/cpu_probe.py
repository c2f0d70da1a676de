#!/usr/bin/env python3
"""What the host cores of the box do with one dense expm(A dt) @ state at N = 64 (dev tool behind the
`cpu_baseline` numbers of bench.py): CPU model, BLAS build and threads, SciPy vs the oracle's own Pade-13."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy, scipy.linalg as la
from oracle import krotov_oracle as ko

print('cpu:', [l.split(':')[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')][0],
      '| visible cores', len(os.sched_getaffinity(0)), '| scipy', scipy.__version__, 'numpy', np.__version__)
try:
    import threadpoolctl
    for i in threadpoolctl.threadpool_info():
        print(' blas:', i.get('internal_api'), i.get('version'), 'threads', i.get('num_threads'), 'arch', i.get('architecture'))
except Exception as e:
    threadpoolctl = None
    print(' threadpoolctl:', e)
rng = np.random.default_rng(0)
N = 64
G = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
H = (G + G.conj().T) / 2
H *= 0.5 / np.linalg.norm(H, 2)
A = -1j * H
v = rng.standard_normal(N) + 0j


def bench(tag):
    for name, f in (('scipy.linalg.expm', la.expm), ('oracle Pade-13', ko.expm_pade13)):
        f(A)
        t = time.perf_counter()
        for _ in range(200):
            f(A) @ v
        print('  %-28s %-18s %.3f ms per expm @ v' % (tag, name, (time.perf_counter() - t) / 200 * 1e3))


bench('default BLAS threads')
if threadpoolctl is not None:
    with threadpoolctl.threadpool_limits(limits=1, user_api='blas'):
        bench('1 BLAS thread')
t = time.perf_counter()
for _ in range(2000):
    A @ A
print('  64x64 complex GEMM: %.1f us' % ((time.perf_counter() - t) / 2000 * 1e6))

# the bench's own calibration: one Krotov iteration of a K=1 problem in reference-structured mode
from krotov_amd import configs
from oracle import cpu_baseline as cb
for nt in (41, 401):
    r = cb.timed_iteration(configs.config_c5(K=1, N=64, nt=nt, L=1), processes=1)
    print('  timed_iteration K=1 nt=%d: %.3f ms per propagation (backward %.3f s, update %.3f s)' % (
        nt, r['seconds'] / r['props'] * 1e3, r['backward_seconds'], r['update_seconds']))
import cProfile, pstats
spec = configs.config_c5(K=1, N=64, nt=201, L=1)
ops = [[spec.H0[0], spec.Hc[0][0]]]
st = spec.init[0].copy()
t = time.perf_counter()
for n in range(200):
    st = ko.step(ops[0], [0.3], 2.5e-4, st, False, False, True)
print('  ko.step (scipy): %.3f ms per call' % ((time.perf_counter() - t) / 200 * 1e3))
