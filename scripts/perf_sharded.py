#!/usr/bin/env python3
"""Cost of the stepwise (sharded) update sweep on one rank: no-op vs RCCL all-reduce, eager vs graph."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.distributed as dist
from krotov_amd import configs
from krotov_amd.engine import HipKrotovEngine
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29544', RANK='0', WORLD_SIZE='1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
K, N, nt = 256, 64, 4001
spec = configs.config_c5(K=K, N=N, nt=nt)
ops = [[spec.H0[k], spec.Hc[k][0]] for k in range(K)]
eng = HipKrotovEngine(ops, np.diff(spec.tlist))
tl = spec.tlist
pulses = np.array([[0.5 * np.sin(np.pi * (t + 0.5 * (tl[1] - tl[0])) / tl[-1]) for t in tl[:-1]]])
chi_T = spec.target / np.linalg.norm(spec.target, axis=1)[:, None]
chi = eng.backward(chi_T, pulses)
norms = np.full(K, 1.0 / (2 * K)); S = np.ones((1, nt - 1)); lam = np.array([50.0])
ref = eng.forward_update(chi, norms, spec.init, pulses, S, lam)
def ar_nccl(x): dist.all_reduce(x)
def ar_noop(x): pass
for name, ar in (('noop', ar_noop), ('nccl', ar_nccl)):
    for chunk in (0, 64):
        for _ in range(2):
            torch.cuda.synchronize(); t = time.perf_counter()
            out = eng.forward_update_sharded(chi, norms, spec.init, pulses, S, lam, ar, graph_chunk=chunk)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
        eng._sh['graph'] = None
        err = float((out[0] - ref[0]).abs().max())
        print('%-5s chunk=%3d  %.1f ms  (%.1f us/interval)  max|d pulse| vs single-launch %.1e' % (name, chunk, dt * 1e3, dt * 1e6 / (nt - 1), err))
dist.destroy_process_group()
