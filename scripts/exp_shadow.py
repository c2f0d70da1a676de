#!/usr/bin/env python3
"""A/B of an experiment build of the library (KH_LIB=<path>) on the config-5 update sweep: time per interval and the
updated pulses / final states saved to gpurun_out/<tag>.npz for a comparison between builds (dev tool).
usage: KH_LIB=<lib> python scripts/exp_shadow.py <tag> [K] [nt] [L]; python scripts/exp_shadow.py --compare <tagA> <tagB>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
out_dir = os.path.join(ROOT, 'gpurun_out')
os.makedirs(out_dir, exist_ok=True)
if sys.argv[1] == '--compare':
    a = np.load(os.path.join(out_dir, sys.argv[2] + '.npz'))
    b = np.load(os.path.join(out_dir, sys.argv[3] + '.npz'))
    print('max |d pulses| = %.3e   max |d psi(T)| = %.3e   max |d g_a| = %.3e' % (
        np.abs(a['opt'] - b['opt']).max(), np.abs(a['psi'] - b['psi']).max(), np.abs(a['ga'] - b['ga']).max()))
    if 'chi0' in a and 'chi0' in b:
        print('max |d chi(0)| = %.3e' % np.abs(a['chi0'] - b['chi0']).max())
    sys.exit(0)
from krotov_amd import _lib
if os.environ.get('KH_LIB'):
    _lib.LIB_PATH = os.environ['KH_LIB']
from krotov_amd import configs
from krotov_amd.engine import HipKrotovEngine
tag = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 4001
L = int(sys.argv[4]) if len(sys.argv) > 4 else 1
N = 64
spec = configs.config_c5(K=K, N=N, nt=nt, L=L)
ops = [[spec.H0[k]] + [spec.Hc[k][l] for l in range(L)] for k in range(K)]
eng = HipKrotovEngine(ops, np.diff(spec.tlist))
eng.profile = True
tl = spec.tlist
pulses = np.array([[0.5 * np.sin((l + 1) * np.pi * (t + 0.5 * (tl[1] - tl[0])) / tl[-1]) for t in tl[:-1]] for l in range(L)])
chi_T = spec.target / np.linalg.norm(spec.target, axis=1)[:, None]
for _ in range(4):
    chi = eng.backward(chi_T, pulses)
norms = np.full(K, 1.0 / (2 * K))
S = np.array([[spec.update_shape(0.5 * (tl[i] + tl[i + 1])) for i in range(nt - 1)]] * L)
for _ in range(4):
    opt, psi, ga = eng.forward_update(chi, norms, spec.init, pulses, S, np.array([spec.lambda_a] * L))
eng.check()
t = eng.kernel_times_ms()
ms = min(t['update'])
print('%s %s K=%d nt=%d L=%d lib=%s: update sweep %.3f ms = %.3f us per interval (backward %.3f ms)' % (
    tag, eng.kernel, K, nt, L, os.path.basename(_lib.LIB_PATH), ms, ms * 1e3 / (nt - 1), min(t['backward'])))
np.savez(os.path.join(out_dir, tag + '.npz'), chi0=chi[:, 0].cpu().numpy(), opt=opt.cpu().numpy(), psi=psi.cpu().numpy(), ga=ga.cpu().numpy())
