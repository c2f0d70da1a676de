import os, sys, ctypes
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from krotov_amd import _lib
_lib.LIB_PATH = os.environ.get('KH_TIMING_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'libkrotov_hip_timing.so'))
from krotov_amd import configs
from krotov_amd.engine import HipKrotovEngine
spec = configs.config_c4(nt=1001)
K, N, L = spec.K, spec.N, spec.L
ops = [[spec.H0[k]] + [spec.Hc[k][l] for l in range(L)] for k in range(K)]
eng = HipKrotovEngine(ops, np.diff(spec.tlist), is_super=True)
tl = spec.tlist
pulses = np.array([[spec.controls[l](t + 0.5 * (tl[1] - tl[0]), None) for t in tl[:-1]] for l in range(L)])
chi_T = spec.target / np.linalg.norm(spec.target, axis=1)[:, None]
for _ in range(2):
    chi = eng.backward(chi_T, pulses)
buf = (ctypes.c_double * 4)()
os.environ['KH_TRACE'] = '1'  # (the timing build prints the stamps behind the four statistics: [0] = term blocks through one XCD's L2?)
torch.cuda.synchronize()
eng._lib.kh_last_stats(eng._handle, buf)
print('rounds*cols %d  cycles/round: poll %.0f (fast pass %d)  mfma+lds write %.0f  barrier %.0f | stale lanes after the fast pass %.2f/64, '
      'agent-scope passes per round %.2f' % (buf[0], buf[1] % 1e6, int(buf[1] / 1e6), buf[2] % 1e6, buf[3] % 1e6,
                                           int(buf[2] / 1e6) / 100.0, int(buf[3] / 1e6) / 100.0))

# update sweep: how long every column group (one per XCD) waits in the exchange of the update sums
S = np.ones((L, len(tl) - 1)); lam = np.full(L, 1.0); norms = np.full(K, 1.0 / (2 * K))
eng.profile = True
for _ in range(2):
    out = eng.forward_update(chi, norms, spec.init, pulses, S, lam)
torch.cuda.synchronize()
print('update sweep %.2f ms; KH_TRACE raw [16..23] = exchange wait per column group, [24] fragment update, [25] inside the rounds, '
      '[26] between rounds + interval boundary: cycles per interval' % min(eng.kernel_times_ms()['update']))
eng._lib.kh_last_stats(eng._handle, buf)
