"""Per-round cycle breakdown of the four-wave cooperative sweep (kh_coop4w.h): needs a -DKH_TIMING build (KH_TIMING_LIB)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from krotov_amd import _lib
_lib.LIB_PATH = os.environ.get('KH_TIMING_LIB', _lib.LIB_PATH)
os.environ.setdefault('KH_COOP4W', '1')
from krotov_amd import configs
from krotov_amd.engine import HipKrotovEngine
spec = configs.config_c4(nt=1001)
K, N, L = spec.K, spec.N, spec.L
ops = [[spec.H0[k]] + [spec.Hc[k][l] for l in range(L)] for k in range(K)]
eng = HipKrotovEngine(ops, np.diff(spec.tlist), is_super=True)
eng.profile = True
tl = spec.tlist
pulses = np.array([[spec.controls[l](t + 0.5 * (tl[1] - tl[0]), None) for t in tl[:-1]] for l in range(L)])
chi_T = spec.target / np.linalg.norm(spec.target, axis=1)[:, None]
for _ in range(2):
    chi = eng.backward(chi_T, pulses)
torch.cuda.synchronize()
buf = (ctypes.c_double * 4)()
eng._lib.kh_last_stats(eng._handle, buf)
print('backward %.2f ms; cycles per round (wave 0 of workgroup 0): fetch %.0f | matrix cores + block sums %.0f | between rounds %.0f; '
      'rounds entering the slow path %.0f %%' % (min(eng.kernel_times_ms()['backward']), buf[1], buf[2], buf[3] % 1e6, int(buf[3] / 1e6)))
