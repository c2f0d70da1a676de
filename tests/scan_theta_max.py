"""Dev check (GPU, run by hand: python tests/scan_theta_max.py): Taylor sub-step bound theta_max of the
cooperative kernels vs rounds per interval and error against the oracle.  Lives under tests/ because it uses the
oracle (test infrastructure)."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from krotov_amd import configs
from krotov_amd.engine import HipKrotovEngine
from oracle import krotov_oracle as ko
from helpers import spec_to_oracle, oracle_controls
for name, spec in (('c4_d9', configs.config_c4(d=9, nt=41, n_logical=2)), ('c4_d10', configs.config_c4(d=10, nt=21, n_logical=3))):
    prob = spec_to_oracle(spec)
    gp, S, lam = oracle_controls(spec)
    ref_T, ref_states = ko.forward_propagation(prob, gp, store=True)
    ops = [[spec.H0[k]] + [spec.Hc[k][l] for l in range(spec.L)] for k in range(spec.K)]
    for tm in (1.0, 2.0, 3.0, 4.0, 6.0):
        eng = HipKrotovEngine(ops, np.diff(spec.tlist), is_super=True, theta_max=tm)
        fw_T, states = eng.forward(np.array(gp), spec.init, store=True)
        mv = eng.stats()['matvecs'] / (spec.K * (len(spec.tlist) - 1))
        print(name, 'theta_max', tm, 'rounds/step %.1f' % mv, 'max err vs oracle %.2e' % np.abs(states.cpu().numpy() - ref_states).max())
        eng.close()
