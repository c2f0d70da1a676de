"""Shared test helpers: golden loading and spec -> oracle conversion."""
import os

import numpy as np

from oracle import krotov_oracle as ko

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

CHI = {'re': ko.chis_re, 'ss': ko.chis_ss, 'sm': ko.chis_sm, 'hs': ko.chis_hs}


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def spec_to_oracle(spec):
    ops = [[spec.H0[k]] + [spec.Hc[k][l] for l in range(spec.L)] for k in range(spec.K)]
    return ko.OracleProblem(ops, spec.init, spec.target, spec.tlist, spec.is_super, spec.weights)


def oracle_controls(spec):
    """(guess_pulses, shape_arrays, lambdas) of a spec through the oracle."""
    _, gp, S = ko.initialize_controls(spec.controls, [spec.update_shape] * spec.L, spec.tlist)
    return gp, S, [spec.lambda_a] * spec.L


def oracle_optimize(spec, iter_stop, **kw):
    gp, S, lam = oracle_controls(spec)
    # numpy-mode reference runs pass norm=np.linalg.norm (notebook 09, cell 35)
    kw.setdefault('norm', lambda prob, chi: float(np.linalg.norm(chi)))
    return ko.optimize(spec_to_oracle(spec), gp, S, lam, CHI[spec.chi], iter_stop, **kw)
