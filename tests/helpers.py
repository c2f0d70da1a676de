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


class SigmaA:
    """sigma(t) = -max(epsA, 2A + epsA) with A re-estimated every iteration (the
    reference's notebook 07, cell 30); oracle-side array form of the Sigma used
    for tests/golden/ref_so_c3.npz."""

    def __init__(self, A=0.0, epsA=2.0):
        self.A, self.epsA, self.history = A, epsA, []

    def __call__(self, t):
        return -max(self.epsA, 2 * self.A + self.epsA)

    def refresh(self, fw_T, fw_T0, chi_T, chi_norms, taus):
        J = lambda tau: 1 - abs(np.sum(tau) / len(tau)) ** 2  # noqa: E731  (J_T_sm)
        self.A = ko.numerical_estimate_A(fw_T, fw_T0, chi_T, chi_norms, J(taus[-1]) - J(taus[-2]))
        self.history.append(self.A)


def product_sigma(A=0.0, epsA=2.0):
    """The same sigma through the product's plugin surface: a
    ``krotov_amd.second_order.Sigma`` subclass whose ``refresh`` calls
    ``krotov_amd.second_order.numerical_estimate_A`` with Delta J_T (J_T_sm) taken
    from ``result.tau_vals``."""
    import krotov_amd
    from krotov_amd.second_order import Sigma, numerical_estimate_A

    class _Sigma(Sigma):
        def __init__(self):
            self.A, self.epsA, self.history, self.calls = A, epsA, [], []

        def __call__(self, t):
            return -max(self.epsA, 2 * self.A + self.epsA)

        def refresh(self, forward_states, forward_states0, chi_states, chi_norms, optimized_pulses, guess_pulses,
                    objectives, result):
            J = krotov_amd.functionals.J_T_sm
            dJ = J(None, objectives, tau_vals=result.tau_vals[-1]) - J(None, objectives, tau_vals=result.tau_vals[-2])
            self.A = numerical_estimate_A(forward_states, forward_states0, chi_states, chi_norms, dJ)
            self.history.append(self.A)
            self.calls.append((len(forward_states), len(forward_states[0]), guess_pulses is optimized_pulses))

    return _Sigma()


def numpy_plugins(is_super=False):
    """propagator / mu / overlap callables as in reference notebook 09."""

    def propagator(H, state, dt, c_ops=None, backwards=False, initialize=False):
        f = (1.0 + 0j) if is_super else -1j
        if backwards:
            f = f.conjugate()
        A = f * H[0]
        for part in H[1:]:
            A = A + (f * part[1]) * part[0]
        return ko.expm_dense(A * dt, use_scipy=False) @ state

    def mu(objs, i_obj, pulses, mapping, i_pulse, n):
        op = objs[i_obj].H[1 + i_pulse][0]
        return (lambda s: 1j * (op @ s)) if is_super else (lambda s: op @ s)

    def overlap(a, b):
        return complex(np.vdot(a, b))

    return propagator, mu, overlap


def check_infohook_chaining(**optimize_kwargs):
    """The scenario of reference tests/test_infohooks.py:15-72 (5-level transmon, two
    intervals, lambda_a halved after every iteration by modify_params_after_iter, two
    chained info_hooks) through ``krotov_amd.optimize_pulses(**optimize_kwargs)``."""
    import scipy.linalg

    import krotov_amd

    Ec, EjEc, nstates, T = 0.386, 45, 2, 10.0
    Ej = EjEc * Ec
    n = np.arange(-nstates, nstates + 1)
    up = np.diag(np.ones(2 * nstates), k=-1)
    H0 = (np.diag(4 * Ec * n**2) - Ej * (up + up.T) / 2.0).astype(complex)
    H1 = (-2 * np.diag(n)).astype(complex)
    eigenvals, eigenvecs = scipy.linalg.eig(H0)
    ndx = np.argsort(eigenvals.real)
    E, V = eigenvals[ndx].real, eigenvecs[:, ndx]
    w01 = E[1] - E[0]
    psi0, psi1 = V[:, 0].astype(complex), V[:, 1].astype(complex)
    eps0 = lambda t, args: 0.5 * np.exp(-40.0 * (t / T - 0.5) ** 2) * np.cos(8 * np.pi * w01 * t)  # noqa: E731
    H = [H0, [H1, eps0]]
    obj = krotov_amd.Objective(initial_state=psi0, target=psi1, H=H)
    tlist = np.array([0, 0.01, 0.02])
    printed = []

    def adjust_lambda_a(**args):
        before = args['lambda_vals'][0]
        args['lambda_vals'][0] *= 0.5
        args['shared_data'].setdefault('messages', []).append('λₐ: %s → %s' % (before, args['lambda_vals'][0]))

    def print_fidelity(**args):
        F_re = np.average(np.array(args['tau_vals']).real)
        printed.append("Iteration %d: \tF = %f" % (args['iteration'], F_re))
        return F_re

    def print_messages(**args):
        if 'messages' in args['shared_data']:
            message = "; ".join(args['shared_data']['messages'])
            printed.append("\tmsg: " + message)
            return message

    import io

    report = io.StringIO()

    def debug_information(**args):  # the full-signature hook works on whatever state containers the path hands out
        krotov_amd.info_hooks.print_debug_information(out=report, **args)

    res = krotov_amd.optimize_pulses(
        [obj], pulse_options={H[1][1]: dict(lambda_a=1, update_shape=1)}, tlist=tlist,
        chi_constructor=krotov_amd.functionals.chis_re,
        info_hook=krotov_amd.info_hooks.chain(print_fidelity, print_messages, debug_information),
        modify_params_after_iter=adjust_lambda_a, iter_stop=2, **optimize_kwargs)
    out = "\n".join(printed)
    assert report.getvalue().count('    storage (bw, fw, fw0): [1 * ') == 2  # iterations 1 and 2
    assert '    fw_states_T norm: 1.000000\n' in report.getvalue()
    assert len(res.info_vals) == 3
    assert isinstance(res.info_vals[1], tuple) and len(res.info_vals[1]) == 2
    assert abs(res.info_vals[1][0] - 0.001978333994757067) < 1e-8
    assert res.info_vals[1][1] == 'λₐ: 0.5 → 0.25'
    assert 'Iteration 0: \tF = 0.000000' in out
    assert 'msg: λₐ: 1.0 → 0.5' in out
    assert 'Iteration 1: \tF = 0.001978' in out
    assert 'msg: λₐ: 0.5 → 0.25' in out
    return res
