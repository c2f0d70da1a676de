#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the reference.

Runs ONLY in the build container (needs /root/reference); the GPU box and the
test-suite never execute this file -- they read the committed ``.npz`` data.

Two sources:

A. ``dumps``   -- numeric arrays extracted from the pickled ``Result`` dumps the
   reference ships (tests/test_result_serialization/oct_result.dump and
   docs/notebooks/*.dump), read with a restricted unpickler that only
   reconstructs numpy arrays and stubs every other class.  The fixture stores
   the *inputs* of the run (operators, states, time grid, controls at the
   continuation iteration) next to the reference's recorded outputs
   (tau_vals, pulses, info_vals).

B. ``ref``     -- the reference's real ``krotov.optimize_pulses`` loop executed
   here in its documented "numpy mode" (reference
   docs/notebooks/09_example_numpy.ipynb), with QuTiP / glom / grapheme
   replaced by empty stub modules (they are not installed), on the seeded
   synthetic inputs of ``krotov_amd.configs``.  Outputs: all pulses, tau_vals.

Usage:  python tests/golden/make_reference_goldens.py [dumps] [ref] [c5full] [c4full] [c5full5] [c4full5] [second_order]

The reference is BSD-3-Clause (c) 2018-2024 Michael Goerz et al.; the fixtures
derived from its shipped data keep that attribution (tests/golden/README.md).
"""
import importlib
import os
import pickle
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)


# ---------------------------------------------------------------------------
# A. restricted unpickler
# ---------------------------------------------------------------------------

_ALLOW = {
    ('numpy.core.multiarray', '_reconstruct'),
    ('numpy._core.multiarray', '_reconstruct'),
    ('numpy.core.multiarray', 'scalar'),
    ('numpy._core.multiarray', 'scalar'),
    ('numpy', 'ndarray'),
    ('numpy', 'dtype'),
    ('time', 'struct_time'),
    ('builtins', 'complex'),
}


class _Stub:
    def __init__(self, *a, **k):
        self._args = a

    def __setstate__(self, st):
        self.__dict__['_state'] = st


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOW:
            return getattr(importlib.import_module(module), name)
        return type(name, (_Stub,), {})


def load_dump(path):
    with open(path, 'rb') as fh:
        obj = _Unpickler(fh).load()
    return obj.__dict__['_state']


def _lambda_ops(gamma=0.0):
    """Lambda-system operators of reference notebooks 02/03/08 (cell 6/7)."""
    E1, E2, E3, wP, wS = 0.0, 10.0, 5.0, 9.5, 4.5
    dP = E1 + wP - E2
    dS = E3 + wS - E2
    H0 = np.array([[dP, 0, 0], [0, -1j * gamma, 0], [0, 0, dS]], dtype=np.complex128)
    HP_re = -0.5 * np.array([[0, 1, 0], [1, 0, 0], [0, 0, 0]], dtype=np.complex128)
    HP_im = -0.5 * np.array([[0, 1j, 0], [-1j, 0, 0], [0, 0, 0]], dtype=np.complex128)
    HS_re = -0.5 * np.array([[0, 0, 0], [0, 0, 1], [0, 1, 0]], dtype=np.complex128)
    HS_im = -0.5 * np.array([[0, 0, 0], [0, 0, 1j], [0, -1j, 0]], dtype=np.complex128)
    tgt = np.exp(1j * (E2 - wS) * 5.0) * np.array([0, 0, 1], dtype=np.complex128)
    return H0, [HP_re, HP_im, HS_re, HS_im], tgt


def make_dump_fixtures():
    from krotov_amd import shapes

    # --- TLS, chis_ss, 19 iterations of pulses (notebook 01) ----------------
    st = load_dump(os.path.join(REF, 'tests/test_result_serialization/oct_result.dump'))
    np.savez_compressed(
        os.path.join(HERE, 'dump_tls_ss.npz'),
        tlist=np.asarray(st['tlist']),
        all_pulses=np.array([np.array(p) for p in st['all_pulses']]),  # (19, 1, 499)
        tau_vals=np.array(st['tau_vals']),
        info_vals=np.array(st['info_vals'], dtype=np.float64),
        guess_controls=np.array(st['guess_controls']),
        optimized_controls=np.array(st['optimized_controls']),
        iters=np.array(st['iters']),
    )
    print('dump_tls_ss: iters', st['iters'][:3], '...', st['iters'][-1])

    # --- ensemble K=5, N=3, L=4, chis_re (notebook 08) ----------------------
    st = load_dump(os.path.join(REF, 'docs/notebooks/ensemble_opt_result.dump'))
    H0, Hc, tgt = _lambda_ops()
    np.savez_compressed(
        os.path.join(HERE, 'dump_ensemble.npz'),
        tlist=np.asarray(st['tlist']),
        H0=H0, Hc=np.array(Hc), target=tgt,
        mu=np.array([0.9, 0.95, 1.0, 1.05, 1.1]),
        controls_it12=np.array(st['guess_controls']),  # continuation overwrote them
        tau_vals=np.array(st['tau_vals'][:40]),
        iters=np.array(st['iters'][:40]),
        lambda_a=0.5,
    )
    print('dump_ensemble: n tau', len(st['tau_vals']), 'iters[:14]', st['iters'][:14])

    # --- non-Hermitian Lambda system (notebook 03) ---------------------------
    st = load_dump(os.path.join(REF, 'docs/notebooks/non_herm_opt_result.dump'))
    H0, Hc, tgt = _lambda_ops(gamma=0.5)
    np.savez_compressed(
        os.path.join(HERE, 'dump_nonherm.npz'),
        tlist=np.asarray(st['tlist']),
        H0=H0, Hc=np.array(Hc), target=tgt,
        controls_it40=np.array(st['guess_controls']),
        tau_vals=np.array(st['tau_vals'][:60]),
        iters=np.array(st['iters'][:60]),
        lambda_a=2.0,
    )
    print('dump_nonherm: n tau', len(st['tau_vals']), 'iters[38:44]', st['iters'][38:44])

    # --- Lambda system RWA from the true guess (notebook 02) -----------------
    st = load_dump(os.path.join(REF, 'docs/notebooks/lambda_rwa_opt_result.dump'))
    H0, Hc, tgt = _lambda_ops()
    np.savez_compressed(
        os.path.join(HERE, 'dump_lambda_rwa.npz'),
        tlist=np.asarray(st['tlist']),
        H0=H0, Hc=np.array(Hc), target=tgt,
        guess_controls=np.array(st['guess_controls']),
        optimized_controls=np.array(st['optimized_controls']),
        tau_vals=np.array(st['tau_vals']),
        iters=np.array(st['iters']),
    )
    print('dump_lambda_rwa: iters', st['iters'])

    # --- transmon X gate, K=2, N=17 (notebook 05) ---------------------------
    st = load_dump(os.path.join(REF, 'docs/notebooks/transmonxgate_opt_result.dump'))
    Ec, EjEc, nstates = 0.386, 45, 8
    Ej = EjEc * Ec
    n = np.arange(-nstates, nstates + 1)
    up = np.diag(np.ones(2 * nstates), k=-1)
    H0 = (np.diag(4 * Ec * (n - 0.0) ** 2) - Ej * (up + up.T) / 2.0).astype(np.complex128)
    H1 = (-2 * np.diag(n)).astype(np.complex128)
    import scipy.linalg

    evals, evecs = scipy.linalg.eig(H0)  # as the notebook's logical_basis()
    ndx = np.argsort(evals.real)
    V = evecs[:, ndx]
    np.savez_compressed(
        os.path.join(HERE, 'dump_transmon17.npz'),
        tlist=np.asarray(st['tlist']),
        H0=H0, H1=H1, psi0=V[:, 0].astype(np.complex128), psi1=V[:, 1].astype(np.complex128),
        controls_it5=np.array(st['guess_controls']),
        tau_vals=np.array(st['tau_vals'][:12]),
        iters=np.array(st['iters'][:12]),
        lambda_a=1.0,
    )
    print('dump_transmon17: iters[:8]', st['iters'][:8])

    # --- two transmons in Liouville space, 625-dim sparse Liouvillian, K=3, L=2 (notebook 06) ----------
    # The reference's one result for its DensityMatrixODEPropagator (propagators.py:162-327).  The dump holds the
    # objectives themselves (QuTiP objects): this package's own Result.load rebuilds them as arrays; the super-
    # operators are kept in CSR form (a few entries per row).  The run was continued from iteration 3
    # (notebook cell 54), so `guess_controls` are the controls AT iteration 3.
    import scipy.sparse as sp

    import krotov_amd

    res = krotov_amd.result.Result.load(os.path.join(REF, 'docs/notebooks/3states_opt_result.dump'))
    objs = res.objectives
    L_ops = [np.asarray(objs[0].H[0])] + [np.asarray(objs[0].H[i][0]) for i in (1, 2)]
    for o in objs[1:]:  # all three objectives share the Liouvillian
        assert all(np.array_equal(np.asarray(o.H[0] if i == 0 else o.H[i][0]), L_ops[i]) for i in range(3))
    csr = [sp.csr_matrix(A) for A in L_ops]
    for c in csr:
        c.sum_duplicates()
        c.eliminate_zeros()
    arrays = {}
    for i, c in enumerate(csr):
        arrays['L%d_data' % i], arrays['L%d_indices' % i], arrays['L%d_indptr' % i] = c.data, c.indices, c.indptr
    # The reference integrates with zvode at rtol = 1e-6 / atol = 1e-8, and over 2 000 steps that is what its tau values
    # are good to (~1e-4).  For a tight comparison with an EXACT exponential action the fixture also holds the same
    # propagator's result with its tolerances tightened to rtol = 1e-11 / atol = 1e-13 -- through the oracle's
    # restatement of the propagator's zvode step (oracle/krotov_oracle.py: step_ode), which reproduces the dump at the
    # default tolerances to 2e-12 (the propagator itself needs QuTiP): tau at iteration 3 (one forward propagation
    # under `controls_it3`) and after one more Krotov iteration.
    from oracle import krotov_oracle as ko

    tl = np.asarray(res.tlist)
    rho0 = np.array([np.asarray(o.initial_state).ravel(order='F') for o in objs])
    tgt = np.array([np.asarray(o.target).ravel(order='F') for o in objs])
    prob = ko.OracleProblem([csr] * 3, rho0, tgt, tl, is_super=True, weights=np.array([o.weight for o in objs]),
                            ode=dict(rtol=1e-11, atol=1e-13))
    pulses3 = [ko.control_onto_interval(c) for c in np.array(res.guess_controls)]
    S = np.clip(ko.control_onto_interval(ko.discretize(lambda t: ko.flattop(t, 0.0, tl[-1], 20.0), tl, args=(),
                                                       via_midpoints=True)), 0, 1)
    tau3 = ko.tau_vals(prob, ko.forward_propagation(prob, pulses3))
    _, _, tau4, _ = ko.krotov_iteration(prob, pulses3, [S, S], [1.0, 1.0], None, None, ko.chis_re)
    print('dump_3states: tau(it 3) dump - tight zvode', np.abs(np.array(res.tau_vals[3]) - tau3).max(),
          ' tau(it 4)', np.abs(np.array(res.tau_vals[4]) - tau4).max())
    np.savez_compressed(
        os.path.join(HERE, 'dump_3states.npz'),
        tau_tight_it3=tau3, tau_tight_it4=tau4,
        tlist=np.asarray(res.tlist), N=625,
        rho0=np.array([np.asarray(o.initial_state) for o in objs]),    # (3, 25, 25)
        rho_tgt=np.array([np.asarray(o.target) for o in objs]),
        weights=np.array([o.weight for o in objs]),
        controls_it3=np.array(res.guess_controls),                     # (2, 2000) on the time grid
        tau_vals=np.array(res.tau_vals[:12]), info_vals=np.array(res.info_vals[:12], dtype=np.float64),
        iters=np.array(res.iters[:12]), lambda_a=1.0, t_rise=20.0,
        **arrays)
    print('dump_3states: nnz', [c.nnz for c in csr], 'iters[:6]', res.iters[:6], 'J_T[:6]', res.info_vals[:6])


# ---------------------------------------------------------------------------
# B. the real reference loop under stub third-party modules
# ---------------------------------------------------------------------------


def import_reference_krotov():
    """Import /root/reference/src/krotov with QuTiP & friends stubbed out."""
    if 'krotov' in sys.modules:
        return sys.modules['krotov']
    np.ComplexWarning = np.exceptions.ComplexWarning  # conversions.py:103

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _PB:
        def __init__(self, *a, **k):
            pass

        start = update = finished = lambda self, *a, **k: None

    class _Qobj:  # numpy mode: "Qobj(array)" (optimize.py:437-440) is just the array
        def __new__(cls, inpt=None, **kw):
            return np.asarray(inpt)

    mod('qutip', Qobj=_Qobj, expect=lambda *a: None)
    mod(
        'qutip.parallel',
        serial_map=lambda task, values, task_args=(), task_kwargs={}, **k: [
            task(v, *task_args, **task_kwargs) for v in values
        ],
    )
    mod('qutip.cy')
    mod('qutip.cy.spconvert', dense2D_to_fastcsr_fmode=None)
    mod('qutip.cy.spmatfuncs', spmvpy_csr=None)
    mod('qutip.superoperator', mat2vec=None, vec2mat=None)
    mod('qutip.solver', Options=object, Result=object)
    mod('qutip.ui')
    mod('qutip.ui.progressbar', BaseProgressBar=_PB, TextProgressBar=_PB)
    mod('glom', T=type('T', (), {'__getitem__': lambda s, i: ('T', i)})(), glom=None, GlomError=Exception)
    mod('grapheme', length=len)
    sys.path.insert(0, os.path.join(REF, 'src'))
    import krotov

    krotov.Objective.type_checking = False  # objectives.py:154-158
    return krotov


def run_reference(spec, iter_stop, krotov=None, sigma=None, after_iter=None):
    """Run the reference loop on a ProblemSpec; returns dict of outputs.  ``after_iter(outputs so far)`` is called
    from the reference's ``check_convergence`` hook after every iteration (optimize.py:550-551)."""
    import scipy.linalg as la

    from krotov_amd import configs

    if krotov is None:
        krotov = import_reference_krotov()
    objectives, pulse_options = configs.spec_to_objectives(spec, krotov)
    is_super = spec.is_super

    # numpy-mode plugins, exactly the reference's notebook 09 (cells 16, 30, 32)
    def expm(H, state, dt, c_ops=None, backwards=False, initialize=False):
        f = 1.0 + 0j if is_super else -1j  # propagators.py:94-99
        if backwards:
            f = f.conjugate()
        A = f * H[0]
        for part in H[1:]:
            A = A + (f * part[1]) * part[0]
        return la.expm(A * dt) @ state

    def mu(objs, i_obj, pulses, mapping, i_pulse, n):
        op = objs[i_obj].H[1 + i_pulse][0]
        if is_super:  # mu.py:130-134
            return lambda s: 1j * (op @ s)
        return lambda s: op @ s

    def overlap(a, b):
        return complex(np.vdot(a, b))

    chi = getattr(krotov.functionals, 'chis_' + spec.chi)
    t0 = time.time()

    def outputs(res, with_controls=True):
        out = dict(
            all_pulses=np.array([np.array(p) for p in res.all_pulses]),
            tau_vals=np.array(res.tau_vals),
            fw_T=np.array([np.asarray(s).ravel() for s in res.states]),
            seconds=time.time() - t0,
        )
        if with_controls:
            out['optimized_controls'] = np.array(res.optimized_controls)
        return out

    def hook(result):
        if after_iter is not None:
            after_iter(outputs(result, with_controls=False))
        return None

    res = krotov.optimize_pulses(
        objectives, pulse_options, spec.tlist,
        propagator=expm, chi_constructor=chi, mu=mu, overlap=overlap,
        norm=np.linalg.norm, iter_stop=iter_stop, store_all_pulses=True, sigma=sigma,
        check_convergence=hook,
    )
    return outputs(res)


REF_CASES = {
    # name: (builder kwargs -> spec, iter_stop)
    'ref_c1_tls': (lambda c: c.config_c1(), 5),
    'ref_c2_hilbert': (lambda c: c.config_c2_hilbert(), 5),
    'ref_c2_liouville': (lambda c: c.config_c2_liouville(), 5),
    'ref_c3_iswap': (lambda c: c.config_c3(), 5),
    'ref_c4_small': (lambda c: c.config_c4(d=5, nt=201, n_logical=2), 5),
    'ref_c5_small': (lambda c: c.config_c5(K=6, N=16, nt=201, L=1), 5),
    'ref_c5_small_L3': (lambda c: c.config_c5(K=5, N=12, nt=151, L=3, distinct=True), 5),
    'ref_c5_n64': (lambda c: c.config_c5(K=8, N=64, nt=401, L=1), 5),
    # chis_hs (functionals.py:389-437): the boundary co-state depends on rho(T) itself, not only on tau
    # (config_c2_liouville is no use here: one of its three states is invariant, chi_k(T) = 0, and the reference
    # divides by its norm -- NaN pulses)
    'ref_c4_small_hs': (lambda c: _with_chi(c.config_c4(d=5, nt=201, n_logical=2), 'hs'), 5),
}


def _with_chi(spec, chi):
    spec.chi = chi
    return spec


def make_ref_fixtures(names=None):
    from krotov_amd import configs

    krotov = import_reference_krotov()
    for name, (builder, iters) in REF_CASES.items():
        if names and name not in names:
            continue
        spec = builder(configs)
        out = run_reference(spec, iters, krotov)
        np.savez_compressed(os.path.join(HERE, name + '.npz'), iter_stop=iters, **out)
        print('%-18s K=%d N=%d nt=%d L=%d  %.1fs  tau[-1][:2]=%s' % (
            name, spec.K, spec.N, len(spec.tlist), spec.L, out['seconds'], out['tau_vals'][-1][:2]))


def make_second_order():
    """Second-order update (sigma(t) term; reference optimize.py:434-443, 468-469,
    492-500, 566-577) on the two-qubit iSWAP system with chis_sm; sigma as in the
    reference's notebook 07 (cell 30), A re-estimated every iteration from
    Delta J_T and the final states (numerical_estimate_A's formula, evaluated
    with NumPy because the reference's own helper needs Qobj states)."""
    from krotov_amd import configs

    krotov = import_reference_krotov()
    spec = configs.config_c3(nt=201)
    spec.lambda_a = 20.0

    class Sigma(krotov.second_order.Sigma):
        def __init__(self, A, epsA):
            self.A, self.epsA, self.history = A, epsA, []

        def __call__(self, t):
            return -max(self.epsA, 2 * self.A + self.epsA)

        def refresh(self, forward_states, forward_states0, chi_states, chi_norms, optimized_pulses,
                    guess_pulses, objectives, result):
            J = lambda tau: 1 - abs(np.sum(tau) / len(tau)) ** 2  # noqa: E731  (J_T_sm)
            dJ = J(result.tau_vals[-1]) - J(result.tau_vals[-2])
            n = len(objectives)
            dphi = [np.asarray(forward_states[k][-1]) - np.asarray(forward_states0[k][-1]) for k in range(n)]
            denom = sum(np.vdot(d, d).real for d in dphi)
            numer = sum((2 * chi_norms[k] * np.vdot(chi_states[k], dphi[k])).real for k in range(n)) + dJ
            self.A = numer / denom if denom > 1e-30 else 0
            self.history.append(self.A)

    sig = Sigma(0.0, 2.0)
    out = run_reference(spec, 3, krotov, sigma=sig)
    np.savez_compressed(os.path.join(HERE, 'ref_so_c3.npz'), iter_stop=3, A_history=np.array(sig.history), **out)
    print('ref_so_c3: A history', sig.history, ' tau[-1][:2]', out['tau_vals'][-1][:2])


def make_print_table_cases():
    """Text written by the reference's own ``krotov.info_hooks.print_table`` for three
    synthetic iteration records (two pulses with per-pulse columns; ASCII headers;
    custom formats/headers) -> tests/golden/print_table_cases.txt."""
    import io

    krotov = import_reference_krotov()
    out = io.StringIO()
    J = lambda **kw: kw['J']  # noqa: E731
    common = dict(guess_pulses=[None, None], iter_stop=10, start_time=0.0, stop_time=2.4)
    hook = krotov.info_hooks.print_table(J_T=J, show_g_a_int_per_pulse=True, out=out)
    hook(iteration=0, J=1.0, g_a_integrals=np.zeros(2), info_vals=[], **common)
    hook(iteration=1, J=0.5, g_a_integrals=np.array([0.1, 0.2]), info_vals=[1.0], **common)
    hook(iteration=2, J=0.6, g_a_integrals=np.array([0.0, 0.05]), info_vals=[1.0, 0.5], **common)
    out.write("--\n")
    hook = krotov.info_hooks.print_table(J_T=J, unicode=False, out=out)
    one = dict(guess_pulses=[None], iter_stop=12345, start_time=0.0, stop_time=0.0)
    hook(iteration=0, J=1.0, g_a_integrals=np.zeros(1), info_vals=[], **one)
    hook(iteration=1, J=0.25, g_a_integrals=np.array([0.5]), info_vals=[1.0], **one)
    out.write("--\n")
    hook = krotov.info_hooks.print_table(
        J_T=J, show_g_a_int_per_pulse=True, out=out,
        col_formats=('%03d', '%.6f', '%.3e', '%.3e', '%.6f', '%+.1e', '%+.1e', '%4d'),
        col_headers=('#', 'error', 'ga[{l}]', 'ga', 'total', 'd(error)', 'd(total)', 's'))
    hook(iteration=0, J=1.0, g_a_integrals=np.zeros(2), info_vals=[], **common)
    hook(iteration=1, J=0.5, g_a_integrals=np.array([0.1, 0.2]), info_vals=[1.0], **common)
    with open(os.path.join(HERE, 'print_table_cases.txt'), 'w', encoding='utf8') as fh:
        fh.write(out.getvalue())
    print(out.getvalue())


def debug_information_cases():
    """Keyword arguments of three synthetic iterations for ``print_debug_information`` (shared with
    tests/test_host_helpers.py, which feeds them to krotov_amd's)."""
    def expm():
        pass

    def chis_re():
        pass

    def derivative_wrt_pulse():
        pass

    base = dict(objectives=['<objective 1>', '<objective 2>'], adjoint_objectives=['<adjoint 1>', '<adjoint 2>'],
                guess_pulses=[np.zeros(4), np.zeros(4)], lambda_vals=np.array([5.0, 0.25]),
                shape_arrays=[np.array([0.0, 0.5, 1.0, 0.0]), np.ones(4)], tlist=np.linspace(0, 1, 5),
                start_time=0.0, info_vals=[], shared_data={}, propagator=expm, chi_constructor=chis_re,
                mu=derivative_wrt_pulse, sigma=None, iter_start=0, iter_stop=7, fw_states_T=[])
    bw = [np.zeros((5, 3), dtype=complex), np.zeros((5, 3), dtype=complex)]
    return [
        dict(base, iteration=0, backward_states=None, forward_states=None, forward_states0=None,
             optimized_pulses=[np.array([-1.0, 1.0, 5.0, 0.0]), np.array([0.0, 0.25, 0.5, 0.125])],
             g_a_integrals=np.zeros(2), tau_vals=np.array([0.5 + 0.5j, -0.25j]), stop_time=1.25),
        dict(base, iteration=1, backward_states=bw, forward_states=None, forward_states0=None,
             optimized_pulses=[np.array([-1.5, 1.0, 5.5, 0.0]), np.array([0.0, 0.25 + 1j, 0.5, 0.125 - 2j])],
             g_a_integrals=np.array([0.0123, 4.5e-6]), tau_vals=np.array([0.9, -0.8 + 0.1j]), stop_time=63.0),
        dict(base, iteration=2, backward_states=bw, forward_states=bw, forward_states0=bw, propagator=[expm, expm],
             optimized_pulses=[np.array([-1.5, 1.0, 5.5, 0.0]), np.array([0.0, 0.25, 0.5, 0.125])],
             g_a_integrals=np.array([1.0, 2.0]), tau_vals=np.array([None, None]), stop_time=0.04),
    ]


def make_print_debug_cases():
    """Text written by the reference's own ``krotov.info_hooks.print_debug_information`` for the synthetic
    iterations above -> tests/golden/print_debug_cases.txt.  (Empty list of final states: the reference sizes them through
    Qobj internals.  The iteration-0 block with a LIST of propagators is not exercised: the reference raises a
    TypeError there, info_hooks.py:184-188.)"""
    import io

    krotov = import_reference_krotov()
    out = io.StringIO()
    for kw in debug_information_cases():
        krotov.info_hooks.print_debug_information(out=out, **kw)
        out.write("--\n")
    with open(os.path.join(HERE, 'print_debug_cases.txt'), 'w', encoding='utf8') as fh:
        fh.write(out.getvalue())
    print(out.getvalue())


def summarize_cases(objective_cls, make_q):
    """Texts of Objective.summarize for objectives of NumPy arrays and of QuTiP-like objects (shared with
    tests/test_reference_unit_cases.py).  ``make_q(type, dims, isherm)`` builds a QuTiP-like object."""
    lines = []
    u1, u2 = (lambda t, args: 1.0), (lambda t, args: 1.0)
    a1, a2 = np.zeros(100, dtype=complex), np.ones(100, dtype=complex)
    # NumPy mode
    H = [np.eye(4, dtype=complex), [np.ones((4, 4), dtype=complex), u1], [np.zeros((4, 4), dtype=complex), u2]]
    psi0, psi1 = np.zeros(4, dtype=complex), np.ones(4, dtype=complex)
    C1, C2 = [[np.eye(4), a1]], [[np.eye(4) * 2, a2]]
    obj = objective_cls(initial_state=psi0, target=psi1, H=H)
    obj.reset_symbol_counters()
    lines.append(obj.summarize())
    obj2 = objective_cls(initial_state=psi0, target=psi1, H=H, c_ops=[C1, C2])
    lines.append(obj2.summarize())
    lines.append(obj2.summarize(use_unicode=False))
    lines.append(str(obj2))
    lines.append(repr(obj))
    # QuTiP-like components
    ket = lambda: make_q('ket', [[2, 2], [1, 1]], False)  # noqa: E731
    oper = lambda herm: make_q('oper', [[2, 2], [2, 2]], herm)  # noqa: E731
    sup = lambda: make_q('super', [[[2, 2], [2, 2]], [[2, 2], [2, 2]]], False)  # noqa: E731
    k0, k1 = ket(), ket()
    Hq = [oper(True), [oper(True), u1], [oper(False), u2]]
    obj3 = objective_cls(initial_state=k0, target=k1, H=Hq, c_ops=[[[oper(False), a1]], oper(False)])
    lines.append(obj3.summarize(reset_symbol_counters=True))
    lines.append(obj3.summarize(use_unicode=False))
    rho0, rho1 = oper(True), oper(True)
    obj4 = objective_cls(initial_state=rho0, target=rho1, H=[sup(), [sup(), u1]])
    lines.append(obj4.summarize())
    lines.append(objective_cls(initial_state=k0, target='PE', H=Hq).summarize())
    lines.append(objective_cls(initial_state=make_q('bra', [[1, 1], [2, 2]], False), target=None,
                               H=[Hq[0], [Hq[1][0], 0.5]]).summarize(use_unicode=False))
    return lines


def make_summarize_cases():
    """Text of the reference's own ``Objective.summarize`` / ``str`` / ``repr`` for the cases above ->
    tests/golden/summarize_cases.txt (QuTiP-like objects: instances of a subclass of the stubbed qutip.Qobj)."""
    krotov = import_reference_krotov()
    import qutip

    class FakeQ(qutip.Qobj):
        def __new__(cls, *a, **k):
            return object.__new__(cls)

    def make_q(kind, dims, isherm):
        q = FakeQ()
        q.type, q.dims, q.isherm = kind, dims, isherm
        return q

    lines = summarize_cases(krotov.Objective, make_q)
    with open(os.path.join(HERE, 'summarize_cases.txt'), 'w', encoding='utf8') as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))


def make_c5_full():
    """Headline configuration through the real reference loop: 1 iteration
    (~2.5 sweeps * 256 * 4000 props at ~0.8 ms each => ~35 min, one core)."""
    from krotov_amd import configs

    spec = configs.config_c5()
    out = run_reference(spec, 1)
    np.savez_compressed(os.path.join(HERE, 'ref_c5_full.npz'), iter_stop=1, **out)
    print('ref_c5_full: %.0fs' % out['seconds'])


def make_c4_full():
    """BASELINE config 4 at full size (16 density matrices, 400-dim Liouvillian,
    1000 intervals) through the real reference loop: 1 iteration = 3 sweeps *
    16 * 1000 dense 400x400 ``expm`` (~80 ms each on one core)."""
    from krotov_amd import configs

    spec = configs.config_c4()
    out = run_reference(spec, 1)
    np.savez_compressed(os.path.join(HERE, 'ref_c4_full.npz'), iter_stop=1, **out)
    print('ref_c4_full: %.0fs' % out['seconds'])


def make_full5(which):
    """FIVE iterations of the reference's loop at full size ("after 1, 2, 5 iterations on every config"):
    BASELINE config 5 (11 sweeps * 256 * 4000 dense 64 x 64 expm, ~2.6 CPU-hours) -> ref_c5_full5.npz; config 4
    (11 * 16 * 1000 dense 400 x 400 expm, ~3.2 CPU-hours) -> ref_c4_full5.npz.  The fixture is written after EVERY
    iteration (iter_stop = iterations it holds so far): an interrupted run still leaves a usable one."""
    from krotov_amd import configs

    spec = configs.config_c5() if which == 'c5' else configs.config_c4()
    # (KH_FULL5_OUT: another file -- a re-run next to a partial fixture that the suite is using)
    path = os.environ.get('KH_FULL5_OUT') or os.path.join(HERE, 'ref_%s_full5.npz' % which)

    def save(out):
        n_done = len(out['all_pulses']) - 1
        if n_done >= 1:
            np.savez_compressed(path + '.tmp.npz', iter_stop=n_done, **out)
            os.replace(path + '.tmp.npz', path)
            print('ref_%s_full5: %d iteration(s), %.0fs' % (which, n_done, out['seconds']), flush=True)

    out = run_reference(spec, 5, after_iter=save)
    save(out)


if __name__ == '__main__':
    what = sys.argv[1:] or ['dumps', 'ref']
    if 'c5full5' in what:
        make_full5('c5')
    if 'c4full5' in what:
        make_full5('c4')
    if 'dumps' in what:
        make_dump_fixtures()
    if 'ref' in what:
        make_ref_fixtures()
    if 'c5full' in what:
        make_c5_full()
    if 'c4full' in what:
        make_c4_full()
    if 'second_order' in what:
        make_second_order()
    if 'print_table' in what:
        make_print_table_cases()
    if 'print_debug' in what:
        make_print_debug_cases()
    if 'summarize' in what:
        make_summarize_cases()
    for w in what:
        if w in REF_CASES:
            make_ref_fixtures([w])
