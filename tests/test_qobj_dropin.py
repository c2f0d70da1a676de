"""QuTiP-shaped control problems through ``optimize_pulses`` (the north star's "QuTiP-defined
control problems drop in unchanged").

The problems are BASELINE configs 1-5 built the way a user of the reference builds them -- ``krotov_amd.gate_objectives``
(reference objectives.py:704-1051), ``ensemble_objectives`` (:1054-1094), ``objectives.liouvillian`` (:1097-1121) -- from
``tests/qobj_double.QobjDouble`` objects (kets with tensor ``dims``, '3states' / 'full' density matrices, super-operator
``H`` with ``.type == 'super'``) instead of ndarrays.  Every case runs twice: ``-m gpu`` on the HIP engine, and without a
GPU on the oracle-backed engine double (tests/oracle_engine_double.py), which exercises the same host code (ingestion,
type dispatch, states handed back) so that it can be debugged where there is no GPU.

Bars: pulses and tau BIT-IDENTICAL to the ndarray run of the same spec (the engine must see the same bytes), <= 2e-12
vs the fixtures of the reference's own loop (``ref_c1_tls`` / ``ref_c2_*`` / ``ref_c3_iswap`` ...); ``result.states`` and the
states ``info_hook`` sees come back as the double's class with the right ``dims``.
"""
import numpy as np
import pytest

import krotov_amd
from krotov_amd import configs

from helpers import golden, product_sigma
from qobj_double import QobjDouble, ket, oper

# ---------------------------------------------------------------------------------------------------------------------
# the problems: (objectives from doubles, pulse_options, spec of plain arrays, propagator)
# ---------------------------------------------------------------------------------------------------------------------


def _options(spec):
    return {c: dict(lambda_a=spec.lambda_a, update_shape=spec.update_shape) for c in spec.controls}


def problem_c1():
    """Config 1: |0> -> |1> of a two-level system, one Objective built by hand."""
    spec = configs.config_c1()
    H = [oper(spec.H0[0]), [oper(spec.Hc[0][0]), spec.controls[0]]]
    objectives = [krotov_amd.Objective(initial_state=ket(0, 2), target=ket(1, 2), H=H)]
    return objectives, spec, krotov_amd.propagators.expm


def problem_c2_hilbert():
    """Config 2, Hilbert space: gate_objectives(basis, X, H)."""
    spec = configs.config_c2_hilbert()
    H = [oper(spec.H0[0]), [oper(spec.Hc[0][0]), spec.controls[0]]]
    X = oper([[0, 1], [1, 0]])
    objectives = krotov_amd.gate_objectives([ket(0, 2), ket(1, 2)], X, H)
    return objectives, spec, krotov_amd.propagators.expm


def problem_c2_liouville():
    """Config 2, Liouville space: H -> liouvillian(H, c_ops=[]) (a nested list of super-operators),
    gate_objectives(..., liouville_states_set='3states') (density matrices)."""
    spec = configs.config_c2_liouville()
    tls = configs.config_c1()
    H = [oper(tls.H0[0]), [oper(tls.Hc[0][0]), spec.controls[0]]]
    L = krotov_amd.objectives.liouvillian(H, c_ops=[])
    assert isinstance(L[0], QobjDouble) and L[0].type == 'super' and L[1][0].type == 'super'
    assert L[0].dims == [[[2], [2]], [[2], [2]]] and L[1][1] is spec.controls[0]
    X = np.array([[0, 1], [1, 0]], dtype=complex)  # (a gate may be any matrix-like: reference objectives.py:722)
    objectives = krotov_amd.gate_objectives([ket(0, 2), ket(1, 2)], X, L, liouville_states_set='3states')
    assert all(o.initial_state.type == 'oper' and o.initial_state.dims == [[2], [2]] for o in objectives)
    return objectives, spec, krotov_amd.propagators.expm


def problem_c3():
    """Config 3: two-qubit iSWAP; kets and operators with tensor dims [[2, 2], ...]."""
    spec = configs.config_c3()
    H = [oper(spec.H0[0], [2, 2]), [oper(spec.Hc[0][0], [2, 2]), spec.controls[0]]]
    basis = [ket(idx, (2, 2)) for idx in ((0, 0), (0, 1), (1, 0), (1, 1))]
    iswap = oper([[1, 0, 0, 0], [0, 0, 1j, 0], [0, 1j, 0, 0], [0, 0, 0, 1]], [2, 2])
    objectives = krotov_amd.gate_objectives(basis, iswap, H)
    assert objectives[0].target is basis[0] and objectives[3].target is basis[3]  # (permuted basis states are reused)
    return objectives, spec, krotov_amd.propagators.expm


def problem_c4_small():
    """Config 4 at d = 5: the transmon's Liouvillian WITH a decay operator from liouvillian(H, c_ops=[C]), 'full' set."""
    spec = configs.config_c4(d=5, nt=201, n_logical=2)
    d = 5
    Ec, Ej, ng, gamma = 0.386, 45 * 0.386, 0.0, 1e-3
    n = np.arange(-(d // 2), d - d // 2)
    up = np.diag(np.ones(d - 1), k=-1)
    H0 = (np.diag(4 * Ec * (n - ng) ** 2) - Ej * (up + up.T) / 2.0).astype(np.complex128)
    H1 = (-2 * np.diag(n)).astype(np.complex128)
    evals, V = np.linalg.eigh(H0)
    for j in range(d):
        i = np.argmax(np.abs(V[:, j]))
        if V[i, j].real < 0:
            V[:, j] = -V[:, j]
    C = np.sqrt(gamma) * (V @ np.diag(np.sqrt(np.arange(1, d)), k=1) @ V.conj().T)
    L = krotov_amd.objectives.liouvillian([oper(H0), [oper(H1), spec.controls[0]]], c_ops=[oper(C)])
    basis = [QobjDouble(V[:, j].astype(np.complex128).reshape(-1, 1)) for j in range(2)]
    objectives = krotov_amd.gate_objectives(basis, oper([[0, 1], [1, 0]]), L, liouville_states_set='full')
    return objectives, spec, krotov_amd.propagators.expm


def problem_c5_small():
    """Config 5 (small): ensemble_objectives over Hamiltonians with scaled control operators (the reference's
    notebook 08 pattern), the original objective dropped."""
    spec = configs.config_c5(K=6, N=16, nt=201, L=1)
    H0 = oper(spec.H0[0])
    psi0, psi1 = ket(0, 16), ket(1, 16)
    Hs = [[H0, [oper(spec.Hc[k][0]), spec.controls[0]]] for k in range(spec.K)]
    base = [krotov_amd.Objective(initial_state=psi0, target=psi1, H=Hs[0])]
    objectives = krotov_amd.ensemble_objectives(base, Hs, keep_original_objectives=False)
    assert len(objectives) == spec.K and all(o.H[0] is H0 for o in objectives)
    return objectives, spec, krotov_amd.propagators.expm


def problem_sparse_lindblad():
    """The DensityMatrixODEPropagator drop-in: a Lindbladian whose super-operators are doubles (their ``.data`` is the
    CSR triplet the reference reads, propagators.py:269-273), density-matrix states."""
    spec = configs.config_sparse_lindblad(d=8, nt=41, K=3)
    d = 8
    made = {}

    def sup(arr):
        if id(arr) not in made:
            made[id(arr)] = QobjDouble(arr, dims=[[[d], [d]], [[d], [d]]])
        return made[id(arr)]

    objectives = []
    for k in range(spec.K):
        H = [sup(spec.H0[k])] + [[sup(spec.Hc[k][l]), spec.controls[l]] for l in range(spec.L)]
        rho0 = QobjDouble(spec.init[k].reshape(d, d, order='F'), dims=[[d], [d]])
        rho1 = QobjDouble(spec.target[k].reshape(d, d, order='F'), dims=[[d], [d]])
        objectives.append(krotov_amd.Objective(initial_state=rho0, target=rho1, H=H))
    return objectives, spec, krotov_amd.propagators.DensityMatrixODEPropagator()


PROBLEMS = {
    'c1': (problem_c1, 'ref_c1_tls'),
    'c2_hilbert': (problem_c2_hilbert, 'ref_c2_hilbert'),
    'c2_liouville': (problem_c2_liouville, 'ref_c2_liouville'),
    'c3': (problem_c3, 'ref_c3_iswap'),
    'c4_small': (problem_c4_small, 'ref_c4_small'),
    'c5_small': (problem_c5_small, 'ref_c5_small'),
    'sparse_lindblad': (problem_sparse_lindblad, None),
}


def _run(objectives, spec, propagator, iters, **kw):
    return krotov_amd.optimize_pulses(
        objectives, _options(spec), spec.tlist, propagator=propagator,
        chi_constructor=getattr(krotov_amd.functionals, 'chis_' + spec.chi),
        iter_stop=iters, store_all_pulses=True, **kw)


def _array_run(spec, propagator, iters, **kw):
    objectives, _ = configs.spec_to_objectives(spec, krotov_amd)
    if isinstance(propagator, krotov_amd.propagators.DensityMatrixODEPropagator):
        prop = propagator
    else:
        prop = krotov_amd.propagators.HipExpm(liouville=True) if spec.is_super else krotov_amd.propagators.expm
    return _run(objectives, spec, prop, iters, **kw)


def _check_problem(name, on_gpu):
    build, fixture = PROBLEMS[name]
    objectives, spec, propagator = build()
    g = golden(fixture) if fixture else None
    iters = int(g['iter_stop']) if g is not None else 2
    if not on_gpu:
        iters = min(iters, 2)
        if name in ('c1', 'c2_hilbert', 'c2_liouville', 'c3'):  # (the oracle double steps in Python)
            spec.tlist = spec.tlist[:61]
    seen = []

    def hook(**args):
        it = args['iteration']
        fw = args['fw_states_T']
        rec = {'fw': [fw[k] for k in range(len(fw))]}
        if it > 0:
            bw = args['backward_states']
            rec['bw'] = [bw[k][0] for k in range(len(objectives))] + [bw[0][len(spec.tlist) - 1]]
        seen.append(rec)
        return it

    n0 = QobjDouble.N_CONSTRUCTED
    res = _run(objectives, spec, propagator, iters, info_hook=hook)
    arr = _array_run(spec, propagator, iters)
    got = np.array([np.array(p) for p in res.all_pulses])
    want = np.array([np.array(p) for p in arr.all_pulses])
    # the engine saw the same bytes: nothing may differ
    assert np.array_equal(got, want)
    assert np.array_equal(np.array(res.tau_vals), np.array(arr.tau_vals))
    assert np.array_equal(np.array(res.optimized_controls), np.array(arr.optimized_controls))
    if g is not None and on_gpu:
        tol = 2e-11 if name == 'c4_small' else 2e-12
        scale = max(1.0, np.abs(g['all_pulses']).max())
        assert np.abs(got - g['all_pulses']).max() < tol * scale
        assert np.abs(np.array(res.tau_vals) - g['tau_vals']).max() < tol
    # states come back as what went in: the double's class, through its constructor, with the initial state's dims
    assert QobjDouble.N_CONSTRUCTED > n0
    assert len(res.states) == len(objectives)
    for k, (state, obj) in enumerate(zip(res.states, objectives)):
        assert type(state) is QobjDouble and state.dims == obj.initial_state.dims
        assert state.type == obj.initial_state.type
        flat = state.full().ravel(order='F')
        assert np.array_equal(flat, np.asarray(arr.states[k]).ravel(order='F'))
        if g is not None and on_gpu:
            assert np.abs(flat - g['fw_T'][k]).max() < (2e-11 if name == 'c4_small' else 2e-12)
    assert len(seen) == iters + 1
    for rec in seen:
        assert all(type(s) is QobjDouble and s.dims == o.initial_state.dims for s, o in zip(rec['fw'], objectives))
        for s in rec.get('bw', []):
            assert type(s) is QobjDouble and s.dims == objectives[0].initial_state.dims
    # the co-state at T handed to the hook is the normalised boundary condition of the functional
    last = seen[-1]['bw'][-1]
    assert abs(np.linalg.norm(last.full()) - 1.0) < 1e-12


@pytest.fixture
def oracle_engine(monkeypatch):
    """No GPU: the oracle-backed engine double stands in for the HIP engine (host logic only)."""
    import krotov_amd.engine as engine_mod
    from oracle_engine_double import OracleEngineDouble

    monkeypatch.setattr(engine_mod, 'HipKrotovEngine', OracleEngineDouble)


@pytest.mark.parametrize('name', sorted(PROBLEMS))
def test_qobj_problems_host_logic(name, oracle_engine):
    _check_problem(name, on_gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(PROBLEMS))
def test_qobj_problems_on_device(name):
    _check_problem(name, on_gpu=True)
    if name == 'sparse_lindblad':
        from krotov_amd.engine import LAST_ENGINE

        assert LAST_ENGINE().kernel == 'ell/csr'


# ---------------------------------------------------------------------------------------------------------------------
# the rest of the surface: user chi_constructor on doubles, second order, mixed lists, single-step expm
# ---------------------------------------------------------------------------------------------------------------------


def _check_user_chi_constructor_and_second_order(on_gpu):
    """A chi_constructor the optimiser does not know (so the co-states are built ON THE HOST from the doubles:
    scalar * target, ``.norm()``, ``/``), and the second-order update whose ``sigma.refresh`` works on states of the
    double's class (``-``, ``_overlap`` -> ``.overlap()`` / ``.dag()`` / ``.tr()``, second_order.py:69-83)."""
    objectives, spec, propagator = problem_c3()
    if not on_gpu:
        spec.tlist = spec.tlist[:41]
    calls = []

    def my_chis_sm(fw_states_T, objectives, tau_vals):
        calls.append(all(type(s) is QobjDouble for s in fw_states_T))
        return krotov_amd.functionals.chis_sm(fw_states_T, objectives, tau_vals)

    res = krotov_amd.optimize_pulses(objectives, _options(spec), spec.tlist, propagator=propagator,
                                     chi_constructor=my_chis_sm, iter_stop=2, store_all_pulses=True)
    arr = _array_run(spec, propagator, 2)
    assert calls == [True, True]
    got, want = np.array(res.all_pulses), np.array(arr.all_pulses)
    # (the host-side normalisation rounds differently from kh_chi_boundary: 1e-15, not bits)
    assert np.abs(got - want).max() < 1e-14 * max(1.0, np.abs(want).max())
    assert np.abs(np.array(res.tau_vals) - np.array(arr.tau_vals)).max() < 1e-14

    sig_q, sig_a = product_sigma(), product_sigma()
    res = _run(objectives, spec, propagator, 3, sigma=sig_q)
    arr = _array_run(spec, propagator, 3, sigma=sig_a)
    assert len(sig_q.history) == 2 and np.abs(np.array(sig_q.history) - np.array(sig_a.history)).max() < 1e-9
    got, want = np.array(res.all_pulses), np.array(arr.all_pulses)
    assert np.abs(got - want).max() < 1e-12 * max(1.0, np.abs(want).max())


def test_user_chi_constructor_and_second_order_host_logic(oracle_engine):
    _check_user_chi_constructor_and_second_order(False)


@pytest.mark.gpu
def test_user_chi_constructor_and_second_order_on_device():
    _check_user_chi_constructor_and_second_order(True)


def _check_mixed_and_density_matrix_inference():
    """Objectives of doubles and of ndarrays in one list are handled (every objective is ingested by its own kind;
    each state comes back as what that objective's initial state is)."""
    objectives, spec, propagator = problem_c2_hilbert()
    spec.tlist = spec.tlist[:41]
    arr_objs, _ = configs.spec_to_objectives(spec, krotov_amd)
    H = objectives[0].H
    mixed = [objectives[0],
             krotov_amd.Objective(initial_state=arr_objs[1].initial_state, target=arr_objs[1].target, H=H)]
    res = _run(mixed, spec, propagator, 2)
    arr = _array_run(spec, propagator, 2)
    assert np.array_equal(np.array(res.all_pulses), np.array(arr.all_pulses))
    assert type(res.states[0]) is QobjDouble and isinstance(res.states[1], np.ndarray)
    assert res.states[1].shape == (2, 1)
    # array super-operators (no ``.type``) with density matrices of the double's class: Liouville space is recognised
    # from the state's shape, read through ``.full()``
    spec = configs.config_c2_liouville()
    spec.tlist = spec.tlist[:41]
    arr_objs, _ = configs.spec_to_objectives(spec, krotov_amd)
    objs = [krotov_amd.Objective(
        initial_state=QobjDouble(spec.init[k].reshape(2, 2, order='F')),
        target=QobjDouble(spec.target[k].reshape(2, 2, order='F')), H=arr_objs[k].H) for k in range(spec.K)]
    res = _run(objs, spec, krotov_amd.propagators.expm, 2)
    arr = _array_run(spec, propagator, 2)
    assert np.array_equal(np.array(res.all_pulses), np.array(arr.all_pulses))
    assert all(type(s) is QobjDouble and s.type == 'oper' for s in res.states)


def test_mixed_objective_list_host_logic(oracle_engine):
    _check_mixed_and_density_matrix_inference()


@pytest.mark.gpu
def test_mixed_objective_list_on_device():
    _check_mixed_and_density_matrix_inference()


@pytest.mark.gpu
def test_single_step_expm_type_dispatch_on_device():
    """krotov_amd.propagators.expm called directly, the reference's dispatch on ``.type`` (propagators.py:96-117):
    oper on ket, super on density matrix, anything else NotImplementedError; result in the state's class."""
    import scipy.linalg

    spec = configs.config_c3()
    H0, H1 = oper(spec.H0[0], [2, 2]), oper(spec.Hc[0][0], [2, 2])
    psi = ket((0, 1), (2, 2))
    out = krotov_amd.propagators.expm([H0, [H1, 0.3]], psi, 0.01)
    assert type(out) is QobjDouble and out.dims == psi.dims and out.type == 'ket'
    want = scipy.linalg.expm(-1j * (spec.H0[0] + 0.3 * spec.Hc[0][0]) * 0.01) @ psi.full()
    assert np.abs(out.full() - want).max() < 1e-14
    back = krotov_amd.propagators.expm([H0, [H1, 0.3]], psi, 0.01, backwards=True)
    want = scipy.linalg.expm(+1j * (spec.H0[0] + 0.3 * spec.Hc[0][0]) * 0.01) @ psi.full()
    assert np.abs(back.full() - want).max() < 1e-14
    L = krotov_amd.objectives.liouvillian([H0, [H1, 0.3]], c_ops=[])
    rho = psi * psi.dag()
    out = krotov_amd.propagators.expm(L, rho, 0.01)
    assert type(out) is QobjDouble and out.dims == rho.dims and out.type == 'oper'
    U = scipy.linalg.expm(-1j * (spec.H0[0] + 0.3 * spec.Hc[0][0]) * 0.01)
    assert np.abs(out.full() - U @ rho.full() @ U.conj().T).max() < 1e-14
    with pytest.raises(NotImplementedError, match="Cannot handle argument types A:super, state:ket"):
        krotov_amd.propagators.expm(L, psi, 0.01)
    with pytest.raises(NotImplementedError, match="Cannot handle argument types A:oper, state:oper"):
        krotov_amd.propagators.expm([H0, [H1, 0.3]], rho, 0.01)
    with pytest.raises(NotImplementedError, match="Liouville exponentiation not implemented"):
        krotov_amd.propagators.expm([H0, [H1, 0.3]], rho, 0.01, c_ops=[H1])
