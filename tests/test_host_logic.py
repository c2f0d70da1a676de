"""Host-side logic on CPU: conversions, shapes, functionals, objectives and the
plugin path of optimize_pulses (driven by user callables -- here NumPy closures
built on the oracle's single-step function, the reference's "numpy mode").

KAT sources: reference tests/test_structural_conversions.py, test_shapes.py,
test_functionals.py, test_mu.py, test_overlap.py, test_krotov.py, and the TLS
dump of tests/test_result_serialization.
"""
import copy
import os

import numpy as np
import pytest

import krotov_amd
from krotov_amd import configs, conversions, functionals, shapes
from oracle import krotov_oracle as ko

from helpers import golden, numpy_plugins


def test_controls_roundtrip_and_boundaries():
    rng = np.random.default_rng(0)
    c = rng.standard_normal(50)
    p = conversions.control_onto_interval(c)
    assert len(p) == 49 and p[0] == c[0] and p[-1] == c[-1]
    back = conversions.pulse_onto_tlist(p)
    assert np.abs(back[:-2] - c[:-2]).max() < 1e-12  # the recurrence inverts the averaging
    assert np.abs(conversions.control_onto_interval(back) - p).max() < 1e-12  # pulse -> control -> pulse = id
    assert back[0] == c[0] and back[-1] == p[-1]
    # matches the oracle's restatement bit for bit
    assert np.array_equal(p, ko.control_onto_interval(c))
    assert np.array_equal(back, ko.pulse_onto_tlist(p))


def test_discretize_rejects_complex_and_wrong_length():
    tl = np.linspace(0, 1, 11)
    with pytest.raises(TypeError):
        conversions.discretize(lambda t, a: 1j * t, tl)
    with pytest.raises(ValueError):
        conversions.discretize(np.zeros(5), tl)
    with pytest.raises(TypeError):
        conversions.discretize("nope", tl)
    vals = conversions.discretize(lambda t, a: t**2, tl, via_midpoints=True)
    assert len(vals) == 11 and vals[0] == 0.0


def test_shapes_match_oracle_bitwise():
    for t in np.linspace(-0.5, 5.5, 241):
        for func in ('blackman', 'sinsq'):
            assert shapes.flattop(t, 0, 5, 0.3, func=func) == ko.flattop(t, 0, 5, 0.3, func=func)
        assert shapes.blackman(t, 1.0, 4.0) == ko.blackman(t, 1.0, 4.0)
    assert abs(shapes.flattop(0.0, 0, 5, 0.3)) < 1e-16 and shapes.flattop(2.5, 0, 5, 0.3) == 1.0
    with pytest.raises(ValueError):
        shapes.flattop(1.0, 0, 5, 0.3, func='nope')
    cb = shapes.qutip_callback(shapes.flattop, t_start=0, t_stop=5, t_rise=0.3)
    assert cb(2.5, None) == 1.0 and cb(2.5, {'func': 'sinsq'}) == 1.0


def test_mapping_and_plug_in(dummy_objectives):
    X, Y, Z = np.eye(2), 2 * np.eye(2), 3 * np.eye(2)
    u1, u2 = np.zeros(3), np.ones(3)
    o1 = krotov_amd.Objective(initial_state=None, target=None, H=[X, [Y, u1], [Z, u1]])
    o2 = krotov_amd.Objective(initial_state=None, target=None, H=[X, [Y, u2]])
    controls = conversions.extract_controls([o1, o2])
    assert len(controls) == 2 and controls[0] is u1 and controls[1] is u2
    mapping = conversions.extract_controls_mapping([o1, o2], controls)
    assert mapping == [[[[1, 2], []]], [[[], [1]]]]
    H = conversions.plug_in_pulse_values(['X', ['X', None], ['Y', None], ['Z', None]],
                                         [np.array([0, 10, 0]), np.array([0, 20, 0])], [[1, 2], [3]], 1)
    assert H == ['X', ['X', 10], ['Y', 10], ['Z', 20]]
    with pytest.raises(ValueError):
        conversions.pulse_options_dict_to_list({}, controls)


def test_functionals_known_answers():
    """reference tests/test_functionals.py: J_T_ss=0.25? -> values on fixed taus."""
    class O:
        def __init__(self, target):
            self.target = target
    objs = [O(np.array([1, 0], dtype=complex)), O(np.array([0, 1], dtype=complex))]
    taus = np.array([0.5, 1.0], dtype=complex)
    assert abs(functionals.F_ss(None, objs, taus) - (0.25 + 1.0) / 2) < 1e-14
    assert abs(functionals.F_sm(None, objs, taus) - 0.75**2) < 1e-14
    assert abs(functionals.F_re(None, objs, taus) - 0.75) < 1e-14
    assert abs(functionals.J_T_re(None, objs, taus) - 0.25) < 1e-14
    for name in ('re', 'ss', 'sm', 'hs'):
        fn = getattr(functionals, 'chis_' + name)
        fw = [np.array([0.6, 0.8j]), np.array([0.1, 0.9])]
        listed = np.array(fn(fw, objs, taus))
        stacked = functionals.chi_stacked(fn, np.array([o.target for o in objs]), None, np.array(fw), taus)
        assert np.abs(listed - stacked).max() < 1e-15, name
        c, d = functionals.chi_coefficients(fn, None, taus, 2)  # what kh_chi_boundary is fed
        combined = c[:, None] * np.array([o.target for o in objs]) + d[:, None] * np.array(fw)
        assert np.abs(listed - combined).max() < 1e-15, name
    assert functionals.chi_coefficients(functionals.chis_ss, None, [None, None], 2) is None
    assert functionals.chi_coefficients(lambda **kw: None, None, taus, 2) is None
    objs[0].weight, objs[1].weight = 0.5, 1.5
    listed = np.array(functionals.chis_sm(None, objs, taus))
    stacked = functionals.chi_stacked(functionals.chis_sm, np.array([o.target for o in objs]),
                                      np.array([0.5, 1.5]), None, taus)
    assert np.abs(listed - stacked).max() < 1e-15
    c, d = functionals.chi_coefficients(functionals.chis_sm, np.array([0.5, 1.5]), taus, 2)
    assert np.abs(listed - c[:, None] * np.array([o.target for o in objs])).max() < 1e-15 and not d.any()


def test_overlap_and_mu(dummy_objectives):
    from krotov_amd.mu import derivative_wrt_pulse
    from krotov_amd.second_order import _overlap

    a = np.array([[1, 2j], [0, 1]], dtype=complex)  # non-Hermitian: must use a^dagger
    b = np.array([[0.5, 1], [1j, 2]], dtype=complex)
    assert abs(_overlap(a, b) - np.trace(a.conj().T @ b)) < 1e-14
    assert _overlap('PE', b) is None
    sp = np.array([[0, 1], [0, 0]], dtype=complex)
    sm = sp.T.copy()
    u = np.zeros(3)
    obj = krotov_amd.Objective(initial_state=None, target=None, H=[np.eye(2), [sp, u], [sm, u]])
    mapping = conversions.extract_controls_mapping([obj], [u])
    mu = derivative_wrt_pulse([obj], 0, [u], mapping, 0, 0)
    v = np.array([0.3, 0.7j])
    assert np.abs(mu(v) - (sp + sm) @ v).max() < 1e-15  # repeated control: sum of terms
    other = np.ones(3)
    mapping2 = conversions.extract_controls_mapping([obj], [u, other])
    zero = derivative_wrt_pulse([obj], 0, [u, other], mapping2, 1, 0)
    assert np.all(zero(v) == 0)


def test_objective_copy_eq_adjoint_and_constructors():
    H0 = np.diag([1.0, -1.0]).astype(complex)
    H1 = np.array([[0, 1j], [-1j, 0]])
    u = lambda t, args: 1.0  # noqa: E731
    psi = np.array([1, 0], dtype=complex)
    obj = krotov_amd.Objective(initial_state=psi, target=psi[::-1].copy(), H=[H0, [H1 * 1j, u]])
    obj.weight = 0.7
    c = copy.copy(obj)
    assert c == obj and c.H is not obj.H and c.H[1] is not obj.H[1] and c.H[1][0] is obj.H[1][0]
    assert c.weight == 0.7
    adj = obj.adjoint()
    assert np.array_equal(adj.H[1][0], (H1 * 1j).conj().T) and adj.H[1][1] is u and adj.weight == 0.7
    d = copy.deepcopy(obj)
    assert d == obj
    d.weight = 0.1
    assert d != obj
    # gate objectives: X gate on a qubit reuses the basis objects
    basis = [np.array([1, 0], dtype=complex), np.array([0, 1], dtype=complex)]
    X = np.array([[0, 1], [1, 0]])
    objs = krotov_amd.gate_objectives(basis, X, [H0, [H1, u]])
    assert len(objs) == 2 and objs[0].target is basis[1] and objs[1].target is basis[0]
    o3 = krotov_amd.gate_objectives(basis, X, [H0, [H1, u]], liouville_states_set='3states', weights=[20, 1, 1])
    assert len(o3) == 3 and abs(sum(o.weight for o in o3) - 3) < 1e-12
    assert abs(np.trace(o3[0].initial_state) - 1) < 1e-14
    full = krotov_amd.gate_objectives(basis, X, [H0, [H1, u]], liouville_states_set='full')
    assert len(full) == 4
    with pytest.raises(ValueError):
        krotov_amd.gate_objectives(basis, 'nope', [H0])
    ens = krotov_amd.ensemble_objectives(objs, [[H0, [0.9 * H1, u]], [H0, [1.1 * H1, u]]])
    assert len(ens) == 6 and ens[2].initial_state is basis[0]
    L = krotov_amd.objectives.liouvillian([H0, [H1, u]], c_ops=[np.array([[0, 1], [0, 0]])])
    assert L[0].shape == (4, 4) and L[1][1] is u
    assert np.abs(L[1][0] - configs.liouvillian_dense(H1)).max() == 0


def _run_plugin(spec, iters, **kw):
    objectives, pulse_options = configs.spec_to_objectives(spec, krotov_amd)
    prop, mu, overlap = numpy_plugins(spec.is_super)
    return krotov_amd.optimize_pulses(
        objectives, pulse_options, spec.tlist, propagator=prop,
        chi_constructor=getattr(krotov_amd.functionals, 'chis_' + spec.chi),
        mu=mu, overlap=overlap, norm=np.linalg.norm, iter_stop=iters, store_all_pulses=True, **kw)


def test_plugin_path_reproduces_tls_dump():
    """Config 1 (plumbing, no GPU): the host loop with NumPy plugins vs the
    reference's shipped TLS result, all pulses of the first 6 iterations."""
    g = golden('dump_tls_ss')
    res = _run_plugin(configs.config_c1(), 6)
    got = np.array([np.array(p) for p in res.all_pulses])
    assert np.abs(got[0] - g['all_pulses'][0]).max() == 0.0
    assert np.abs(got - g['all_pulses'][:7]).max() < 1e-9
    assert np.abs(np.array(res.tau_vals)[:, 0] - g['tau_vals'][:7, 0]).max() < 1e-9
    assert res.iters == list(range(7)) and res.message == "Reached 6 iterations"
    assert len(res.optimized_controls[0]) == len(g['tlist'])
    assert np.abs(res.optimized_controls[0] - conversions.pulse_onto_tlist(got[-1][0])).max() == 0


@pytest.mark.parametrize('name,builder', [
    ('ref_c2_liouville', lambda: configs.config_c2_liouville()),
    ('ref_c5_small_L3', lambda: configs.config_c5(K=5, N=12, nt=151, L=3, distinct=True)),
])
def test_plugin_path_matches_reference_loop(name, builder):
    g = golden(name)
    res = _run_plugin(builder(), int(g['iter_stop']))
    got = np.array([np.array(p) for p in res.all_pulses])
    assert np.abs(got - g['all_pulses']).max() < 1e-11
    assert np.abs(np.array(res.tau_vals) - g['tau_vals']).max() < 1e-11


def test_plugin_path_second_order_matches_reference_loop():
    """sigma= through the host loop (reference optimize.py:429-442, 451-452, 468-469,
    492-500, 566-577) vs the reference's own loop on the same problem."""
    from helpers import product_sigma

    g = golden('ref_so_c3')
    spec = configs.config_c3(nt=201)
    spec.lambda_a = 20.0
    sig = product_sigma(0.0, 2.0)
    seen = []

    def hook(**kw):
        seen.append((kw['iteration'], kw['forward_states'] is not None, kw['forward_states0'] is not None))

    res = _run_plugin(spec, int(g['iter_stop']), sigma=sig, info_hook=hook)
    got = np.array([np.array(p) for p in res.all_pulses])
    assert np.abs(got - g['all_pulses']).max() < 1e-11
    assert np.abs(np.array(res.tau_vals) - g['tau_vals']).max() < 1e-11
    # refresh after every iteration but the last, with the full trajectories, and
    # (like the reference) after guess_pulses was advanced to the optimized ones
    assert np.abs(np.array(sig.history) - g['A_history']).max() < 1e-10
    assert sig.calls == [(len(g['tau_vals'][0]), len(spec.tlist), True)] * 2
    assert seen == [(i, True, True) for i in range(4)]
    objectives, pulse_options = configs.spec_to_objectives(spec, krotov_amd)
    prop, mu, overlap = numpy_plugins(spec.is_super)
    with pytest.raises(ValueError, match='skip_initial_forward_propagation'):
        krotov_amd.optimize_pulses(
            objectives, pulse_options, spec.tlist, propagator=prop, mu=mu, overlap=overlap,
            chi_constructor=krotov_amd.functionals.chis_sm, sigma=sig, iter_stop=1,
            skip_initial_forward_propagation=True)


def test_numerical_estimate_A_known_answers():
    """reference second_order.py:148-164 on hand-checkable vectors."""
    from krotov_amd.second_order import numerical_estimate_A

    fw0 = [[None, np.array([1.0, 0.0], dtype=complex)], [None, np.array([0.0, 1.0], dtype=complex)]]
    fw = [[None, np.array([1.0, 0.5j], dtype=complex)], [None, np.array([0.5, 1.0], dtype=complex)]]
    chis = [np.array([0.0, 1.0j]), np.array([1.0, 0.0], dtype=complex)]
    # dphi = (0, .5j), (.5, 0): denominator .5; numerator 2*2*Re(conj(1j)*.5j) + 2*3*.5 - .25 = 2 + 3 - .25
    assert abs(numerical_estimate_A(fw, fw0, chis, [2.0, 3.0], -0.25) - 9.5) < 1e-14
    assert numerical_estimate_A(fw0, fw0, chis, [2.0, 3.0], -0.25) == 0


def test_continue_from_equals_uninterrupted():
    """reference tests/test_krotov.py:426-432: continuation == one long run (1e-10)."""
    spec = configs.config_c1(nt=120)
    full = _run_plugin(spec, 4)
    objectives, pulse_options = configs.spec_to_objectives(spec, krotov_amd)
    prop, mu, overlap = numpy_plugins()
    kw = dict(propagator=prop, chi_constructor=krotov_amd.functionals.chis_ss, mu=mu, overlap=overlap,
              norm=np.linalg.norm, store_all_pulses=True)
    first = krotov_amd.optimize_pulses(objectives, pulse_options, spec.tlist, iter_stop=2, **kw)
    cont = krotov_amd.optimize_pulses(objectives, pulse_options, spec.tlist, iter_stop=4, continue_from=first, **kw)
    assert cont.iters == [0, 1, 2, 3, 4]
    assert np.abs(cont.optimized_controls[0] - full.optimized_controls[0]).max() < 1e-10
    with pytest.raises(ValueError):
        krotov_amd.optimize_pulses(objectives, pulse_options, spec.tlist[:-1], iter_stop=3, continue_from=first, **kw)
    with pytest.raises(ValueError):
        krotov_amd.optimize_pulses(objectives[:0] + objectives + objectives, pulse_options, spec.tlist,
                                   iter_stop=3, continue_from=first, **kw)


def test_validation_errors():
    """reference tests/test_krotov.py:22-134, test_pulse_options.py."""
    spec = configs.config_c1(nt=30)
    objectives, pulse_options = configs.spec_to_objectives(spec, krotov_amd)
    prop, mu, overlap = numpy_plugins()
    kw = dict(propagator=prop, chi_constructor=krotov_amd.functionals.chis_re, mu=mu, overlap=overlap,
              norm=np.linalg.norm, iter_stop=1)
    ctrl = spec.controls[0]
    with pytest.raises(ValueError, match='lambda_a'):
        krotov_amd.optimize_pulses(objectives, {ctrl: dict(update_shape=1)}, spec.tlist, **kw)
    with pytest.raises(ValueError, match='update_shape'):
        krotov_amd.optimize_pulses(objectives, {ctrl: dict(lambda_a=1.0)}, spec.tlist, **kw)
    with pytest.raises(ValueError, match=r'range \[0, 1\]'):
        krotov_amd.optimize_pulses(objectives, {ctrl: dict(lambda_a=1.0, update_shape=lambda t: 2.0)}, spec.tlist, **kw)
    with pytest.raises(ValueError, match='real-valued'):
        krotov_amd.optimize_pulses(objectives, {ctrl: dict(lambda_a=1.0, update_shape=lambda t: 1j)}, spec.tlist, **kw)
    with pytest.raises(ValueError, match='pulse options'):
        krotov_amd.optimize_pulses(objectives, {}, spec.tlist, **kw)
    # complex control
    H = [spec.H0[0], [spec.Hc[0][0], lambda t, args: 1j]]
    objs = [krotov_amd.Objective(initial_state=spec.init[0], target=spec.target[0], H=H)]
    with pytest.raises(ValueError, match='real-valued'):
        krotov_amd.optimize_pulses(objs, {H[1][1]: dict(lambda_a=1.0, update_shape=1)}, spec.tlist, **kw)


def test_info_hook_and_convergence_contract():
    spec = configs.config_c1(nt=60)
    seen = []

    def hook(**kwargs):
        needed = {'objectives', 'adjoint_objectives', 'backward_states', 'forward_states', 'forward_states0',
                  'guess_pulses', 'optimized_pulses', 'g_a_integrals', 'lambda_vals', 'shape_arrays',
                  'fw_states_T', 'tlist', 'tau_vals', 'start_time', 'stop_time', 'iteration', 'info_vals',
                  'shared_data', 'propagator', 'chi_constructor', 'mu', 'sigma', 'iter_start', 'iter_stop'}
        assert needed <= set(kwargs)
        seen.append(kwargs['iteration'])
        return 1 - abs(kwargs['tau_vals'][0]) ** 2

    def halve(**kwargs):
        kwargs['lambda_vals'][:] *= 0.5

    def converged(result):
        return "done" if len(result.iters) > 2 else None

    res = _run_plugin(spec, 10, info_hook=hook, modify_params_after_iter=halve, check_convergence=converged)
    assert seen == [0, 1, 2] and res.message == "Reached convergence: done"
    assert len(res.info_vals) == 3 and res.info_vals[2] < res.info_vals[0]


def test_sparse_ingestion_and_ode_propagator_surface():
    """to_sparse: SciPy matrices, QuTiP-4-like objects (sparse .data), dense arrays; the
    DensityMatrixODEPropagator drop-in keeps the reference's constructor (propagators.py:180-205)."""
    import scipy.sparse as sp

    from krotov_amd._ingest import to_sparse
    from krotov_amd.optimize import _use_device_path
    from krotov_amd.propagators import DensityMatrixODEPropagator, HipExpm

    a = np.array([[0, 2j, 0], [0, 0, 0], [1, 0, 3]], dtype=complex)

    class QobjLike:
        def __init__(self, arr):
            self.data = sp.csr_matrix(arr)

        def full(self):
            return self.data.toarray()

    for src in (a, sp.coo_matrix(a), sp.csc_matrix(a), QobjLike(a)):
        m = to_sparse(src)
        assert sp.isspmatrix_csr(m) and m.dtype == np.complex128 and np.array_equal(m.toarray(), a)
    with pytest.raises(ValueError):
        to_sparse(np.zeros((2, 3)))
    p = DensityMatrixODEPropagator(method='bdf', order=5, atol=1e-10, rtol=1e-8, nsteps=10, reentrant=True)
    assert (p.method, p.order, p.atol, p.rtol, p.nsteps, p.reentrant) == ('bdf', 5, 1e-10, 1e-8, 10, True)
    assert p.sparse and p.liouville is True and isinstance(p, HipExpm) and not HipExpm().sparse

    class Obj:
        c_ops = []

    assert _use_device_path(p, None, None, None, 'array', [Obj()])
    assert _use_device_path([p, HipExpm()], None, None, None, 'array', [Obj(), Obj()])
    assert not _use_device_path(lambda *a, **k: None, None, None, None, 'array', [Obj()])


def test_device_path_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    spec = configs.config_c1(nt=20)
    objectives, pulse_options = configs.spec_to_objectives(spec, krotov_amd)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        krotov_amd.optimize_pulses(objectives, pulse_options, spec.tlist,
                                   propagator=krotov_amd.propagators.expm,
                                   chi_constructor=krotov_amd.functionals.chis_ss, iter_stop=1)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        krotov_amd.propagators.expm([spec.H0[0], [spec.Hc[0][0], 0.1]], spec.init[0], 0.01)


def test_bench_starts_its_own_ranks(monkeypatch):
    """``python bench.py --gpus N`` without a launcher (no WORLD_SIZE): bench.self_launch re-executes the same command
    line under torch.distributed.run -- one rank per GPU, rendezvous on 127.0.0.1 with a free port -- and hands the
    launcher's return code on (the driver's multi-GPU command is exactly this form)."""
    import importlib
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    bench = importlib.import_module('bench')
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen['cmd'], seen['env'] = list(cmd), dict(env)
        return Done()

    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3', '--warmup', '1'])
    assert bench.self_launch(4) == 7
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4' and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert 1024 < int(cmd[cmd.index('--master-port') + 1]) < 65536
    tail = cmd[cmd.index(os.path.join(root, 'bench.py')) + 1:]
    assert tail == ['--gpus', '4', '--steps', '3', '--warmup', '1']
    assert seen['env'].get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'


def test_update_sweep_retry_ladder_on_the_host(monkeypatch, caplog):
    """``_HipBackend.iterate`` after KH_ERR_TIMEOUT (the single-launch update sweep's workgroups were not all resident):
    half the workgroups, then an eighth, through ``set_update_workgroups`` -- two rungs at most, each costs a timed-out
    sweep --; only then (or when the engine has no smaller grid) one launch per interval; a reduced grid that got through
    is kept, the full one probed again after ``_FULL_GRID_PROBE_EVERY`` sweeps in a row.  Driven on the
    CPU with the oracle-backed engine double, whose single-launch sweep "times out" above a scripted number of
    workgroups; the pulses must be the oracle's whatever path was taken."""
    import logging

    import krotov_amd
    import krotov_amd.engine as engine_mod
    from helpers import oracle_optimize
    from krotov_amd import _lib, configs
    from oracle_engine_double import OracleEngineDouble

    script = {'fits': 2, 'floor': 1, 'calls': [], 'stepwise': 0}

    class Flaky(OracleEngineDouble):
        grid = None  # None: the engine's own (K workgroups)

        def set_update_workgroups(self, g=0):
            script['calls'].append(g)
            if g == 0:
                self.grid = None
                return self.K
            if g < script['floor']:
                raise _lib.KrotovHipError("no such grid", _lib.KH_ERR_UNSUPPORTED)
            self.grid = g
            return g

        def forward_update(self, *a):
            if (self.grid or self.K) > script['fits']:
                raise _lib.KrotovHipError("timed out", _lib.KH_ERR_TIMEOUT)
            return super().forward_update(*a)

        def forward_update_sharded(self, *a, **kw):
            script['stepwise'] += 1 if kw.get('graph_chunk') == 0 else 0
            return super().forward_update_sharded(*a, **kw)

    monkeypatch.setattr(engine_mod, 'HipKrotovEngine', Flaky)
    caplog.set_level(logging.WARNING, logger='krotov')
    spec = configs.config_c5(K=8, N=4, nt=9, L=1, distinct=True)
    objectives, pulse_options = configs.spec_to_objectives(spec, krotov_amd)
    kw = dict(propagator=krotov_amd.propagators.expm, chi_constructor=krotov_amd.functionals.chis_re, store_all_pulses=True)
    ref = oracle_optimize(spec, 5)

    # 8 workgroups never fit, 2 do: 8 -> 4 (times out) -> 1 (an eighth: two rungs at most) in the first iteration; the
    # grid that got through is kept: no further call, no further timed-out sweep
    res = krotov_amd.optimize_pulses(objectives, pulse_options, spec.tlist, iter_stop=5, **kw)
    assert np.abs(np.array(res.all_pulses) - ref['all_pulses']).max() < 1e-12
    assert script['stepwise'] == 0
    assert script['calls'] == [0, 4, 0, 1]  # (query the full grid, ask for half; query, ask for an eighth)
    assert caplog.text.count('repeating it on 4 workgroups') == 1 and caplog.text.count('repeating it on 1 workgroups') == 1

    # ... and after _FULL_GRID_PROBE_EVERY sweeps in a row on it the full grid gets one more chance (the co-tenant may
    # be gone); the reduced grid of the last time is then asked for directly
    import krotov_amd.optimize as optimize_mod

    monkeypatch.setattr(optimize_mod, '_FULL_GRID_PROBE_EVERY', 2)
    script.update(fits=4, calls=[])
    caplog.clear()
    res = krotov_amd.optimize_pulses(objectives, pulse_options, spec.tlist, iter_stop=5, **kw)
    assert np.abs(np.array(res.all_pulses) - ref['all_pulses']).max() < 1e-12
    # it 1: 8 -> 4 fits; it 2: kept; it 3: probe (reset 0), times out, back to 4 at once; it 4: kept; it 5: probe again
    assert script['calls'] == [0, 4] + [0, 0, 4] * 2
    monkeypatch.setattr(optimize_mod, '_FULL_GRID_PROBE_EVERY', 16)

    # no grid fits and the engine refuses anything below 4: half (times out), an eighth (refused) -> one launch per
    # interval and the engine's own grid again; for good after three such sweeps
    script.update(fits=0, floor=4, calls=[], stepwise=0)
    caplog.clear()
    res = krotov_amd.optimize_pulses(objectives, pulse_options, spec.tlist, iter_stop=5, **kw)
    assert np.abs(np.array(res.all_pulses) - ref['all_pulses']).max() < 1e-12
    assert script['stepwise'] == 5
    assert caplog.text.count('one launch per interval') == 3 and 'staying with that form' in caplog.text
    assert script['calls'] == [0, 4, 0, 1, 0] * 3  # (4 is tried and times out, 1 is refused: reset) -- then never again


def test_bench_leg_traffic_is_keyed_on_the_build_and_scaled_per_interval(tmp_path, monkeypatch):
    """bench.py's ``pmc_traffic_leg``: counters of another build of the kernels are refused, bytes = (2 x FETCH_SIZE +
    WRITE_SIZE) x 1024 (the one convention), scaled from the profiled run's intervals to the leg's; the
    instantiation with the largest traffic stands for a kernel name without template arguments."""
    import json

    import bench

    prof = tmp_path / 'profiles'
    prof.mkdir()
    rec = {'_build': 'B1', '_convention': 'x',
           'K1024': {'_config': {'K': 1024, 'N': 64, 'nt': 501, 'L': 1, 'command': 'perf_sweeps.py 1024 64 501 1'},
                     'kh_ens_forward_update<2, false>': {'FETCH_SIZE': {'avg_per_launch': 1000.0, 'launches': 3},
                                                         'WRITE_SIZE': {'avg_per_launch': 48.0, 'launches': 3}},
                     'kh_ens_forward_update<2, true>': {'FETCH_SIZE': {'avg_per_launch': 10.0, 'launches': 1},
                                                        'WRITE_SIZE': {'avg_per_launch': 1.0, 'launches': 1}},
                     'kh_q2_sweep_store': {'FETCH_SIZE': {'avg_per_launch': 7.0, 'launches': 3}}}}
    (prof / 'pmc_tile_latest.json').write_text(json.dumps(rec))
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    monkeypatch.setattr(bench, 'build_id', lambda: 'B1')
    t, src = bench.pmc_traffic_leg('K1024', 'kh_ens_forward_update', 4000)
    assert t == (2 * 1000.0 + 48.0) * 1024.0 * 4000 / 500 and 'scaled to 4000' in src
    assert bench.pmc_traffic_leg('K1024', 'kh_q2_sweep_store', 4000)[0] is None  # (no WRITE_SIZE pass of that kernel)
    assert bench.pmc_traffic_leg('L4', 'kh_tile_forward_update', 4000)[0] is None
    monkeypatch.setattr(bench, 'build_id', lambda: 'B2')
    t, src = bench.pmc_traffic_leg('K1024', 'kh_ens_forward_update', 4000)
    assert t is None and 'another build' in src
    assert bench.pmc_traffic_leg('config4', 'kh_coop_forward_update', 1000) == (None, 'no profiles/pmc_config4_latest.json')
