"""The reference's own unit tests of the host logic around the path, restated against
``krotov_amd`` with NumPy stand-ins for QuTiP objects (same inputs, same expectations):
tests/test_shapes.py, test_pulse_options.py, test_structural_conversions.py,
test_overlap.py (file:line given per test)."""
import logging
from functools import partial

import os

import numpy as np
import pytest

import krotov_amd
from krotov_amd import conversions, shapes
from krotov_amd.conversions import (
    discretize, extract_controls, extract_controls_mapping, pulse_options_dict_to_list,
)
from krotov_amd.optimize import _initialize_krotov_controls
from krotov_amd.second_order import _overlap
from krotov_amd.shapes import flattop, qutip_callback


class Op:
    """An opaque operator stand-in (the structural conversions never look inside)."""


@pytest.mark.parametrize('func', ['blackman', 'sinsq'])
def test_flattop_basic_properties(func):
    """reference tests/test_shapes.py:8-25"""
    shape = partial(flattop, t_start=10, t_stop=20, t_rise=2, func=func)
    assert shape(9.9) == 0
    assert shape(10) < 1e-14
    assert shape(20) < 1e-14
    assert shape(20.1) == 0
    assert shape(15) == 1


def test_invalid_flattop():
    """reference tests/test_shapes.py:28-31"""
    with pytest.raises(ValueError):
        flattop(0, t_start=10, t_stop=20, t_rise=2, func='xxx')


def _dummy_objective(control):
    H = [Op(), [Op(), control]]
    return [krotov_amd.Objective(initial_state=np.zeros(2, dtype=complex), target=None, H=H)], H[1][1]


def test_shape_validation(dummy_objectives):
    """reference tests/test_pulse_options.py:9-73"""
    objectives, u = _dummy_objective(lambda t, args: 0)
    tlist = np.linspace(0, 10, 100)
    res = _initialize_krotov_controls(objectives, {u: dict(lambda_a=1, update_shape=1)}, tlist)
    shape_arrays, lambda_vals = res[4], res[3]
    assert len(shape_arrays) == 1 and len(shape_arrays[0]) == len(tlist) - 1 and np.all(shape_arrays[0] == 1)
    assert len(lambda_vals) == 1 and lambda_vals[0] == 1 and isinstance(lambda_vals[0], float)
    res = _initialize_krotov_controls(objectives, {u: dict(lambda_a=1, update_shape=0)}, tlist)
    assert np.all(res[4][0] == 0)
    for options, message in (
        (dict(lambda_a=1), "key 'update_shape'"),
        ({'update_shape': 1}, "key 'lambda_a'"),
        (dict(lambda_a=1, update_shape=2), 'update_shape must be a callable'),
        (dict(lambda_a=1, update_shape=lambda t: 2.0), 'in the range [0, 1]'),
        (dict(lambda_a=1, update_shape=lambda t: 0.5j), 'real-valued'),
    ):
        with pytest.raises(ValueError) as exc_info:
            _initialize_krotov_controls(objectives, {u: options}, tlist)
        assert message in str(exc_info.value)


def test_conversion_control_pulse_inverse():
    """reference tests/test_structural_conversions.py:18-32"""
    tlist = np.linspace(0, 10, 20)
    blackman = qutip_callback(shapes.blackman, t_start=0, t_stop=10)
    pulse_orig = conversions.control_onto_interval(discretize(blackman, tlist))
    control = conversions.pulse_onto_tlist(pulse_orig)
    assert np.max(np.abs(conversions.control_onto_interval(control) - pulse_orig)) < 1e-14


def test_discretize():
    """reference tests/test_structural_conversions.py:35-60"""
    tlist = np.linspace(0, 10, 20)
    with pytest.raises(TypeError):
        discretize(partial(shapes.blackman, t_start=0, t_stop=10), tlist)  # not a (t, args) callback
    with pytest.raises(TypeError):
        discretize('sin(t)', tlist)
    control = qutip_callback(shapes.blackman, t_start=0, t_stop=10)
    with pytest.raises(ValueError):
        discretize(np.array([control(t, None) for t in tlist[:-1]]), tlist)
    control_array = discretize(control, tlist)
    assert len(control_array) == len(tlist)
    assert abs(control_array[0]) < 1e-15 and abs(control_array[-1]) < 1e-15
    sampled = np.array([control(t, None) for t in tlist])
    assert np.max(np.abs(sampled - control_array)) < 1e-15
    assert np.max(np.abs(discretize(sampled, tlist) - control_array)) < 1e-15


def test_discretization_as_float(dummy_objectives):
    """reference tests/test_structural_conversions.py:63-82 (an int-valued control is discretised as float)"""
    objectives, u = _dummy_objective(lambda t, args: 0)
    tlist = np.linspace(0, 10, 100)
    res = _initialize_krotov_controls(objectives, {u: dict(lambda_a=1, update_shape=lambda t: 0)}, tlist)
    assert res[0][0].dtype == np.float64 and res[1][0].dtype == np.float64 and res[4][0].dtype == np.float64


def test_initialize_krotov_controls_boundary_conditions(dummy_objectives):
    """reference tests/test_structural_conversions.py:85-141"""
    T = 10
    blackman = qutip_callback(shapes.blackman, t_start=0, t_stop=T)
    objectives = [krotov_amd.Objective(initial_state=np.zeros(2, dtype=complex), target=None,
                                       H=['H0', ['H1', blackman]])]
    tlist = np.linspace(0, T, 10)
    assert abs(blackman(0, None)) < 1e-15 and abs(blackman(T, None)) < 1e-15
    guess_controls, guess_pulses, pulses_mapping, lambda_vals, shape_arrays = _initialize_krotov_controls(
        objectives, {blackman: dict(lambda_a=1.0, update_shape=1)}, tlist)
    assert isinstance(guess_controls[0], np.ndarray) and len(guess_controls[0]) == len(tlist)
    assert abs(guess_controls[0][0]) < 1e-15 and abs(guess_controls[0][-1]) < 1e-15
    assert isinstance(guess_pulses[0], np.ndarray) and len(guess_pulses[0]) == len(tlist) - 1
    assert abs(guess_pulses[0][0]) < 1e-15 and abs(guess_pulses[0][-1]) < 1e-15
    assert pulses_mapping == [[[[1]]]]
    assert lambda_vals == [1.0]
    assert len(shape_arrays) == 1 and isinstance(shape_arrays[0], np.ndarray) and len(shape_arrays[0]) == len(tlist) - 1


def test_extract_controls_with_arrays(dummy_objectives):
    """reference tests/test_structural_conversions.py:144-167"""
    X, Y, Z = Op(), Op(), Op()
    u1, u2 = np.array([]), np.array([])
    psi0, psi_tgt = np.zeros(2), np.ones(2)
    objectives = [
        krotov_amd.Objective(initial_state=psi0, target=psi_tgt, H=[X, [Y, u1], [Z, u2]]),
        krotov_amd.Objective(initial_state=psi0, target=psi_tgt, H=[X, [Y, u2]]),
    ]
    controls = extract_controls(objectives)
    control_map = extract_controls_mapping(objectives, controls)
    assert len(controls) == 2 and controls[0] is u1 and controls[1] is u2
    assert control_map[0] == [[[1], [2]]]
    assert control_map[1] == [[[], [1]]]


def test_extract_controls(dummy_objectives):
    """reference tests/test_structural_conversions.py:170-218"""
    X, Y = Op(), Op()
    f, g, h, d = (lambda t: 0), (lambda t: 0), (lambda t: 0), (lambda t: 0)
    H1, H2, H3 = [X, [X, f], [X, g]], [X, [X, f], [X, h]], [X, [X, d], X]
    objectives = [
        krotov_amd.Objective(initial_state=np.zeros(1), target=Y, H=H1),
        krotov_amd.Objective(initial_state=np.ones(1), target=X, H=H1),
    ]
    controls = extract_controls(objectives)
    assert len(controls) == 2 and f in controls and g in controls
    assert extract_controls_mapping(objectives, controls) == [[[[1], [2]]], [[[1], [2]]]]
    objectives = [
        krotov_amd.Objective(initial_state=np.zeros(1), target=Y, H=H1),
        krotov_amd.Objective(initial_state=np.ones(1), target=X, H=H2),
        krotov_amd.Objective(initial_state=np.ones(1), target=X, H=H3),
    ]
    controls = extract_controls(objectives)
    assert len(controls) == 4 and all(c in controls for c in (f, g, h, d))
    maps = extract_controls_mapping(objectives, controls)
    assert maps[0] == [[[1], [2], [], []]]
    assert maps[1] == [[[1], [], [2], []]]
    assert maps[2] == [[[], [], [], [1]]]


def test_pulse_options_dict_to_list(caplog):
    """reference tests/test_structural_conversions.py:221-254"""
    u1, u2, u3 = np.array([]), np.array([]), np.array([])
    controls = [u1, u2]
    pulse_options = {id(u1): dict(lambda_a=1.0, update_shape=1), id(u2): dict(lambda_a=2.0, update_shape=1)}
    as_list = pulse_options_dict_to_list(pulse_options, controls)
    assert as_list == [pulse_options[id(u1)], pulse_options[id(u2)]]
    with pytest.raises(ValueError) as exc_info:
        pulse_options_dict_to_list({id(u1): dict(lambda_a=1.0, update_shape=1)}, controls)
    assert 'does not have any associated pulse options' in str(exc_info.value)
    pulse_options[id(u3)] = dict(lambda_a=1.0, update_shape=1)
    with caplog.at_level(logging.WARNING):
        pulse_options_dict_to_list(pulse_options, controls)
    assert 'extra elements' in caplog.text


def test_overlap_of_operators():
    """reference tests/test_overlap.py:7-25: tr(Q^dagger rho) for a non-Hermitian Q (the magic basis)."""
    Q = (1.0 / np.sqrt(2.0)) * np.array(
        [[1, 0, 0, 1j], [0, 1j, 1, 0], [0, 1j, -1, 0], [1, 0, 0, -1j]], dtype=np.complex128)
    ket01, ket10 = np.zeros(4, dtype=complex), np.zeros(4, dtype=complex)
    ket01[1], ket10[2] = 1, 1
    rho_2 = np.outer(ket01, ket10.conj())
    expected = complex(np.trace(Q.conj().T @ rho_2))
    assert abs(_overlap(Q, rho_2) - expected) < 1e-14


# ---- tests/test_functionals.py: known answers on two-qubit gates (NumPy kets) -------------------

SQRT_SWAP = np.array([[1, 0, 0, 0], [0, 0.5 + 0.5j, 0.5 - 0.5j, 0], [0, 0.5 - 0.5j, 0.5 + 0.5j, 0], [0, 0, 0, 1]])
CPHASE_PI = np.diag([1, 1, 1, -1]).astype(complex)
SQRT_ISWAP = np.array([[1, 0, 0, 0], [0, 1 / np.sqrt(2), 1j / np.sqrt(2), 0], [0, 1j / np.sqrt(2), 1 / np.sqrt(2), 0],
                       [0, 0, 0, 1]])


def _canonical_basis():
    return [np.eye(4, dtype=complex)[i] for i in range(4)]


def _mapped(gate, basis):
    """|phi_i> = sum_j O_ji |basis_j> (reference functionals.py `mapped_basis`)."""
    return [sum(gate[j, i] * basis[j] for j in range(len(basis))) for i in range(len(basis))]


def _with_weights(objectives):
    import copy

    out = copy.deepcopy(objectives)
    out[1].weight, out[2].weight, out[3].weight = 2.0, 0.5, 0
    return out


def test_f_tau_and_functionals_known_answers():
    """reference tests/test_functionals.py:91-145"""
    from krotov_amd import functionals

    basis = _canonical_basis()
    states = _mapped(SQRT_SWAP, basis)
    objectives = krotov_amd.gate_objectives(basis, CPHASE_PI, [np.zeros((4, 4), dtype=complex)])
    tau = [np.vdot(obj.target, psi) for psi, obj in zip(states, objectives)]
    for got, want in zip(tau, (1 + 0j, 0.5 + 0.5j, 0.5 + 0.5j, -1 + 0j)):
        assert abs(got - want) < 1e-14
    assert abs(functionals.f_tau(states, objectives) - (1 + 1j) / 4) < 1e-14
    weighted = _with_weights(objectives)
    assert abs(functionals.f_tau(states, weighted) - (2.25 + 1.25j) / 4) < 1e-14
    assert all(not hasattr(obj, 'weight') for obj in objectives)  # the originals are untouched
    assert abs(functionals.J_T_ss(states, objectives) - 0.25) < 1e-14
    assert abs(functionals.J_T_sm(states, objectives) - 0.875) < 1e-14
    assert abs(functionals.J_T_re(states, objectives) - 0.75) < 1e-14
    assert abs(functionals.J_T_ss(states, weighted) - 1.75 / 4) < 1e-14


def test_chi_constructors_known_answers():
    """reference tests/test_functionals.py:206-302 (chis_ss, chis_sm, chis_re with and without weights)"""
    from krotov_amd import functionals

    basis = _canonical_basis()
    objectives = krotov_amd.gate_objectives(basis, SQRT_ISWAP, [np.zeros((4, 4), dtype=complex)])
    tau = [1, 0.5 * (1 + 1j), 0.5 * (1 + 1j), 1]

    def check(chis, factors, objs):
        for chi, fac, obj in zip(chis, factors, objs):
            assert np.abs(chi - fac * obj.target).max() < 1e-14

    check(functionals.chis_ss(fw_states_T=basis, objectives=objectives, tau_vals=tau),
          [t / 4 for t in tau], objectives)
    check(functionals.chis_sm(fw_states_T=basis, objectives=objectives, tau_vals=tau), [(3 + 1j) / 16] * 4, objectives)
    check(functionals.chis_re(basis, objectives, None), [1 / 8] * 4, objectives)
    weighted = _with_weights(objectives)
    w = [1.0, 2.0, 0.5, 0.0]
    check(functionals.chis_ss(fw_states_T=basis, objectives=weighted, tau_vals=tau),
          [wk * t / 4 for wk, t in zip(w, tau)], weighted)
    check(functionals.chis_sm(fw_states_T=basis, objectives=weighted, tau_vals=tau),
          [(2.25 + 1.25j) / 16 * wk for wk in w], weighted)
    check(functionals.chis_re(basis, weighted, None), [wk / 8 for wk in w], weighted)
    # the scalars handed to the device-side construction (kh_chi_boundary) give the same co-states
    for fn in (functionals.chis_ss, functionals.chis_sm, functionals.chis_re):
        c, d = functionals.chi_coefficients(fn, np.array(w), tau, 4)
        want = fn(basis, weighted, tau)
        for k in range(4):
            assert np.abs(c[k] * weighted[k].target + d[k] * basis[k] - want[k]).max() < 1e-14


# ---- tests/test_objectives.py: gate_objectives, ensemble_objectives, liouvillian -----------------

SX = np.array([[0, 1], [1, 0]], dtype=complex)
SY = np.array([[0, -1j], [1j, 0]], dtype=complex)
SZ = np.array([[1, 0], [0, -1]], dtype=complex)
SM = np.array([[0, 0], [1, 0]], dtype=complex)  # qutip.sigmam(): |1><0| in QuTiP's basis order
ID2 = np.eye(2, dtype=complex)
CNOT = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=complex)


def test_gate_analysis_known_answers():
    """F_avg, gate, mapped_basis (reference tests/test_functionals.py:79-89, 304-323 and the doctests of
    functionals.py:590-641): sqrt(SWAP) against a controlled phase has F_avg = 0.3, in Hilbert space and from the
    16 propagated dyads."""
    from itertools import product
    basis = [np.eye(4, dtype=complex)[:, i] for i in range(4)]
    CNOT = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=complex)
    states = krotov_amd.functionals.mapped_basis(CNOT, basis)
    assert isinstance(states, tuple) and len(states) == 4
    assert np.array_equal(states[2], basis[3]) and np.array_equal(states[3], basis[2])
    assert np.abs(krotov_amd.functionals.gate(basis, states) - CNOT).max() < 1e-15
    sqrt_swap = np.array([[1, 0, 0, 0], [0, 0.5 + 0.5j, 0.5 - 0.5j, 0], [0, 0.5 - 0.5j, 0.5 + 0.5j, 0], [0, 0, 0, 1]])
    cphase = np.diag([1, 1, 1, -1]).astype(complex)
    fw = krotov_amd.functionals.mapped_basis(sqrt_swap, basis)
    assert abs(krotov_amd.functionals.F_avg(fw_states_T=fw, basis_states=basis, gate=cphase) - 0.3) < 1e-14
    dyads = [np.outer(psi, phi.conj()) for psi, phi in product(fw, fw)]
    assert abs(krotov_amd.functionals.F_avg(fw_states_T=dyads, basis_states=basis, gate=cphase) - 0.3) < 1e-14
    assert abs(krotov_amd.functionals.F_avg(dyads, basis, cphase,
                                            mapped_basis_states=krotov_amd.functionals.mapped_basis(cphase, basis)) - 0.3) < 1e-14
    assert abs(krotov_amd.functionals.F_avg(krotov_amd.functionals.mapped_basis(cphase, basis), basis, cphase) - 1.0) < 1e-14
    with pytest.raises(ValueError, match="Shape of gate"):
        krotov_amd.functionals.F_avg(fw, basis, np.eye(2))
    with pytest.raises(ValueError, match="requires 4 states"):
        krotov_amd.functionals.F_avg(fw[:3], basis, cphase)
    with pytest.raises(ValueError, match="requires 16 states"):
        krotov_amd.functionals.F_avg(dyads[:5], basis, cphase)


def test_invalid_objective_and_adjoint_of_invalid_list():
    """reference tests/test_objectives.py:193-215: what cannot be a state / Hamiltonian is rejected at
    construction (type_checking is on by default); _adjoint of a malformed nested list raises unless told to
    ignore errors."""
    H0 = np.diag([1.0, -1.0]).astype(complex)
    H = [H0, [np.array([[0, 1], [1, 0]], dtype=complex), lambda t, args: 1.0]]
    psi0, psi1 = np.array([1, 0], dtype=complex), np.array([0, 1], dtype=complex)
    krotov_amd.Objective(initial_state=psi0, target=psi1, H=H)
    krotov_amd.Objective(initial_state=psi0, target='PE', H=H0, c_ops=[[H0, lambda t, args: 1.0]])
    for bad in (psi0.conj, None):  # a bound method (psi0.full in the reference's test), nothing
        with pytest.raises(ValueError, match="Invalid initial_state"):
            krotov_amd.Objective(initial_state=bad, target=psi1, H=H)
    for bad in (tuple(H), None):
        with pytest.raises(ValueError, match="Invalid H"):
            krotov_amd.Objective(initial_state=psi0, target=psi1, H=bad)
    with pytest.raises(ValueError, match="Invalid c_ops"):
        krotov_amd.Objective(initial_state=psi0, target=psi1, H=H, c_ops=(H0,))
    nested = ['H0', ['H1', lambda t, args: 1], ['H2', 'H3', lambda t, args: 1]]
    with pytest.raises(ValueError, match="expected format"):
        krotov_amd.objectives._adjoint(nested, ignore_errors=False)
    assert krotov_amd.objectives._adjoint(nested, ignore_errors=True) == nested


def test_gate_objectives_pe_and_midpoints():
    """reference tests/test_objectives.py:350-375 (Bell-basis objectives with target 'PE' under the three
    spellings) and tests/test_structural_conversions.py:257-264 (_tlist_midpoints)."""
    sz, sx, one = np.diag([1.0, -1.0]).astype(complex), np.array([[0, 1], [1, 0]], dtype=complex), np.eye(2, dtype=complex)
    basis = [np.eye(4, dtype=complex)[:, i] for i in range(4)]
    H = [np.kron(sz, one) + np.kron(one, sz), [np.kron(sx, one), lambda t, args: 1.0],
         [np.kron(one, sx), lambda t, args: 1.0]]
    objectives = krotov_amd.gate_objectives(basis, 'PE', H)
    assert len(objectives) == 4 and all(obj.target == 'PE' for obj in objectives)
    b00, b01, b10, b11 = basis
    bell = [(b00 + b11) / np.sqrt(2), (1j * b01 + 1j * b10) / np.sqrt(2), (b01 - b10) / np.sqrt(2),
            (1j * b00 - 1j * b11) / np.sqrt(2)]  # weylchamber.bell_basis
    for obj, state in zip(objectives, bell):
        assert np.abs(obj.initial_state - state).max() < 1e-15
        assert obj == krotov_amd.Objective(initial_state=obj.initial_state, target='PE', H=H)
    for spelling in ('perfect_entangler', 'perfect entangler', 'Perfect Entangler'):
        assert krotov_amd.gate_objectives(basis, spelling, H) == objectives
    with pytest.raises(ValueError):
        krotov_amd.gate_objectives(basis, 'prefect(!) entanglers', H)
    mid = krotov_amd.conversions._tlist_midpoints(np.array([0, 1.0, 2.0, 2.2]))
    assert len(mid) == 3 and mid[0] == 0.5 and mid[1] == 1.5 and mid[2] == 2.1


def test_objective_pickle_with_reduction_function():
    """reference tests/test_objectives.py:698-741: pickled through the reduction function, an objective comes
    back like a deep copy except that its control functions have become placeholders."""
    import copyreg
    import io
    import pickle
    H0 = np.diag([1.0, -1.0]).astype(complex)
    H1 = np.array([[0, 1], [1, 0]], dtype=complex)
    u1, u2 = (lambda t, args: 1.0), (lambda t, args: 2.0)
    C = np.array([[0, 1], [0, 0]], dtype=complex)
    obj1 = krotov_amd.Objective(initial_state=np.array([1, 0], dtype=complex), target=np.array([0, 1], dtype=complex),
                                H=[H0, [H1, u1], [H0, u2]], c_ops=[[C, u1]])
    obj1.weight = 0.5
    with io.BytesIO() as buffer:
        pickler = pickle.Pickler(buffer)
        pickler.dispatch_table = copyreg.dispatch_table.copy()
        pickler.dispatch_table[krotov_amd.Objective] = krotov_amd.objectives._Objective_reduce
        pickler.dump(obj1)
        buffer.seek(0)
        obj2 = pickle.load(buffer)
    assert obj2 is not obj1 and obj2 != obj1
    assert obj2.initial_state is not obj1.initial_state and np.array_equal(obj2.initial_state, obj1.initial_state)
    assert obj2.target is not obj1.target and np.array_equal(obj2.target, obj1.target)
    assert obj2.H[0] is not obj1.H[0] and np.array_equal(obj2.H[0], obj1.H[0])
    assert np.array_equal(obj2.H[1][0], H1) and np.array_equal(obj2.c_ops[0][0], C)
    placeholder = krotov_amd.result.ControlPlaceholder
    assert isinstance(obj2.H[1][1], placeholder) and isinstance(obj2.H[2][1], placeholder)
    assert obj2.H[1][1] != obj2.H[2][1]
    assert isinstance(obj2.c_ops[0][1], placeholder)
    assert obj2.weight == 0.5
    with pytest.raises(Exception):  # lambdas are not picklable without the reduction function
        pickle.dumps(obj1)


def test_derivative_wrt_pulse_cases():
    """reference tests/test_mu.py:54-139: a control appearing in several terms gives the sum of their operators
    (0.5 (s+ + s-) ... here s+ + s- = sigma_x); a control that is not in the objective gives the zero map;
    collapse operators depending on the differentiated control are not implemented, on another control they
    are fine."""
    from krotov_amd.mu import derivative_wrt_pulse
    sp = np.array([[0, 1], [0, 0]], dtype=complex)
    sm, sz, sx = sp.T.copy(), np.diag([1.0, -1.0]).astype(complex), np.array([[0, 1], [1, 0]], dtype=complex)
    eps1, eps2 = (lambda t, args: 0.5), (lambda t, args: 1)
    H1, H2 = [0.5 * sz, [sp, eps1], [sm, eps1]], [0.5 * sz, [sz, eps2]]
    k0, k1 = np.array([1, 0], dtype=complex), np.array([0, 1], dtype=complex)
    controls = [eps1, eps2]

    def system(c_ops):
        objs = [krotov_amd.Objective(initial_state=k0, target=k1, H=H, c_ops=c_ops) for H in (H1, H2)]
        return objs, krotov_amd.conversions.extract_controls_mapping(objs, controls)

    objs, mapping = system([0.1 * sp])
    mu = derivative_wrt_pulse(objs, 0, controls, mapping, i_pulse=0, time_index=0)
    for state in (k0, k1):
        assert np.abs(mu(state) - sx @ state).max() == 0 and mu(state).shape == state.shape
    for i_objective, i_pulse in ((0, 1), (1, 0)):
        zero = derivative_wrt_pulse(objs, i_objective, controls, mapping, i_pulse=i_pulse, time_index=0)
        for state in (k0, k1):
            assert np.abs(zero(state)).max() == 0 and zero(state).shape == state.shape
    objs, mapping = system([[[0.1 * sp, eps1]]])
    with pytest.raises(NotImplementedError):
        derivative_wrt_pulse(objs, 0, controls, mapping, i_pulse=0, time_index=0)
    mu = derivative_wrt_pulse(objs, 1, controls, mapping, i_pulse=1, time_index=0)
    assert np.abs(mu(k1) - sz @ k1).max() == 0


def test_objective_summarize_text():
    """Objective.summarize / str / repr (reference objectives.py:445-578 and its doctests; the scenarios of
    tests/test_objectives.py:559-651): the same text as the reference's own implementation writes for objectives
    of NumPy arrays and of QuTiP-like objects (tests/golden/summarize_cases.txt, made by
    make_reference_goldens.py summarize), the same object always with the same number, copies with new ones."""
    import copy
    import importlib.util
    golden_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    spec = importlib.util.spec_from_file_location('make_goldens', os.path.join(golden_dir, 'make_reference_goldens.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    class FakeQ:  # what the summary looks at in a qutip.Qobj
        def __init__(self, kind, dims, isherm):
            self.type, self.dims, self.isherm = kind, dims, isherm
            self.shape = (4, 1 if kind == 'ket' else 4)

    got = mod.summarize_cases(krotov_amd.Objective, FakeQ)
    want = open(os.path.join(golden_dir, 'summarize_cases.txt'), encoding='utf8').read().splitlines()
    assert got == want
    obj = krotov_amd.Objective(initial_state=np.zeros(2), target=np.ones(2), H=[np.eye(2), [np.eye(2), lambda t, a: 0]])
    first = obj.summarize(reset_symbol_counters=True)
    assert obj.summarize() == first  # same objects, same numbers
    assert copy.deepcopy(obj).summarize() != first  # new objects, new numbers
    assert copy.deepcopy(obj).summarize(reset_symbol_counters=True) == first
    krotov_amd.Objective.reset_symbol_counters()


def test_gate_objectives_single_qubit_gate():
    """reference tests/test_objectives.py:307-316"""
    basis = [np.array([1, 0], dtype=complex), np.array([0, 1], dtype=complex)]
    H = [SZ, [SX, lambda t, args: 1.0]]
    objectives = krotov_amd.gate_objectives(basis, SY, H)  # sigma_y = -i|0><1| + i|1><0|
    assert len(objectives) == 2
    assert objectives[0].initial_state is basis[0] and np.array_equal(objectives[0].target, 1j * basis[1])
    assert objectives[1].initial_state is basis[1] and np.array_equal(objectives[1].target, -1j * basis[0])
    assert objectives[0].H[1][1] is H[1][1]  # the control object is shared


def test_gate_objectives_shape_error():
    """reference tests/test_objectives.py:319-330"""
    basis = [np.array([1, 0], dtype=complex), np.array([0, 1], dtype=complex)]
    with pytest.raises(ValueError) as exc_info:
        krotov_amd.gate_objectives(basis, np.kron(SY, ID2), [SZ, [SX, lambda t, args: 1.0]])
    assert "same dimension as the number of basis" in str(exc_info.value)


def test_ensemble_objectives():
    """reference tests/test_objectives.py:333-347"""
    rng = np.random.default_rng(1)
    H0, H1 = rng.standard_normal((3, 3)) + 0j, rng.standard_normal((3, 3)) + 0j
    eps = lambda t, args: 1.0  # noqa: E731
    psi0, psi1 = np.array([1, 0, 0], dtype=complex), np.array([0, 1, 0], dtype=complex)
    H = [H0, [H1, eps]]
    objectives = [krotov_amd.Objective(initial_state=psi0, target=psi1, H=H),
                  krotov_amd.Objective(initial_state=psi1, target=psi0, H=H)]
    Hs = [[H0, [mu * H1, eps]] for mu in [0.95, 0.99, 1.01, 1.05]]
    ens = krotov_amd.ensemble_objectives(objectives, Hs)
    assert len(ens) == 10
    assert ens[0] == objectives[0] and ens[1] == objectives[1]
    assert np.abs(ens[2].H[1][0] - 0.95 * H1).max() < 1e-15
    assert np.abs(ens[9].H[1][0] - 1.05 * H1).max() < 1e-15


def _two_qubit_liouvillian():
    H = [np.kron(SZ, ID2) + np.kron(ID2, SZ), [np.kron(SX, ID2), lambda t, args: 1.0],
         [np.kron(ID2, SX), lambda t, args: 1.0]]
    c_ops = [np.kron(SM, ID2), np.kron(ID2, SM)]
    return H, c_ops, krotov_amd.objectives.liouvillian(H, c_ops)


def test_liouvillian():
    """reference tests/test_objectives.py:378-402: d/dt vec(rho) = L vec(rho), column-stacked vec."""
    H, c_ops, L = _two_qubit_liouvillian()
    assert isinstance(L, list) and len(L) == 3
    assert L[1][1] is H[1][1] and L[2][1] is H[2][1]
    rng = np.random.default_rng(2)
    rho = rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4))

    def lindblad(Hm, cs):
        out = -1j * (Hm @ rho - rho @ Hm)
        for c in cs:
            cd = c.conj().T
            out = out + c @ rho @ cd - 0.5 * (cd @ c @ rho + rho @ cd @ c)
        return out

    vec = rho.ravel(order='F')
    assert np.abs((L[0] @ vec).reshape(4, 4, order='F') - lindblad(H[0], c_ops)).max() < 1e-14
    assert np.abs((L[1][0] @ vec).reshape(4, 4, order='F') - lindblad(H[1][0], [])).max() < 1e-14
    assert np.abs(krotov_amd.objectives.liouvillian(H[0], c_ops) - L[0]).max() < 1e-15
    with pytest.raises(ValueError):
        krotov_amd.objectives.liouvillian(tuple(H), c_ops)


def test_gate_objectives_liouville_state_sets():
    """reference tests/test_objectives.py:416-540 ('3states', 'd+1', 'full' with a CNOT)"""
    _, _, L = _two_qubit_liouvillian()
    basis = [np.eye(4, dtype=complex)[i] for i in range(4)]

    def conj_by_gate(rho):
        return CNOT @ rho @ CNOT.conj().T

    objs = krotov_amd.gate_objectives(basis, CNOT, L, liouville_states_set='3states')
    rho_1 = np.diag([0.1 * (4 - i) for i in range(4)]).astype(complex)
    rho_2 = np.full((4, 4), 1 / 4, dtype=complex)
    rho_3 = np.diag([1 / 4] * 4).astype(complex)
    assert len(objs) == 3
    for obj, rho in zip(objs, (rho_1, rho_2, rho_3)):
        assert np.abs(obj.initial_state - rho).max() < 1e-14
        assert np.abs(obj.target - conj_by_gate(rho)).max() < 1e-14
        assert not hasattr(obj, 'weight')
    objs = krotov_amd.gate_objectives(basis, CNOT, L, liouville_states_set='3states', weights=[1, 0, 2])
    assert len(objs) == 2
    assert np.abs(objs[0].initial_state - rho_1).max() < 1e-14 and np.abs(objs[1].initial_state - rho_3).max() < 1e-14
    assert all(isinstance(o.weight, float) for o in objs) and objs[0].weight == 1 and objs[1].weight == 2
    for bad in ([1, 2], [1, 1, -1]):
        with pytest.raises(ValueError):
            krotov_amd.gate_objectives(basis, CNOT, L, liouville_states_set='3states', weights=bad)

    objs = krotov_amd.gate_objectives(basis, CNOT, L, liouville_states_set='d+1')
    assert len(objs) == 5
    rhos = [np.outer(b, b.conj()) for b in basis] + [rho_2]
    for obj, rho in zip(objs, rhos):
        assert np.abs(obj.initial_state - rho).max() < 1e-14
        assert np.abs(obj.target - conj_by_gate(rho)).max() < 1e-14

    objs = krotov_amd.gate_objectives(basis, CNOT, L, liouville_states_set='full')
    assert len(objs) == 16
    rhos = [np.outer(basis[i], basis[j].conj()) for i in range(4) for j in range(4)]
    for obj, rho in zip(objs, rhos):
        assert np.abs(obj.initial_state - rho).max() < 1e-14
        assert np.abs(obj.target - conj_by_gate(rho)).max() < 1e-14


# ---- tests/test_infohooks.py:15-72: chained hooks, modify_params_after_iter, shared_data ----------

def test_infohook_chaining():
    """Return values of several info_hooks combine into a tuple, None (from
    modify_params_after_iter) is ignored, shared_data is passed along the chain and cleared
    every iteration; known answer F_re = 0.001978333994757067 after one iteration."""
    from helpers import check_infohook_chaining, numpy_plugins

    prop, mu, overlap = numpy_plugins()
    check_infohook_chaining(propagator=prop, mu=mu, overlap=overlap, norm=np.linalg.norm)
