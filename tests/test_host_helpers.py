"""Host helpers around the path: ``convergence``,
``info_hooks.print_table``, ``Result.dump/load`` -- the reference's big
integration test (tests/test_krotov.py:202-445, ``test_continue_optimization``)
restated on the plugin path with NumPy plugins, against the reference's own
log file tests/test_krotov/oct.log (kept as tests/golden/oct.log)."""
import functools
import io
import logging
import os

import numpy as np
import pytest

import krotov_amd
from krotov_amd import convergence, shapes
from krotov_amd.result import Result

from helpers import numpy_plugins

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _system():
    """reference tests/test_krotov.py:136-163 (notebook 01's two-level system, constant guess)."""
    H0 = -0.5 * np.array([[1, 0], [0, -1]], dtype=complex)
    H1 = np.array([[0, 1], [1, 0]], dtype=complex)
    H = [H0, [H1, lambda t, args: 0.2]]
    psi0, psi1 = np.array([1, 0], dtype=complex), np.array([0, 1], dtype=complex)
    objectives = [krotov_amd.Objective(initial_state=psi0, target=psi1, H=H)]

    def S(t):
        return shapes.flattop(t, t_start=0, t_stop=5, t_rise=0.3, t_fall=0.3, func='sinsq')

    return objectives, {H[1][1]: dict(lambda_a=5, update_shape=S)}, np.linspace(0, 5, 500)


def test_continue_optimization_log_and_dumps(tmp_path, caplog):
    objectives, pulse_options, tlist = _system()
    prop, mu, vdot = numpy_plugins()

    def overlap(a, b):  # like the default overlap: None when there is no state yet
        return None if a is None or b is None else vdot(a, b)

    dumpfile = str(tmp_path / "oct_result_{iter:03d}.dump")
    log = io.StringIO()

    def run(**kw):
        return krotov_amd.optimize_pulses(
            objectives, pulse_options=pulse_options, tlist=tlist, propagator=prop, mu=mu, overlap=overlap,
            norm=np.linalg.norm, chi_constructor=krotov_amd.functionals.chis_re, store_all_pulses=True,
            info_hook=krotov_amd.info_hooks.print_table(J_T=krotov_amd.functionals.J_T_re, out=log),
            check_convergence=convergence.Or(
                convergence.check_monotonic_error, convergence.dump_result(dumpfile, every=2)),
            **kw)

    with caplog.at_level(logging.WARNING):
        r1 = run(iter_stop=3, skip_initial_forward_propagation=True)
    assert "You should not use `skip_initial_forward_propagation`" in caplog.text
    assert len(r1.iters) == len(r1.iter_seconds) == len(r1.info_vals) == len(r1.all_pulses) == 4
    assert len(r1.states) == 1 and len(r1.guess_controls) == len(r1.optimized_controls) == 1
    assert len(r1.guess_controls[0]) == len(r1.optimized_controls[0]) == len(r1.tlist)
    assert all(len(p) == len(tlist) - 1 for pulses in r1.all_pulses for p in pulses)
    assert "3 iterations" in r1.message
    run(continue_from=r1, iter_stop=3)  # only propagates the guess pulse
    r2 = run(continue_from=r1, iter_stop=5)
    assert len(r2.iters) == len(r2.info_vals) == len(r2.all_pulses) == 6 and "5 iterations" in r2.message
    r3 = run(continue_from=r2, iter_stop=7, skip_initial_forward_propagation=True)
    assert len(r3.iters) == len(r3.info_vals) == len(r3.all_pulses) == 8 and "7 iterations" in r3.message
    r4 = run(continue_from=r3, iter_stop=5, skip_initial_forward_propagation=True)  # no-op
    assert r4.iters == r3.iters and r4.message == r3.message
    assert r4.start_local_time_str == r3.start_local_time_str

    # the combined log vs the reference's (the seconds column, beyond character 63, differs)
    got = log.getvalue().splitlines()
    want = open(os.path.join(GOLDEN, 'oct.log'), encoding='utf8').read().splitlines()
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a[:63] == b[:63]

    # continuing from an incomplete dump
    caplog.clear()
    with caplog.at_level(logging.WARNING):
        loaded = Result.load(str(tmp_path / "oct_result_004.dump"))
    assert 'Result.objectives contains control placeholders' in caplog.text
    plain = dict(propagator=prop, mu=mu, overlap=overlap, norm=np.linalg.norm,
                 chi_constructor=krotov_amd.functionals.chis_re, store_all_pulses=True)
    with pytest.raises(ValueError, match="objectives must remain unchanged"):
        krotov_amd.optimize_pulses(objectives, pulse_options, tlist, continue_from=loaded, iter_stop=7,
                                   skip_initial_forward_propagation=True, **plain)
    loaded = Result.load(str(tmp_path / "oct_result_004.dump"), objectives=objectives)
    assert loaded.iters[-1] == 4
    r5 = krotov_amd.optimize_pulses(objectives, pulse_options, tlist, continue_from=loaded, iter_stop=7,
                                    info_hook=krotov_amd.functionals.J_T_re,
                                    skip_initial_forward_propagation=True, **plain)
    assert r5.iters == r3.iters and len(r5.info_vals) == 8
    assert abs(r5.info_vals[-1] - r3.info_vals[-1]) < 1e-10  # reference tests/test_krotov.py:426-432
    assert np.abs(r5.optimized_controls[0] - r3.optimized_controls[0]).max() < 1e-10

    # dumps taken mid-optimisation hold pulses on the intervals: load(finalize=True) maps them onto tlist
    caplog.clear()
    with caplog.at_level(logging.WARNING):
        raw = Result.load(str(tmp_path / "oct_result_004.dump"), objectives=objectives)
    assert 'not finalized' in caplog.text and len(raw.optimized_controls[0]) == len(tlist) - 1
    fin = Result.load(str(tmp_path / "oct_result_004.dump"), objectives=objectives, finalize=True)
    assert len(fin.optimized_controls[0]) == len(tlist)
    opt_objs = r3.optimized_objectives
    assert opt_objs[0].H[1][1] is r3.optimized_controls[0] and objectives[0].H[1][1] is not opt_objs[0].H[1][1]
    with pytest.raises(ValueError, match="Expected 1 controls"):
        r3.objectives_with_controls([])
    with pytest.raises(ValueError, match="time grid"):
        r3.objectives_with_controls([np.zeros(3)])


def test_convergence_checks():
    """Doctest values of reference convergence.py:140-157, 236-262, 316-367."""
    r = Result()
    check = convergence.value_below(limit='1e-4', spec=lambda res: res.info_vals[-1], name='J_T')
    r.info_vals.append(1e-4)
    assert check(r) is None
    r.info_vals.append(9e-5)
    assert check(r) == 'J_T < 1e-4'
    assert convergence.value_above('0.99', name='F')(r) is None
    r.info_vals.append(0.999)
    assert convergence.value_above('0.99', name='F')(r) == 'F > 0.99'
    r = Result()
    delta = convergence.delta_below(limit='1e-4', name='ΔJ_T')
    r.info_vals.append(9e-1)
    assert delta(r) is None  # only one value yet
    r.info_vals.append(1e-1)
    assert delta(r) is None
    r.info_vals.append(4e-4)
    assert delta(r) is None
    r.info_vals.append(3.5e-4)
    assert delta(r) == 'ΔJ_T < 1e-4'
    with pytest.raises(IndexError):
        delta(Result())  # neither value exists
    r = Result()
    for v, want in ((9e-1, None), (1e-1, None), (2e-1, 'Loss of monotonic convergence; error decrease < 0')):
        r.info_vals.append(v)
        assert convergence.check_monotonic_error(r) == want
    r = Result()
    for v, want in ((0.0, None), (0.2, None), (0.15, 'Loss of monotonic convergence; fidelity increase < 0')):
        r.info_vals.append(v)
        assert convergence.check_monotonic_fidelity(r) == want
    either = convergence.Or(convergence.value_below(0.1), convergence.check_monotonic_fidelity)
    assert either(r) == 'Loss of monotonic convergence; fidelity increase < 0'
    with pytest.raises(ValueError):
        convergence.dump_result('x.dump', every=0)
    r.iters.append(2)
    assert convergence.dump_result('/nonexistent-dir/x_{iter}.dump', every=2)(r).startswith('Could not store')


def test_print_table_layout():
    """Several pulses with per-pulse columns, ASCII headers, a wide iteration column, custom
    formats and headers, the monotonicity flags: character for character what the reference's
    print_table writes (tests/golden/print_table_cases.txt, made by make_reference_goldens.py)."""
    out = io.StringIO()
    J = lambda **kw: kw['J']  # noqa: E731
    common = dict(guess_pulses=[None, None], iter_stop=10, start_time=0.0, stop_time=2.4)
    hook = krotov_amd.info_hooks.print_table(J_T=J, show_g_a_int_per_pulse=True, out=out)
    assert hook(iteration=0, J=1.0, g_a_integrals=np.zeros(2), info_vals=[], **common) == 1.0
    hook(iteration=1, J=0.5, g_a_integrals=np.array([0.1, 0.2]), info_vals=[1.0], **common)
    hook(iteration=2, J=0.6, g_a_integrals=np.array([0.0, 0.05]), info_vals=[1.0, 0.5], **common)
    out.write("--\n")
    hook = krotov_amd.info_hooks.print_table(J_T=J, unicode=False, out=out)
    one = dict(guess_pulses=[None], iter_stop=12345, start_time=0.0, stop_time=0.0)
    hook(iteration=0, J=1.0, g_a_integrals=np.zeros(1), info_vals=[], **one)
    hook(iteration=1, J=0.25, g_a_integrals=np.array([0.5]), info_vals=[1.0], **one)
    out.write("--\n")
    hook = krotov_amd.info_hooks.print_table(
        J_T=J, show_g_a_int_per_pulse=True, out=out,
        col_formats=('%03d', '%.6f', '%.3e', '%.3e', '%.6f', '%+.1e', '%+.1e', '%4d'),
        col_headers=('#', 'error', 'ga[{l}]', 'ga', 'total', 'd(error)', 'd(total)', 's'))
    hook(iteration=0, J=1.0, g_a_integrals=np.zeros(2), info_vals=[], **common)
    hook(iteration=1, J=0.5, g_a_integrals=np.array([0.1, 0.2]), info_vals=[1.0], **common)
    want = open(os.path.join(GOLDEN, 'print_table_cases.txt'), encoding='utf8').read()
    assert out.getvalue() == want
    with pytest.raises(ValueError, match="exactly 8"):
        krotov_amd.info_hooks.print_table(J_T=None, col_formats=('%d',))
    with pytest.raises(ValueError, match="format"):
        krotov_amd.info_hooks.print_table(J_T=None, col_headers=("i", "J", 3, "g", "J", "dJT", "dJ", "s"))
    with pytest.raises(ValueError, match="Invalid col_formats"):
        krotov_amd.info_hooks.print_table(J_T=None, col_formats=('%d', '%.2e', '%.2e', '%.2e %d', '%.2e', '%.2e',
                                                                 '%.2e', '%d'))


def test_print_debug_information_text():
    """The full-signature info_hook writes what the reference's print_debug_information writes for the same
    keyword arguments (tests/golden/print_debug_cases.txt, made by make_reference_goldens.py print_debug; the
    wall-clock start is masked: it is printed in local time), and runs as info_hook of a real optimization."""
    import re
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_goldens', os.path.join(GOLDEN, 'make_reference_goldens.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # (only defines functions; the reference is imported inside them)
    out = io.StringIO()
    for kw in mod.debug_information_cases():
        krotov_amd.info_hooks.print_debug_information(out=out, **kw)
        out.write("--\n")
    mask = lambda text: re.sub(r'started at [0-9: -]+', 'started at X', text)  # noqa: E731
    want = open(os.path.join(GOLDEN, 'print_debug_cases.txt'), encoding='utf8').read()
    assert mask(out.getvalue()) == mask(want)
    # final states given as arrays: sizes and norms come from the arrays
    kw = dict(mod.debug_information_cases()[1], fw_states_T=[np.array([0.6, 0.8j]), np.array([1.0, 0.0])])
    out = io.StringIO()
    krotov_amd.info_hooks.print_debug_information(out=out, **kw)
    assert '    fw_states_T norm: 1.000000, 1.000000\n' in out.getvalue()
    assert '[2 * ndarray(5)] (0.0 MB), None, None' in out.getvalue()
    # as the info_hook of an optimization
    objectives, pulse_options, tlist = _system()
    prop, mu, vdot = numpy_plugins()
    log = io.StringIO()
    krotov_amd.optimize_pulses(
        objectives, pulse_options=pulse_options, tlist=tlist, propagator=prop, mu=mu,
        overlap=lambda a, b: None if a is None or b is None else vdot(a, b), norm=np.linalg.norm,
        chi_constructor=krotov_amd.functionals.chis_re,
        info_hook=functools.partial(krotov_amd.info_hooks.print_debug_information, out=log), iter_stop=1)
    text = log.getvalue()
    assert text.startswith('Iteration 0\n    objectives:\n') and '\nIteration 1\n' in text
    assert 'chi_constructor: chis_re' in text and 'storage (bw, fw, fw0): [1 * ' in text


def test_objective_propagate_host_loop():
    """Objective.propagate (reference objectives.py:338-433; scenario of tests/test_objectives.py:233-270 without
    the mesolve half): states on every grid point under the optimizer's own pulse discretization, expectation
    values instead of states with e_ops, rho0 / H overrides, the attributes of the returned record."""
    import scipy.linalg
    H0 = np.diag([-0.5, 0.5]).astype(complex)
    H1 = np.array([[0, 1], [1, 0]], dtype=complex)
    eps = lambda t, args: 0.3 * np.sin(t) + args.get('offset', 0.0)  # noqa: E731
    psi0, psi1 = np.array([1, 0], dtype=complex), np.array([0, 1], dtype=complex)
    obj = krotov_amd.Objective(initial_state=psi0, target=psi1, H=[H0, [H1, eps]])
    tlist = np.linspace(0, 5, 51)
    prop, _, _ = numpy_plugins()
    res = obj.propagate(tlist, propagator=prop)
    assert res.solver == prop.__name__ and res.num_expect == 0 and res.num_collapse == 0
    assert np.array_equal(res.times, tlist) and len(res.states) == len(tlist) and res.expect == []
    pulse = krotov_amd.conversions.control_onto_interval(krotov_amd.conversions.discretize(eps, tlist, args=({},)))
    state = psi0
    for n in range(len(tlist) - 1):
        state = scipy.linalg.expm(-1j * (H0 + pulse[n] * H1) * (tlist[n + 1] - tlist[n])) @ state
        assert np.abs(res.states[n + 1] - state).max() < 1e-13
    P0, P1 = np.diag([1.0, 0.0]).astype(complex), np.diag([0.0, 1.0]).astype(complex)
    res2 = obj.propagate(tlist, propagator=prop, e_ops=[P0, P1, H1 @ P0])
    assert len(res2.states) == 0 and len(res2.expect) == 3 and res2.num_expect == 3
    assert all(len(e) == len(tlist) for e in res2.expect)
    assert res2.expect[0].dtype == np.float64 and np.iscomplexobj(res2.expect[2])  # Hermitian -> real
    assert np.abs(res2.expect[0] - np.array([abs(s[0]) ** 2 for s in res.states])).max() < 1e-14
    assert np.abs(res2.expect[0] + res2.expect[1] - 1.0).max() < 1e-13
    # other initial state, other Hamiltonian, control arguments, a custom expect
    res3 = obj.propagate(tlist, propagator=prop, rho0=psi1, H=[H0, [2 * H1, eps]], args={'offset': 0.1},
                         e_ops=[P1], expect=lambda op, st: 7.0)
    assert np.all(res3.expect[0] == 7.0)
    res4 = obj.propagate(tlist, propagator=prop, rho0=psi1, H=[H0, [2 * H1, eps]], args={'offset': 0.1})
    pulse4 = krotov_amd.conversions.control_onto_interval(
        krotov_amd.conversions.discretize(eps, tlist, args=({'offset': 0.1},)))
    state = psi1
    for n in range(len(tlist) - 1):
        state = scipy.linalg.expm(-1j * (H0 + pulse4[n] * 2 * H1) * (tlist[n + 1] - tlist[n])) @ state
    assert np.abs(res4.states[-1] - state).max() < 1e-13


def test_reference_parallel_map_names():
    """Scripts that pass the reference's process-pool maps (tests/test_parallelization.py:113-140) or call
    set_parallelization keep running: the names exist and map serially; |tau| after one iteration of the
    transmon X gate is the reference's 0.9693 / 0.7743 (checked on the oracle in test_oracle_golden.py) --
    here only that the three-map form goes through the plugin loop and gives the serial result."""
    par = krotov_amd.parallelization
    par.set_parallelization(use_loky=False, start_method='fork')
    with pytest.raises(ValueError, match="start_method"):
        par.set_parallelization(start_method='loky')
    assert par.parallel_map(lambda v, a: v * a, [1, 2, 3], (2,), num_cpus=4) == [2, 4, 6]
    assert par.parallel_map_fw_prop_step(lambda v, a: v + a, range(3), (1,)) == [1, 2, 3]
    objectives, pulse_options, tlist = _system()
    prop, mu, vdot = numpy_plugins()
    kw = dict(pulse_options=pulse_options, tlist=tlist, propagator=prop, mu=mu, norm=np.linalg.norm,
              overlap=lambda a, b: None if a is None or b is None else vdot(a, b),
              chi_constructor=krotov_amd.functionals.chis_re, iter_stop=1)
    serial = krotov_amd.optimize_pulses(objectives, **kw)
    mapped = krotov_amd.optimize_pulses(
        objectives, parallel_map=(par.parallel_map, par.parallel_map, par.parallel_map_fw_prop_step), **kw)
    assert np.array_equal(serial.optimized_controls[0], mapped.optimized_controls[0])


def test_load_reference_dump_and_continue(caplog):
    """Result.load reads a dump written by the reference itself (tests/golden/reference_tls_oct_result.dump is
    the reference's tests/test_result_serialization/oct_result.dump: QuTiP 4 objects inside) without krotov or
    qutip: fields as extracted independently into dump_tls_ss.npz, QuTiP objects as arrays; and the
    optimisation continues from it exactly as if it had run here from the start."""
    from krotov_amd import configs
    path = os.path.join(GOLDEN, 'reference_tls_oct_result.dump')
    g = np.load(os.path.join(GOLDEN, 'dump_tls_ss.npz'))
    with caplog.at_level('WARNING', logger='krotov'):
        res = krotov_amd.result.Result.load(path)
    assert 'control placeholders' in caplog.text
    assert isinstance(res, krotov_amd.result.Result) and res.message.startswith('Reached convergence')
    assert np.array_equal(res.tlist, g['tlist']) and list(res.iters) == list(g['iters'])
    assert np.array_equal(np.array(res.tau_vals), g['tau_vals'])
    assert np.array_equal(np.array(res.optimized_controls), g['optimized_controls'])
    assert np.array_equal(np.array(res.all_pulses)[:, 0, :], g['all_pulses'][:, 0, :])
    obj = res.objectives[0]
    spec = configs.config_c1()
    assert isinstance(obj, krotov_amd.Objective)
    assert np.array_equal(obj.H[0], spec.H0[0]) and np.array_equal(obj.H[1][0], spec.Hc[0][0])
    assert isinstance(obj.H[1][1], krotov_amd.result.ControlPlaceholder)
    assert np.array_equal(obj.initial_state, spec.init[0]) and np.array_equal(obj.target, spec.target[0])
    assert all(isinstance(st, np.ndarray) and st.shape == (2,) for st in res.states)
    # continue two iterations from the reference's result vs 20 iterations from scratch
    objectives, pulse_options = configs.spec_to_objectives(spec, krotov_amd, column_states=False)
    prop, mu, vdot = numpy_plugins()
    kw = dict(pulse_options=pulse_options, tlist=spec.tlist, propagator=prop, mu=mu, norm=np.linalg.norm,
              overlap=lambda a, b: None if a is None or b is None else vdot(a, b),
              chi_constructor=krotov_amd.functionals.chis_ss, store_all_pulses=True)
    loaded = krotov_amd.result.Result.load(path, objectives=objectives)
    cont = krotov_amd.optimize_pulses(objectives, continue_from=loaded, iter_stop=20, **kw)
    scratch = krotov_amd.optimize_pulses(objectives, iter_stop=20, **kw)
    assert list(cont.iters) == list(range(21)) and len(cont.all_pulses) == 21
    assert np.abs(np.array(cont.all_pulses[19:]) - np.array(scratch.all_pulses[19:])).max() < 1e-9
    assert np.abs(np.array(cont.tau_vals[19:]) - np.array(scratch.tau_vals[19:])).max() < 1e-9
    assert np.abs(np.array(cont.optimized_controls) - np.array(scratch.optimized_controls)).max() < 1e-9


@pytest.mark.parametrize('iter_stop', [0, -1])
def test_zero_iterations(iter_stop):
    """reference tests/test_krotov.py:166-199"""
    objectives, pulse_options, tlist = _system()
    prop, mu, vdot = numpy_plugins()
    log = io.StringIO()
    result = krotov_amd.optimize_pulses(
        objectives, pulse_options=pulse_options, tlist=tlist, propagator=prop, mu=mu,
        overlap=lambda a, b: None if a is None or b is None else vdot(a, b), norm=np.linalg.norm,
        chi_constructor=krotov_amd.functionals.chis_re, store_all_pulses=True,
        info_hook=krotov_amd.info_hooks.print_table(J_T=krotov_amd.functionals.J_T_re, out=log),
        iter_stop=iter_stop, skip_initial_forward_propagation=True)
    assert len(log.getvalue().splitlines()) == 2
    assert result.message == 'Reached 0 iterations'
    assert len(result.guess_controls) == len(result.optimized_controls) == 1
    assert len(result.guess_controls[0]) == len(result.optimized_controls[0]) == len(result.tlist)
    assert all(np.all(c1 == c2) for c1, c2 in zip(result.guess_controls, result.optimized_controls))
    assert all(len(p) == len(result.tlist) - 1 for pulses in result.all_pulses for p in pulses)


def test_broken_continuations():
    """reference tests/test_krotov.py:433-540: every way `continue_from` can be inconsistent."""
    from copy import deepcopy

    objectives, pulse_options, tlist = _system()
    prop, mu, overlap = numpy_plugins()
    kw = dict(pulse_options=pulse_options, tlist=tlist, propagator=prop, mu=mu, overlap=overlap, norm=np.linalg.norm,
              chi_constructor=krotov_amd.functionals.chis_re)
    result = krotov_amd.optimize_pulses(objectives, iter_stop=1, store_all_pulses=True, **kw)

    def broken(res, message, **extra):
        with pytest.raises(ValueError) as exc_info:
            krotov_amd.optimize_pulses(objectives, continue_from=res, **{**kw, **extra})
        assert message in str(exc_info.value)

    extra_obj = deepcopy(result)
    extra_obj.objectives.append(deepcopy(result.objectives[0]))
    broken(extra_obj, "number of objectives must be the same", store_all_pulses=True)
    broken(result, "store_all_pulses parameter cannot be changed", store_all_pulses=False)
    scaled = deepcopy(result)
    scaled.objectives = result.objectives
    scaled.tlist = scaled.tlist * 2
    broken(scaled, "same time grid", store_all_pulses=True)
    changed_nt = deepcopy(result)
    changed_nt.objectives = result.objectives
    changed_nt.tlist = np.linspace(0, 5, 1000)
    broken(changed_nt, "same time grid", store_all_pulses=True)
    incongruent = deepcopy(result)
    incongruent.objectives = result.objectives
    incongruent.optimized_controls[0] = np.stack([result.optimized_controls[0]] * 2).flatten()
    broken(incongruent, "optimized_controls and tlist are incongruent", store_all_pulses=True)
    broken(result.tlist, "only possible from a Result object", store_all_pulses=True)


def test_numpy_array_controls_and_id_keyed_pulse_options():
    """reference tests/test_numpy_controls.py (issue #79): the guess control given as a NumPy array,
    pulse_options keyed by id(control); a few iterations reproduce the run with the callable control."""
    tlist = np.linspace(0, 5, 500)
    H0 = -0.5 * np.array([[1, 0], [0, -1]], dtype=complex)
    H1 = np.array([[0, 1], [1, 0]], dtype=complex)

    def guess_control(t, args):
        return 0.2 * shapes.flattop(t, t_start=0, t_stop=5, t_rise=0.3, func="blackman")

    def S(t):
        return shapes.flattop(t, t_start=0, t_stop=5, t_rise=0.3, t_fall=0.3, func='blackman')

    psi0, psi1 = np.array([1, 0], dtype=complex), np.array([0, 1], dtype=complex)
    guess_array = np.array([guess_control(t, []) for t in tlist])
    H = [H0, [H1, guess_array]]
    objectives = [krotov_amd.Objective(initial_state=psi0, target=psi1, H=H)]
    prop, mu, vdot = numpy_plugins()
    kw = dict(tlist=tlist, propagator=prop, mu=mu, norm=np.linalg.norm,
              overlap=lambda a, b: None if a is None or b is None else vdot(a, b),
              chi_constructor=krotov_amd.functionals.chis_ss)
    res = krotov_amd.optimize_pulses(
        objectives, pulse_options={id(H[1][1]): dict(lambda_a=5, update_shape=S)},
        check_convergence=convergence.Or(convergence.value_below('1e-3', name='J_T'),
                                         convergence.check_monotonic_error),
        iter_stop=0, skip_initial_forward_propagation=True, **kw)
    assert res.message == 'Reached 0 iterations'
    # (beyond the reference's test) the array control optimises like the function it was sampled from:
    # arrays skip the mid-point sampling (conversions.py:126-133), so compare with that array's own run
    res_arr = krotov_amd.optimize_pulses(
        objectives, pulse_options={id(H[1][1]): dict(lambda_a=5, update_shape=S)}, iter_stop=2, **kw)
    g = np.load(os.path.join(GOLDEN, 'dump_tls_ss.npz'))
    # the dump's run used the callable: same system, so J_T_ss decreases alike (3 s.f. of the first iteration)
    J1 = 1 - abs(res_arr.tau_vals[1][0]) ** 2
    J1_ref = 1 - abs(g['tau_vals'][1][0]) ** 2
    assert abs(J1 - J1_ref) < 5e-3
    assert res_arr.optimized_controls[0].shape == tlist.shape and res_arr.guess_controls[0] is not guess_array
