"""NumPy restatement of the Krotov hot path of qucontrol/krotov.

TEST INFRASTRUCTURE ONLY.  Imported by ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py``; never by ``krotov_amd``.

Parity pinning (see tests/test_oracle_golden.py, tests/golden/README.md):
  * golden vectors extracted from the reference's shipped ``Result`` dumps
    (TLS 19 iterations, ensemble it. 12->24, non-Hermitian it. 40->45,
    transmon N=17 it. 5->8), and
  * outputs of the reference's real ``optimize_pulses`` loop run in this
    container under stub third-party modules (tests/golden/make_reference_goldens.py).

What is restated (reference file:line, relative to /root/reference/src/krotov):
  * time discretisation of controls     conversions.py:61-137, 333-390
  * control/shape initialisation        optimize.py:641-704, 605-620
  * single step exp(f*(H0+sum eps H)*dt) propagators.py:79-122
  * iteration-0 forward propagation     optimize.py:302-322, 806-846
  * chi boundary conditions             functionals.py:177-197, 225-253, 293-317, 389-437
  * chi normalisation                   optimize.py:407-410
  * backward sweep (stores chi(t_n))    optimize.py:413-425, 849-886
  * forward sweep + sequential update   optimize.py:444-508, 889-911
  * dH/d eps                            mu.py:123-140
  * overlap                             second_order.py:69-83
  * update shapes                       shapes.py:51-174
  * DensityMatrixODEPropagator step     propagators.py:162-327 (``step_ode``: the same
    ``scipy.integrate.ode`` 'zvode' call with the reference's defaults, re-initialised
    every step as ``reentrant=True`` does; pinned by the reference's own result for it,
    docs/notebooks/3states_opt_result.dump -> tests/golden/dump_3states.npz)

The arithmetic below L1 lives in QuTiP 4.x / SciPy (not vendored in the
reference).  The dense matrix exponential here is the published [13/13] Pade
scaling-and-squaring algorithm (Higham, SIAM J. Matrix Anal. Appl. 26 (2005)
1179), i.e. what ``scipy.linalg.expm`` (under ``Qobj.expm``) evaluates at its
highest order; ``use_scipy=True`` switches to SciPy's own routine where present.

All arrays are complex128 / float64.  States are flat length-N vectors; density
matrices are column-stacked vec(rho) (propagators.py:255-257, 306-307).
"""
import math

import numpy as np

try:  # load SciPy's own OpenBLAS copy *before* pinning, so the pin covers it too
    import scipy.linalg as _scipy_linalg  # noqa: F401
except Exception:  # pragma: no cover
    _scipy_linalg = None

try:  # single-threaded BLAS, as the reference pins it (optimize.py:233-238,
    # propagators.py:116); 8 OpenBLAS threads on 64x64 operands are ~40x slower
    import threadpoolctl as _tpc

    _BLAS_LIMIT = _tpc.threadpool_limits(limits=1, user_api='blas')
except Exception:  # pragma: no cover
    _BLAS_LIMIT = None

__all__ = [
    'expm_pade13',
    'expm_dense',
    'control_onto_interval',
    'pulse_onto_tlist',
    'discretize',
    'flattop',
    'blackman',
    'box',
    'OracleProblem',
    'chis_re',
    'chis_ss',
    'chis_sm',
    'chis_hs',
    'initialize_controls',
    'step',
    'step_ode',
    'forward_propagation',
    'backward_sweep',
    'forward_update_sweep',
    'krotov_iteration',
    'optimize',
]

# --------------------------------------------------------------------------
# dense matrix exponential
# --------------------------------------------------------------------------

_PADE13_B = (
    64764752532480000.0,
    32382376266240000.0,
    7771770303897600.0,
    1187353796428800.0,
    129060195264000.0,
    10559470521600.0,
    670442572800.0,
    33522128640.0,
    1323241920.0,
    40840800.0,
    960960.0,
    16380.0,
    182.0,
    1.0,
)
_THETA13 = 5.371920351148152


def expm_pade13(A):
    """exp(A) by [13/13] Pade approximation with scaling and squaring.

    Published algorithm (Higham 2005, Alg. 2.3, m=13 branch).  The reference
    reaches the same algorithm via ``Qobj.expm`` -> ``scipy.linalg.expm``
    (propagators.py:115-117).
    """
    A = np.asarray(A, dtype=np.complex128)
    n = A.shape[0]
    norm1 = np.abs(A).sum(axis=0).max() if n > 0 else 0.0
    s = 0
    if norm1 > _THETA13:
        s = max(0, int(math.ceil(math.log2(norm1 / _THETA13))))
    if s > 0:
        A = A / (2.0**s)
    b = _PADE13_B
    ident = np.eye(n, dtype=np.complex128)
    A2 = A @ A
    A4 = A2 @ A2
    A6 = A4 @ A2
    U = A @ (
        A6 @ (b[13] * A6 + b[11] * A4 + b[9] * A2)
        + b[7] * A6
        + b[5] * A4
        + b[3] * A2
        + b[1] * ident
    )
    V = (
        A6 @ (b[12] * A6 + b[10] * A4 + b[8] * A2)
        + b[6] * A6
        + b[4] * A4
        + b[2] * A2
        + b[0] * ident
    )
    R = np.linalg.solve(V - U, V + U)
    for _ in range(s):
        R = R @ R
    return R


_scipy_expm = None


def expm_dense(A, use_scipy=False):
    """Dense matrix exponential; SciPy's if asked for and importable."""
    global _scipy_expm
    if use_scipy:
        if _scipy_expm is None:
            try:
                from scipy.linalg import expm as _e

                _scipy_expm = _e
            except ImportError:  # GPU box without scipy: own Pade-13
                _scipy_expm = expm_pade13
        return _scipy_expm(A)
    return expm_pade13(A)


# --------------------------------------------------------------------------
# controls <-> pulses  (conversions.py:61-137, 333-390)
# --------------------------------------------------------------------------


def control_onto_interval(control):
    """Controls on grid points -> pulses on intervals (conversions.py:333-365)."""
    control = np.asarray(control, dtype=np.float64)
    pulse = np.zeros(len(control) - 1, dtype=np.float64)
    pulse[0] = control[0]
    for i in range(1, len(control) - 1):
        pulse[i] = 2.0 * control[i] - pulse[i - 1]
    pulse[-1] = control[-1]
    return pulse


def pulse_onto_tlist(pulse):
    """Pulses on intervals -> controls on grid points (conversions.py:368-390)."""
    pulse = np.asarray(pulse, dtype=np.float64)
    control = np.zeros(len(pulse) + 1, dtype=np.float64)
    control[0] = pulse[0]
    for i in range(1, len(control) - 1):
        control[i] = 0.5 * (pulse[i - 1] + pulse[i])
    control[-1] = pulse[-1]
    return control


def discretize(control, tlist, args=(None,), via_midpoints=False):
    """Sample a callable control (or check an array one) on ``tlist``.

    Follows conversions.py:105-137: with ``via_midpoints`` the callable is
    sampled at ``tlist + dt/2`` (first/last point replaced by the boundary
    values) and un-averaged with :func:`pulse_onto_tlist`.
    """
    tlist = np.asarray(tlist, dtype=np.float64)
    if callable(control):
        if via_midpoints:
            mid = (tlist + 0.5 * (tlist[1] - tlist[0]))[:-1]
            mid[0] = tlist[0]
            mid[-1] = tlist[-1]
            on_mid = np.array(
                [float(control(t, *args)) for t in mid], dtype=np.float64
            )
            return pulse_onto_tlist(on_mid)
        return np.array(
            [float(control(t, *args)) for t in tlist], dtype=np.float64
        )
    control = np.array([float(v) for v in control], dtype=np.float64)
    if len(control) != len(tlist):
        raise ValueError(
            "If control is an array, it must of the same length as tlist"
        )
    return control


# --------------------------------------------------------------------------
# shapes (shapes.py:51-174)
# --------------------------------------------------------------------------


def box(t, t_start, t_stop):
    if t < t_start or t > t_stop:
        return 0.0
    return 1.0


def blackman(t, t_start, t_stop, a=0.16):
    T = t_stop - t_start
    return (
        0.5
        * box(t, t_start, t_stop)
        * (
            1.0
            - a
            - np.cos(2.0 * np.pi * (t - t_start) / T)
            + a * np.cos(4.0 * np.pi * (t - t_start) / T)
        )
    )


def flattop(t, t_start, t_stop, t_rise, t_fall=None, func='blackman'):
    if t_fall is None:
        t_fall = t_rise
    if not (t_start <= t <= t_stop):
        return 0.0
    f = 1.0
    if func == 'blackman':
        if t <= t_start + t_rise:
            f = blackman(t, t_start, t_start + 2 * t_rise)
        elif t >= t_stop - t_fall:
            f = blackman(t, t_stop - 2 * t_fall, t_stop)
    elif func == 'sinsq':
        if t <= t_start + t_rise:
            f = np.sin(np.pi * (t - t_start) / (2.0 * t_rise)) ** 2
        elif t >= t_stop - t_fall:
            f = np.sin(np.pi * (t - t_stop) / (2.0 * t_fall)) ** 2
    else:
        raise ValueError("Invalid func: %s" % func)
    return f


# --------------------------------------------------------------------------
# problem container
# --------------------------------------------------------------------------


class OracleProblem:
    """Array form of a list of objectives sharing L controls.

    Attributes:
        ops: list (K) of lists ``[H0, H_1, ..., H_L]``; each entry an (N, N)
            complex128 array, or None when control l does not occur in
            objective k (mu.py:126-127 -> zero operator).  ``H_l`` is already
            the sum over all places the control occurs in the objective's
            nested list (mu.py:129-134).
        is_super: True when the operators are Liouvillians acting on
            column-stacked vec(rho) (equation-of-motion factor 1 instead of
            -i, propagators.py:94-99; mu carries an extra factor i,
            mu.py:130-134).
        init, target: (K, N) complex128.
        weights: (K,) float or None (``Objective.weight``, functionals.py).
        tlist: (nt,) float64.
    """

    def __init__(self, ops, init, target, tlist, is_super=False, weights=None, ode=None):
        # ``ode``: dict of DensityMatrixODEPropagator options (possibly empty) -> every step is
        # ``step_ode`` on sparse (CSR) operators instead of the dense matrix exponential
        self.ode = ode
        if ode is not None:
            self.ops = [[None if o is None else o.tocsr().astype(np.complex128) for o in row] for row in ops]
        else:
            self.ops = [
                [None if o is None else np.asarray(o, dtype=np.complex128) for o in row]
                for row in ops
            ]
        self.init = np.asarray(init, dtype=np.complex128)
        self.target = np.asarray(target, dtype=np.complex128)
        self.tlist = np.asarray(tlist, dtype=np.float64)
        self.is_super = bool(is_super)
        self.weights = (
            None if weights is None else np.asarray(weights, dtype=np.float64)
        )
        self.K = len(self.ops)
        self.N = self.init.shape[1]
        self.L = len(self.ops[0]) - 1
        assert self.init.shape == (self.K, self.N)
        assert self.target.shape == (self.K, self.N)

    def adjoint_ops(self):
        """Operators of the adjoint objectives (objectives.py:240-258)."""
        if self.ode is not None:
            return [[None if o is None else o.conj().T.tocsr() for o in row] for row in self.ops]
        return [
            [None if o is None else o.conj().T for o in row] for row in self.ops
        ]


# --------------------------------------------------------------------------
# chi constructors (functionals.py)
# --------------------------------------------------------------------------


def _w(problem):
    if problem.weights is None:
        return np.ones(problem.K)
    return problem.weights


def chis_re(problem, fw_T, tau):
    """functionals.py:293-317: chi_k = w_k/(2K) target_k."""
    c = 1.0 / (2 * problem.K)
    return (c * _w(problem))[:, None] * problem.target


def chis_ss(problem, fw_T, tau):
    """functionals.py:177-197: chi_k = (tau_k/K) w_k target_k."""
    return ((tau / problem.K) * _w(problem))[:, None] * problem.target


def chis_sm(problem, fw_T, tau):
    """functionals.py:225-253: chi_k = w_k/K^2 (sum_j w_j tau_j) target_k."""
    w = _w(problem)
    s = 0
    for wk, t in zip(w, tau):  # same left-to-right order as the reference
        s += wk * t
    c = 1.0 / problem.K**2
    return (c * w)[:, None] * problem.target * s


def chis_hs(problem, fw_T, tau):
    """functionals.py:389-437: chi_k = w_k/(2K) (rho_tgt - rho(T))."""
    c = 1.0 / (2 * problem.K)
    return (c * _w(problem))[:, None] * (problem.target - fw_T)


# --------------------------------------------------------------------------
# single step (propagators.py:79-122)
# --------------------------------------------------------------------------


def _eqm_factor(is_super, backwards):
    f = 1.0 + 0.0j if is_super else -1.0j
    if backwards:
        f = f.conjugate()
    return f


def step(ops_k, eps_n, dt, state, is_super=False, backwards=False, use_scipy=False):
    """One call of ``krotov.propagators.expm`` on array operands.

    ``A = f*H0 + sum_l (f*eps_l)*H_l`` (propagators.py:100-111), then
    ``expm(A*dt) @ state`` (propagators.py:117).
    """
    f = _eqm_factor(is_super, backwards)
    A = f * ops_k[0]
    for l in range(1, len(ops_k)):
        if ops_k[l] is not None:
            A = A + (f * eps_n[l - 1]) * ops_k[l]
    return expm_dense(A * dt, use_scipy) @ state


_ODE_DEFAULTS = dict(method='adams', order=12, atol=1e-8, rtol=1e-6, nsteps=1000, first_step=0, min_step=0,
                     max_step=0)  # propagators.py:181-191


def step_ode(ops_k, eps_n, dt, state, options=None):
    """One call of ``DensityMatrixODEPropagator(reentrant=True)`` (propagators.py:162-327): integrate
    d/dt vec(rho) = (L0 + sum_l eps_l L_l) vec(rho) over ``dt`` with SciPy's 'zvode' (Adams, order 12, atol 1e-8,
    rtol 1e-6), the integrator re-initialised for the step (:242-243, :308-327); the right-hand side is the sum of
    coefficient x CSR matrix-vector products (:264-275).  ``backwards`` has no effect there (:216-220): the caller
    passes the adjoint objective's operators."""
    import scipy.integrate

    opts = dict(_ODE_DEFAULTS)
    opts.update(options or {})
    terms = [(ops_k[0], 1.0)] + [(ops_k[l], eps_n[l - 1]) for l in range(1, len(ops_k)) if ops_k[l] is not None]

    def rhs(t, rho):
        out = np.zeros(rho.shape[0], dtype=complex)
        for L, coeff in terms:
            out += coeff * (L @ rho)
        return out

    r = scipy.integrate.ode(rhs)
    r.set_integrator('zvode', **opts)
    r.set_initial_value(np.asarray(state, dtype=np.complex128))
    r.integrate(dt)
    return np.array(r.y)


def _step(problem, ops_k, eps_n, dt, state, backwards, use_scipy):
    if problem.ode is not None:
        return step_ode(ops_k, eps_n, dt, state, problem.ode)
    return step(ops_k, eps_n, dt, state, problem.is_super, backwards, use_scipy)


# --------------------------------------------------------------------------
# control initialisation (optimize.py:641-704)
# --------------------------------------------------------------------------


def initialize_controls(controls, update_shapes, tlist, args=None):
    """Guess controls/pulses and shape arrays exactly as optimize.py:641-704.

    ``controls``: list of callables ``f(t, args)`` or arrays on ``tlist``.
    ``update_shapes``: list of callables ``S(t)`` or the values 0/1.
    Returns ``(guess_controls, guess_pulses, shape_arrays)``.
    """
    if args is None:
        args = [None] * len(controls)
    guess_controls = [
        discretize(c, tlist, args=(a,), via_midpoints=True)
        for c, a in zip(controls, args)
    ]
    guess_pulses = [control_onto_interval(c) for c in guess_controls]
    shape_arrays = []
    for S in update_shapes:
        if not callable(S):
            if S == 1:
                S = lambda t: 1  # noqa: E731  (shapes.py:46-48)
            elif S == 0:
                S = lambda t: 0  # noqa: E731  (shapes.py:41-43)
            else:
                raise ValueError("update_shape must be a callable")
        arr = control_onto_interval(
            discretize(S, tlist, args=(), via_midpoints=True)
        )
        if np.min(arr) < -0.01 or np.max(arr) > 1.01:  # optimize.py:614-619
            raise ValueError("Update shapes must have values in [0, 1]")
        shape_arrays.append(np.clip(arr, 0.0, 1.0))
    return guess_controls, guess_pulses, shape_arrays


# --------------------------------------------------------------------------
# sweeps
# --------------------------------------------------------------------------


def forward_propagation(problem, pulses, store=False, use_scipy=False):
    """Iteration-0 forward propagation (optimize.py:302-313, 806-846).

    Returns ``fw_T`` (K, N), and the full (K, nt, N) history if ``store``.
    """
    tl = problem.tlist
    nt = len(tl)
    K, N = problem.K, problem.N
    out = np.empty((K, nt, N), dtype=np.complex128) if store else None
    fw_T = np.empty((K, N), dtype=np.complex128)
    for k in range(K):
        state = problem.init[k].copy()
        if store:
            out[k, 0] = state
        for n in range(nt - 1):
            dt = tl[n + 1] - tl[n]
            eps = [p[n] for p in pulses]
            state = _step(problem, problem.ops[k], eps, dt, state, False, use_scipy)
            if store:
                out[k, n + 1] = state
        fw_T[k] = state
    if store:
        return fw_T, out
    return fw_T


def tau_vals(problem, fw_T):
    """tau_k = <target_k | phi_k(T)> (optimize.py:316-322, 502-508)."""
    return np.array(
        [np.vdot(problem.target[k], fw_T[k]) for k in range(problem.K)],
        dtype=np.complex128,
    )


def backward_sweep(problem, chi_T, pulses, use_scipy=False, objectives=None):
    """Backward propagation storing chi_k(t_n) (optimize.py:413-425, 849-886).

    Uses the adjoint objectives' operators, the guess ``pulses`` (conjugated,
    a no-op for real pulses) and ``backwards=True``.
    Returns (K, nt, N) with ``[:, -1] = chi_T``.
    """
    tl = problem.tlist
    nt = len(tl)
    adj = problem.adjoint_ops()
    ks = range(problem.K) if objectives is None else objectives
    out = np.empty((len(list(ks)), nt, problem.N), dtype=np.complex128)
    for i, k in enumerate(ks):
        state = chi_T[k].copy()
        out[i, nt - 1] = state
        for n in range(nt - 2, -1, -1):
            dt = tl[n + 1] - tl[n]
            eps = [np.conjugate(p[n]) for p in pulses]
            state = _step(problem, adj[k], eps, dt, state, True, use_scipy)
            out[i, n] = state
    return out


def _mu_apply(problem, k, l, state):
    """(dH/d eps_l) |state> for objective k (mu.py:123-134)."""
    op = problem.ops[k][1 + l]
    if op is None:
        return 0 * state
    if problem.is_super:
        return 1j * (op @ state)
    return op @ state


def forward_update_sweep(problem, chi_store, chi_norms, guess_pulses, shapes, lambdas, use_scipy=False,
                         sigma_vals=None, fw_prev=None, store=False):
    """Forward sweep with sequential pulse update (optimize.py:444-508).

    At interval n: ``D_l = sum_k ||chi_k|| <chi_k(t_n)| mu_lk |phi_k(t_n)>``,
    ``eps_l[n] += S_l[n]/lambda_l * Im D_l``, then every phi_k is propagated
    over interval n with the *updated* pulses.

    Second order (optimize.py:434-443, 468-469, 492-500): with ``sigma_vals``
    (sigma at the interval mid-points) and ``fw_prev`` (K, nt, N), the states
    propagated under the guess pulses, every summand gets the extra term
    ``0.5 sigma_n <Delta phi_k(t_n)| mu |phi_k(t_n)>`` with ``Delta phi =
    phi_k(t_n) - fw_prev[k, n]`` (zero at n = 0).

    Returns ``(optimized_pulses, fw_T, g_a_integrals)`` and, with ``store``, the
    (K, nt, N) forward states as a fourth element.
    """
    tl = problem.tlist
    nt = len(tl)
    K, L = problem.K, len(guess_pulses)
    opt = [np.array(p, dtype=np.float64, copy=True) for p in guess_pulses]
    g_a = np.zeros(L)
    fw = [problem.init[k].copy() for k in range(K)]
    second_order = sigma_vals is not None
    out = np.empty((K, nt, problem.N), dtype=np.complex128) if store else None
    if store:
        out[:, 0] = problem.init
    delta = [np.zeros(problem.N, dtype=np.complex128) for _ in range(K)]  # optimize.py:437-440
    for n in range(nt - 1):
        dt = tl[n + 1] - tl[n]
        for l in range(L):
            d = 0j
            for k in range(K):  # optimize.py:455-470, k-order summation
                mu_psi = _mu_apply(problem, k, l, fw[k])
                update = np.vdot(chi_store[k, n], mu_psi)
                update *= chi_norms[k]
                if second_order:
                    update += 0.5 * sigma_vals[n] * np.vdot(delta[k], mu_psi)
                d += update
            S_t = shapes[l][n]
            d1 = d.imag
            opt[l][n] += (S_t / lambdas[l]) * d1
            g_a[l] += (S_t / lambdas[l]) * abs(d1) ** 2 * dt
        eps = [p[n] for p in opt]
        for k in range(K):
            fw[k] = _step(problem, problem.ops[k], eps, dt, fw[k], False, use_scipy)
            if second_order:
                delta[k] = fw[k] - fw_prev[k, n + 1]  # optimize.py:494-497
            if store:
                out[k, n + 1] = fw[k]
    if store:
        return opt, np.array(fw), g_a, out
    return opt, np.array(fw), g_a


def numerical_estimate_A(fw_T, fw_T0, chi_T, chi_norms, Delta_J_T):
    """Second-order parameter A (second_order.py:86-141), from the final states
    of the current (fw_T) and previous (fw_T0) iteration."""
    dphi = fw_T - fw_T0
    denom = float(np.sum(np.abs(dphi) ** 2))
    if denom > 1.0e-30:
        numer = sum((2 * chi_norms[k] * np.vdot(chi_T[k], dphi[k])).real for k in range(len(dphi))) + Delta_J_T
        return numer / denom
    return 0


def default_norm(problem, chi):
    """``Qobj.norm()`` default: L2 for kets, trace norm for operators
    (optimize.py:243; QuTiP default, third-party)."""
    if problem.is_super:
        d = int(round(math.sqrt(problem.N)))
        mat = chi.reshape((d, d), order='F')
        return float(np.linalg.svd(mat, compute_uv=False).sum())
    return float(np.linalg.norm(chi))


def krotov_iteration(problem, guess_pulses, shapes, lambdas, fw_T, tau, chi_constructor, use_scipy=False, norm=None, return_chi=False):
    """One Krotov iteration, optimize.py:393-510."""
    chi_T = chi_constructor(problem, fw_T, tau)
    nrm = default_norm if norm is None else norm
    chi_norms = np.array([nrm(problem, chi_T[k]) for k in range(problem.K)])
    chi_T = chi_T / chi_norms[:, None]  # optimize.py:410
    chi_store = backward_sweep(problem, chi_T, guess_pulses, use_scipy)
    opt, fw_T, g_a = forward_update_sweep(
        problem, chi_store, chi_norms, guess_pulses, shapes, lambdas, use_scipy
    )
    tau = tau_vals(problem, fw_T)
    if return_chi:
        return opt, fw_T, tau, g_a, chi_store, chi_norms
    return opt, fw_T, tau, g_a


def optimize(problem, guess_pulses, shapes, lambdas, chi_constructor, iter_stop, use_scipy=False, norm=None,
             sigma=None):
    """Iteration 0 + ``iter_stop`` iterations; returns per-iteration records.

    ``sigma``: object with ``__call__(t)`` and ``refresh(fw_T, fw_T0, chi_T,
    chi_norms, tau_history)`` for the second-order update (the oracle's array
    form of ``krotov.second_order.Sigma``; optimize.py:566-577).

    Returns dict with ``all_pulses`` (iter_stop+1, L, nt-1), ``tau_vals``
    (iter_stop+1, K), ``g_a`` (iter_stop+1, L), ``fw_T`` (K, N).
    """
    pulses = [np.array(p, dtype=np.float64, copy=True) for p in guess_pulses]
    tl = problem.tlist
    if sigma is None:
        fw_T = forward_propagation(problem, pulses, use_scipy=use_scipy)
        fw_prev = None
    else:
        fw_T, fw_prev = forward_propagation(problem, pulses, store=True, use_scipy=use_scipy)
    tau = tau_vals(problem, fw_T)
    all_pulses = [np.array(pulses)]
    taus = [tau]
    gas = [np.zeros(len(pulses))]
    nrm = default_norm if norm is None else norm
    for _ in range(iter_stop):
        if sigma is None:
            pulses, fw_T, tau, g_a = krotov_iteration(
                problem, pulses, shapes, lambdas, fw_T, tau, chi_constructor, use_scipy, norm
            )
        else:
            chi_T = chi_constructor(problem, fw_T, tau)
            chi_norms = np.array([nrm(problem, chi_T[k]) for k in range(problem.K)])
            chi_T = chi_T / chi_norms[:, None]
            chi_store = backward_sweep(problem, chi_T, pulses, use_scipy)
            sig = [sigma(tl[n] + 0.5 * (tl[n + 1] - tl[n])) for n in range(len(tl) - 1)]  # optimize.py:452
            fw_T0 = fw_T
            pulses, fw_T, g_a, fw_new = forward_update_sweep(
                problem, chi_store, chi_norms, pulses, shapes, lambdas, use_scipy,
                sigma_vals=sig, fw_prev=fw_prev, store=True)
            tau = tau_vals(problem, fw_T)
        all_pulses.append(np.array(pulses))
        taus.append(tau)
        gas.append(g_a)
        if sigma is not None:
            sigma.refresh(fw_T, fw_T0, chi_T, chi_norms, taus)
            fw_prev = fw_new
    return dict(
        all_pulses=np.array(all_pulses),
        tau_vals=np.array(taus),
        g_a=np.array(gas),
        fw_T=fw_T,
    )
