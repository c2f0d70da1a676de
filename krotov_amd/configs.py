"""Seeded synthetic control problems for the configurations named in BASELINE.json.

Pure NumPy; used by ``bench.py``, the parity tests and the golden-vector
generator so that all of them see bit-identical inputs.  Every builder returns
a :class:`ProblemSpec` of plain arrays plus the guess-control / update-shape
callables; :func:`spec_to_objectives` turns a spec into ``Objective`` instances
of any krotov-compatible module (this package, or the reference in its
"numpy mode", docs/notebooks/09_example_numpy.ipynb of the reference).

Config definitions follow SURVEY.md section 8(d).
"""
import numpy as np

from . import shapes as _shapes

__all__ = [
    'ProblemSpec',
    'config_c1',
    'config_c2_hilbert',
    'config_c2_liouville',
    'config_c3',
    'config_c4',
    'config_c5',
    'spec_to_objectives',
    'liouvillian_dense',
    'herm',
]


class ProblemSpec:
    """Plain-array description of a control problem.

    Attributes:
        name (str)
        H0 (list of ndarray): K drift operators (N, N) -- entries may be the
            *same object* when objectives share an operator.
        Hc (list of list of ndarray): ``Hc[k][l]`` control operator of control
            ``l`` in objective ``k`` (same-object sharing allowed).
        is_super (bool): operators are Liouvillians on column-stacked vec(rho).
        init, target (ndarray): (K, N) complex128.
        tlist (ndarray): (nt,)
        controls (list of callable): L guess controls ``f(t, args)``.
        update_shape (callable): ``S(t)``.
        lambda_a (float)
        chi (str): 're' | 'ss' | 'sm' | 'hs'
        weights (ndarray or None)
    """

    def __init__(self, **kw):
        self.weights = None
        self.__dict__.update(kw)
        self.K = len(self.H0)
        self.N = self.init.shape[1]
        self.L = len(self.controls)


def herm(rng, N, scale):
    """Random Hermitian matrix with spectral norm ``scale``."""
    G = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    Hm = 0.5 * (G + G.conj().T)
    return scale * Hm / np.linalg.norm(Hm, 2)


def liouvillian_dense(H, c_ops=()):
    """Dense Liouvillian for column-stacked vec(rho).

    ``d/dt vec(rho) = L vec(rho)`` with ``L = -i (I (x) H - H^T (x) I) + sum_c
    (conj(C) (x) C - 1/2 I (x) C^dag C - 1/2 (C^dag C)^T (x) I)``.
    """
    H = np.asarray(H, dtype=np.complex128)
    d = H.shape[0]
    eye = np.eye(d, dtype=np.complex128)
    L = -1j * (np.kron(eye, H) - np.kron(H.T, eye))
    for C in c_ops:
        C = np.asarray(C, dtype=np.complex128)
        CdC = C.conj().T @ C
        L = L + np.kron(C.conj(), C) - 0.5 * np.kron(eye, CdC) - 0.5 * np.kron(CdC.T, eye)
    return L


def _vec(rho):
    return np.asarray(rho, dtype=np.complex128).ravel(order='F')


# --------------------------------------------------------------------------
# C1: two-level |0> -> |1> (reference docs/notebooks/01, cells 5-17)
# --------------------------------------------------------------------------


def _tls_ops(omega=1.0):
    sz = np.diag([1.0, -1.0]).astype(np.complex128)
    sx = np.array([[0, 1], [1, 0]], dtype=np.complex128)
    return -0.5 * omega * sz, sx


def config_c1(nt=500):
    H0, H1 = _tls_ops()

    def guess(t, args):
        return 0.2 * _shapes.flattop(t, t_start=0, t_stop=5, t_rise=0.3, func='blackman')

    def S(t):
        return _shapes.flattop(t, t_start=0, t_stop=5, t_rise=0.3, t_fall=0.3, func='blackman')

    e0 = np.array([1, 0], dtype=np.complex128)
    e1 = np.array([0, 1], dtype=np.complex128)
    return ProblemSpec(
        name='c1_tls_state_to_state',
        H0=[H0], Hc=[[H1]], is_super=False,
        init=e0[None, :], target=e1[None, :],
        tlist=np.linspace(0, 5, nt), controls=[guess], update_shape=S,
        lambda_a=5.0, chi='ss',
    )


# --------------------------------------------------------------------------
# C2: single-qubit X gate on the same TLS
# --------------------------------------------------------------------------


def config_c2_hilbert(nt=500):
    """Hilbert-space variant: K=2 (``gate_objectives``, objectives.py:950-970)."""
    spec = config_c1(nt)
    H0, H1 = spec.H0[0], spec.Hc[0][0]
    basis = np.eye(2, dtype=np.complex128)
    X = np.array([[0, 1], [1, 0]], dtype=np.complex128)
    target = np.array([sum(X[i, j] * basis[i] for i in range(2)) for j in range(2)])
    return ProblemSpec(
        name='c2_xgate_hilbert',
        H0=[H0, H0], Hc=[[H1], [H1]], is_super=False,
        init=basis.copy(), target=target,
        tlist=spec.tlist, controls=spec.controls, update_shape=spec.update_shape,
        lambda_a=5.0, chi='re',
    )


def _three_states(basis):
    """rho_1, rho_2, rho_3 of the '3states' set (objectives.py:675-701)."""
    d = len(basis)
    rho1 = sum((2 * (d - i) / (d * (d + 1))) * np.outer(p, p.conj()) for i, p in enumerate(basis))
    rho2 = (1.0 / d) * sum(np.outer(pi, pj.conj()) for pi in basis for pj in basis)
    rho3 = (1.0 / d) * sum(np.outer(p, p.conj()) for p in basis)
    return [rho1, rho2, rho3]


def config_c2_liouville(nt=500):
    """The literal "3 objectives" reading: '3states' on L = -i[H, .] (vec dim 4)."""
    spec = config_c1(nt)
    H0, H1 = spec.H0[0], spec.Hc[0][0]
    L0 = liouvillian_dense(H0)
    L1 = liouvillian_dense(H1)
    basis = list(np.eye(2, dtype=np.complex128))
    X = np.array([[0, 1], [1, 0]], dtype=np.complex128)
    mapped = [sum(X[i, j] * basis[i] for i in range(2)) for j in range(2)]
    init = np.array([_vec(r) for r in _three_states(basis)])
    target = np.array([_vec(r) for r in _three_states(mapped)])
    return ProblemSpec(
        name='c2_xgate_liouville_3states',
        H0=[L0] * 3, Hc=[[L1]] * 3, is_super=True,
        init=init, target=target,
        tlist=spec.tlist, controls=spec.controls, update_shape=spec.update_shape,
        lambda_a=5.0, chi='re',
    )


# --------------------------------------------------------------------------
# C3: two-qubit iSWAP (Hamiltonian of reference docs/notebooks/07, cell 12)
# --------------------------------------------------------------------------


def config_c3(nt=2001):
    w1, w2, J, u0, la, T = 1.1, 2.1, 0.2, 0.3, 1.1, 25.0
    Hq1 = 0.5 * w1 * np.diag([-1, 1])
    Hq2 = 0.5 * w2 * np.diag([-1, 1])
    sx = np.array([[0, 1], [1, 0]])
    sy = np.array([[0, -1j], [1j, 0]])
    H0 = np.kron(Hq1, np.eye(2)) + np.kron(np.eye(2), Hq2)
    H0 = (H0 + 2 * J * (np.kron(sx, sx) + np.kron(sy, sy))).astype(np.complex128)
    H1 = (np.kron(sx, np.eye(2)) + la * np.kron(np.eye(2), sx)).astype(np.complex128)

    def guess(t, args):
        return u0 * _shapes.flattop(t, t_start=0, t_stop=T, t_rise=T / 20, t_fall=T / 20, func='sinsq')

    def S(t):
        return _shapes.flattop(t, t_start=0, t_stop=T, t_rise=T / 20, t_fall=T / 20, func='sinsq')

    iswap = np.array(
        [[1, 0, 0, 0], [0, 0, 1j, 0], [0, 1j, 0, 0], [0, 0, 0, 1]], dtype=np.complex128
    )
    basis = np.eye(4, dtype=np.complex128)
    target = np.array([sum(iswap[i, j] * basis[i] for i in range(4)) for j in range(4)])
    return ProblemSpec(
        name='c3_iswap',
        H0=[H0] * 4, Hc=[[H1]] * 4, is_super=False,
        init=basis.copy(), target=target,
        tlist=np.linspace(0, T, nt), controls=[guess], update_shape=S,
        lambda_a=100.0, chi='sm',
    )


# --------------------------------------------------------------------------
# C4: transmon X gate in Liouville space (the build's concretisation)
# --------------------------------------------------------------------------


def config_c4(d=20, nt=1001, n_logical=4, gamma=1e-3):
    """Transmon of tests/transmon_xgate_system_mod.py:18-27 (reference) with
    ``d`` charge states, one lowering-type decay operator in the eigenbasis,
    Liouvillian of dimension d^2, K = n_logical^2 density-matrix objectives
    ('full' set, objectives.py:971-981) sharing one operator list."""
    Ec, EjEc, ng, T = 0.386, 45, 0.0, 10.0
    Ej = EjEc * Ec
    n = np.arange(-(d // 2), d - d // 2)
    up = np.diag(np.ones(d - 1), k=-1)
    H0 = (np.diag(4 * Ec * (n - ng) ** 2) - Ej * (up + up.T) / 2.0).astype(np.complex128)
    H1 = (-2 * np.diag(n)).astype(np.complex128)
    evals, V = np.linalg.eigh(H0)
    # fix eigenvector signs so the fixture does not depend on LAPACK's choice
    for j in range(d):
        i = np.argmax(np.abs(V[:, j]))
        if V[i, j].real < 0:
            V[:, j] = -V[:, j]
    a_eig = np.diag(np.sqrt(np.arange(1, d)), k=1)  # lowering operator in the eigenbasis
    C = np.sqrt(gamma) * (V @ a_eig @ V.conj().T)
    L0 = liouvillian_dense(H0, [C])
    L1 = liouvillian_dense(H1)

    def guess(t, args):
        return 4 * np.exp(-40.0 * (t / T - 0.5) ** 2)

    def S(t):
        return _shapes.flattop(t, t_start=0.0, t_stop=T, t_rise=0.5, func='sinsq')

    basis = [V[:, j].astype(np.complex128) for j in range(n_logical)]
    gate = np.eye(n_logical, dtype=np.complex128)
    gate[:2, :2] = [[0, 1], [1, 0]]  # X on the lowest two levels
    mapped = [sum(gate[i, j] * basis[i] for i in range(n_logical)) for j in range(n_logical)]
    init = np.array([_vec(np.outer(pi, pj.conj())) for pi in basis for pj in basis])
    target = np.array([_vec(np.outer(pi, pj.conj())) for pi in mapped for pj in mapped])
    K = n_logical * n_logical
    return ProblemSpec(
        name='c4_transmon_liouville_d%d' % d,
        H0=[L0] * K, Hc=[[L1]] * K, is_super=True,
        init=init, target=target,
        tlist=np.linspace(0, T, nt), controls=[guess], update_shape=S,
        lambda_a=1.0, chi='re',
    )


def config_shared(K=20, N=96, nt=21, L=2, seed=3):
    """K state-to-state objectives under ONE operator list (the ``gate_objectives``
    pattern: same H, different basis states), Hilbert dimension N > 64, L controls:
    the shared-operator (dense product) case of BASELINE config 4 in Hilbert space.
    ``||H0|| dt = 0.4``, ``||H_l|| dt = 0.1``."""
    rng = np.random.default_rng(seed)
    T = (nt - 1) / 4000.0
    dt = T / (nt - 1)
    H0 = herm(rng, N, 0.4 / dt)
    Hl = [herm(rng, N, 0.1 / dt) for _ in range(L)]
    init = rng.standard_normal((K, N)) + 1j * rng.standard_normal((K, N))
    init /= np.linalg.norm(init, axis=1)[:, None]
    target = rng.standard_normal((K, N)) + 1j * rng.standard_normal((K, N))
    target /= np.linalg.norm(target, axis=1)[:, None]

    def make_guess(l):
        return lambda t, args: 0.5 * np.sin((l + 1) * np.pi * t / T)

    def S(t):
        return _shapes.flattop(t, t_start=0.0, t_stop=T, t_rise=0.05 * T, func='sinsq')

    return ProblemSpec(
        name='shared_K%d_N%d_L%d' % (K, N, L),
        H0=[H0] * K, Hc=[list(Hl)] * K, is_super=False, init=init, target=target,
        tlist=np.linspace(0, T, nt), controls=[make_guess(l) for l in range(L)], update_shape=S,
        lambda_a=2.0, chi='re',
    )


def config_sparse_lindblad(d=12, nt=61, K=3, gamma=0.05):
    """A driven, damped d-level ladder in Liouville space: nearest-neighbour
    hopping + anharmonic levels, decay through the lowering operator, control on
    the level energies.  The Liouvillians (dimension d^2) have a handful of
    entries per row -- the regime of the reference's DensityMatrixODEPropagator
    (propagators.py:162-327).  ``H0``/``Hc`` hold dense arrays; see
    :func:`sparse_ops` for the ``scipy.sparse`` form."""
    T = 4.0
    n = np.arange(d)
    hop = np.diag(np.sqrt(np.arange(1, d)), k=1)
    H0 = (np.diag(0.3 * n - 0.02 * n * (n - 1)) + 0.25 * (hop + hop.T)).astype(np.complex128)
    H1 = np.diag(n / (d - 1.0)).astype(np.complex128)
    C = np.sqrt(gamma) * hop
    L0 = liouvillian_dense(H0, [C])
    L1 = liouvillian_dense(H1)

    def guess(t, args):
        return 0.8 * np.sin(np.pi * t / T) ** 2

    def S(t):
        return _shapes.flattop(t, t_start=0.0, t_stop=T, t_rise=0.4, func='sinsq')

    basis = np.eye(d, dtype=np.complex128)
    init = np.array([_vec(np.outer(basis[j], basis[j].conj())) for j in range(K)])
    target = np.array([_vec(np.outer(basis[(j + 1) % d], basis[(j + 1) % d].conj())) for j in range(K)])
    return ProblemSpec(
        name='sparse_lindblad_d%d' % d, H0=[L0] * K, Hc=[[L1]] * K, is_super=True, init=init, target=target,
        tlist=np.linspace(0, T, nt), controls=[guess], update_shape=S, lambda_a=2.0, chi='re',
    )


def sparse_ops(spec):
    """``[[op_0, op_1, ...] per objective]`` of a spec as ``scipy.sparse.csr_matrix``
    objects (one per distinct array, so sharing is preserved)."""
    import scipy.sparse as sp

    made = {}

    def conv(a):
        if id(a) not in made:
            m = sp.csr_matrix(a)
            m.eliminate_zeros()
            made[id(a)] = (m, a)
        return made[id(a)][0]

    return [[conv(spec.H0[k])] + [conv(spec.Hc[k][l]) for l in range(spec.L)] for k in range(spec.K)]


# --------------------------------------------------------------------------
# C5: robustness ensemble (the headline configuration)
# --------------------------------------------------------------------------


def config_c5(K=256, N=64, nt=4001, L=1, distinct=False, seed=0, T=None, lambda_a=None):
    """K objectives, Hilbert dimension N, L controls.

    Objective k evolves under ``H0 + mu_k * sum_l eps_l(t) H_l`` with
    ``mu_k = linspace(0.9, 1.1, K)[k]`` (the ``scale_control`` /
    ``ensemble_objectives`` pattern of the reference's notebook 08).  With
    ``distinct=True`` every objective gets its own random drift ``H0_k``
    (seed ``seed + 1 + k``), which defeats any shared-operator shortcut.
    Operator norms are pinned to ``||H0|| dt = 0.4`` and ``||H_l|| dt = 0.1``.
    ``T`` defaults to ``(nt-1)/4000`` so that ``dt = 1/4000`` at every size.
    """
    if T is None:
        T = (nt - 1) / 4000.0
    tlist = np.linspace(0, T, nt)
    dt = tlist[1] - tlist[0]
    rng = np.random.default_rng(seed)
    H0_shared = herm(rng, N, 0.4 / dt)
    Hl = [herm(rng, N, 0.1 / dt) for _ in range(L)]
    mu = np.linspace(0.9, 1.1, K) if K > 1 else np.array([1.0])
    H0, Hc = [], []
    for k in range(K):
        if distinct:
            H0.append(herm(np.random.default_rng(seed + 1 + k), N, 0.4 / dt))
        else:
            H0.append(H0_shared)
        Hc.append([mu[k] * Hl[l] for l in range(L)])
    init = np.zeros((K, N), dtype=np.complex128)
    init[:, 0] = 1.0
    target = np.zeros((K, N), dtype=np.complex128)
    target[:, 1] = 1.0

    def make_guess(l):
        def guess(t, args):
            return 0.5 * np.sin((l + 1) * np.pi * t / T)
        return guess

    def S(t):
        return _shapes.flattop(t, 0.0, T, 0.05 * T, func='sinsq')

    if lambda_a is None:
        # large enough that the first-iteration update (|D| <= ||H_l||/2 =
        # 0.05/dt) stays below the guess amplitude, so ||H(eps)|| dt <= 0.5
        lambda_a = 50.0
    return ProblemSpec(
        name='c5_ensemble_K%d_N%d_L%d%s' % (K, N, L, '_distinct' if distinct else ''),
        H0=H0, Hc=Hc, is_super=False,
        init=init, target=target,
        tlist=tlist, controls=[make_guess(l) for l in range(L)], update_shape=S,
        lambda_a=float(lambda_a), chi='re', mu=mu,
    )


# --------------------------------------------------------------------------
# spec -> Objective list of a krotov-compatible module
# --------------------------------------------------------------------------


def spec_to_objectives(spec, krotov_module, column_states=True):
    """Build ``(objectives, pulse_options)`` for ``krotov_module.optimize_pulses``.

    States are handed over as (N, 1) column arrays when ``column_states`` (the
    convention of the reference's numpy mode), else as flat (N,) vectors.
    Objectives whose ``H0``/``Hc`` entries are the same object share the same
    operator arrays in the nested lists.
    """
    objectives = []
    for k in range(spec.K):
        H = [spec.H0[k]] + [[spec.Hc[k][l], spec.controls[l]] for l in range(spec.L)]
        psi0, tgt = spec.init[k], spec.target[k]
        if column_states:
            psi0, tgt = psi0.reshape(-1, 1), tgt.reshape(-1, 1)
        obj = krotov_module.Objective(initial_state=psi0, target=tgt, H=H)
        if spec.weights is not None:
            obj.weight = float(spec.weights[k])
        objectives.append(obj)
    pulse_options = {
        c: dict(lambda_a=spec.lambda_a, update_shape=spec.update_shape)
        for c in spec.controls
    }
    return objectives, pulse_options
