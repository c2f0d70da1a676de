"""The C-ABI library loads and exports every symbol include/krotov_hip.h declares
(no compute calls: runs without a GPU)."""
import os
import re

from krotov_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'krotov_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(kh_[a-z0-9_]+)\s*\(', text)))


def test_library_builds_and_exports_every_declared_symbol():
    build.build()
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_lib.SYMBOLS) == declared
    assert b'gfx950' in lib.kh_version()


def test_bad_arguments_are_reported_not_crashed():
    lib = _lib.load()
    import ctypes

    out = ctypes.c_void_p()
    rc = lib.kh_engine_create(None, ctypes.byref(out))
    assert rc == -1 and b'null' in lib.kh_last_error()
    pr = _lib.kh_problem()
    pr.K, pr.N, pr.L, pr.nt = 0, 2, 1, 5
    rc = lib.kh_engine_create(ctypes.byref(pr), ctypes.byref(out))
    assert rc == -1 and b'bad sizes' in lib.kh_last_error()
    assert lib.kh_check(None) == -1
    # the other entry points validate before touching the device
    assert lib.kh_engine_create_csr(None, ctypes.byref(out)) == -1
    pc = _lib.kh_problem_csr()
    pc.K, pc.N, pc.L, pc.nt = 1, 2, 0, 3
    assert lib.kh_engine_create_csr(ctypes.byref(pc), ctypes.byref(out)) == -1
    assert b'required' in lib.kh_last_error()
    for call in (lambda: lib.kh_forward_store(None, None, None, None, None, None),
                 lambda: lib.kh_backward_store(None, None, None, None, None),
                 lambda: lib.kh_forward_update(None, None, None, None, None, None, None, None, None, None, None),
                 lambda: lib.kh_set_second_order(None, None, None, None),
                 lambda: lib.kh_chi_boundary(None, None, None, None, None, None, None, None),
                 lambda: lib.kh_tau(None, None, None, None, None)):
        assert call() == -1
        assert lib.kh_last_error() != b''


def test_series_tables_evaluate_the_exponential():
    """kh_series_tables (host code of the library, no GPU): the coefficient tables the kernels use.  Taylor:
    ratios 1/j, degree 14 at theta = 0.5.  Real-spectrum series: fewer terms at the same
    tolerance (degree 12 at theta = 0.5, 14 at theta = 1), thresholds non-decreasing, and -- the point -- the
    polynomial sum_j c_j (-i A)^j v reproduces exp(-i A) v to rounding for Hermitian A with ||A|| = theta[m],
    evaluated exactly the way the two-terms-per-phase kernels do (A^2 chain + one product with A)."""
    import ctypes

    import numpy as np
    import scipy.linalg

    lib = _lib.load()
    theta_t, ratios_t = (ctypes.c_double * 65)(), (ctypes.c_double * (65 * 65))()
    theta_c, ratios_c = (ctypes.c_double * 65)(), (ctypes.c_double * (65 * 65))()
    assert lib.kh_series_tables(0, 0.0, theta_t, ratios_t) == 0
    assert lib.kh_series_tables(1, 0.0, theta_c, ratios_c) == 0
    assert lib.kh_series_tables(1, 0.0, None, ratios_c) == -1
    tt, tc = np.array(theta_t), np.array(theta_c)
    rt, rc = np.array(ratios_t).reshape(65, 65), np.array(ratios_c).reshape(65, 65)
    assert np.all(np.diff(tt) >= 0) and np.all(np.diff(tc) >= 0)
    assert np.all(rt[:, 0] == 1.0) and np.allclose(rt[20, 1:21], 1.0 / np.arange(1, 21), rtol=0, atol=0)
    degree = lambda tab, th: int(np.argmax(tab >= th))  # noqa: E731  smallest m with th <= tab[m]
    assert degree(tt, 0.5) == 14 and degree(tt, 1.0) == 18
    assert degree(tc, 0.5) == 12 and degree(tc, 1.0) == 14
    assert np.all(tc[2:19:2] > tt[2:19:2])  # every even degree up to 18 serves a larger norm
    rng = np.random.default_rng(3)
    N = 24
    G = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    Hm = (G + G.conj().T) / 2
    Hm /= np.linalg.norm(Hm, 2)
    v = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    v /= np.linalg.norm(v)
    for m in (8, 10, 12, 14, 16):
        A = Hm * tc[m]
        c = np.cumprod(np.concatenate([[rc[m, 0], rc[m, 1] / rc[m, 0]], rc[m, 2:m + 1]]))  # c_0 .. c_m
        assert abs(c[0] - 1) < 1e-15 and np.allclose(c, 1 / np.array([scipy.special.factorial(j) for j in range(m + 1)]),
                                                     rtol=1e-3)
        B, f = A @ A, -1j
        state, term, s = c[0] * v, c[0] * v, c[1] * v
        for p in range(m // 2):
            term = (c[2 * p + 2] / c[2 * p]) * (f * f) * (B @ term)
            state = state + term
            if 2 * p + 3 <= m:
                s = s + (c[2 * p + 3] / c[2 * p + 2]) * term
        state = state + f * (A @ s)
        assert np.linalg.norm(state - scipy.linalg.expm(-1j * A) @ v) < 6e-16, m
        # the Taylor polynomial of the same degree is NOT good enough at this norm
        taylor = sum(np.linalg.matrix_power(-1j * A, j) @ v / scipy.special.factorial(j) for j in range(m + 1))
        assert np.linalg.norm(taylor - scipy.linalg.expm(-1j * A) @ v) > 3e-15, m


def test_series_tables_with_hermitian_defect():
    """kh_series_tables_defect (host code, no GPU): the Chebyshev-form tables an engine builds for a generator that is
    anti-Hermitian only up to a small Hermitian part (weakly damped Liouvillians), up to theta = 4 for the cooperative
    kernels.  With defect 0 and cap 2 they are the real-spectrum tables; a defect costs at most a little theta per
    degree; and -- the point -- for a NON-NORMAL matrix A = S + D with S anti-Hermitian, ||S|| = theta[m] (as far as
    D leaves room) and a Hermitian-part norm within the defect, the degree-m polynomial evaluated the way the
    kernels do (A^2 chain + one product with A) reproduces expm(A) v to the rounding level Taylor itself reaches
    there, where the Taylor polynomial of the same degree is far off."""
    import ctypes

    import numpy as np
    import scipy.linalg

    lib = _lib.load()

    def tables(cap, defect):
        th, ra = (ctypes.c_double * 65)(), (ctypes.c_double * (65 * 65))()
        assert lib.kh_series_tables_defect(0.0, cap, defect, th, ra) == 0
        return np.array(th), np.array(ra).reshape(65, 65)

    th_ref, ra_ref = (ctypes.c_double * 65)(), (ctypes.c_double * (65 * 65))()
    assert lib.kh_series_tables(1, 0.0, th_ref, ra_ref) == 0
    t0, r0 = tables(2.0, 0.0)
    assert np.array_equal(t0, np.array(th_ref)) and np.array_equal(r0, np.array(ra_ref).reshape(65, 65))
    assert lib.kh_series_tables_defect(0.0, 9.0, 0.0, th_ref, ra_ref) == -1
    defect = 3e-3
    tc, rc = tables(4.0, defect)
    assert np.all(np.diff(tc) >= 0)
    even = np.arange(8, 23, 2)
    t_cap4 = tables(4.0, 0.0)[0]
    assert np.all(tc[even] <= t_cap4[even]) and np.all(tc[even] > 0.8 * t_cap4[even])  # the margin is small (eta ~ sqrt(defect / theta))
    degree = lambda tab, th: int(np.argmax(tab >= th))  # noqa: E731
    assert degree(tc, 2.7) == 20  # (BASELINE config 4: Taylor needs 24)
    rng = np.random.default_rng(11)
    N = 40
    G = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    S = (G - G.conj().T) / 2
    S /= np.linalg.norm(S, 2)
    D = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))  # non-normal perturbation
    D *= 0.5 * defect / np.linalg.norm((D + D.conj().T) / 2, 2)          # Hermitian part well inside the defect
    v = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    v /= np.linalg.norm(v)
    for m in (12, 16, 20, 22):
        A = S * (tc[m] - np.linalg.norm(D, 2)) + D
        assert np.linalg.norm(A, 2) <= tc[m] and np.linalg.norm((A + A.conj().T) / 2, 2) <= defect
        c = np.cumprod(np.concatenate([[rc[m, 0], rc[m, 1] / rc[m, 0]], rc[m, 2:m + 1]]))  # c_0 .. c_m
        B = A @ A
        state, term, s = c[0] * v, c[0] * v, c[1] * v
        for p in range(m // 2):
            term = (c[2 * p + 2] / c[2 * p]) * (B @ term)
            state = state + term
            if 2 * p + 3 <= m:
                s = s + (c[2 * p + 3] / c[2 * p + 2]) * term
        state = state + A @ s
        exact = scipy.linalg.expm(A) @ v
        assert np.linalg.norm(state - exact) < 4e-15, (m, np.linalg.norm(state - exact))
        taylor = sum(np.linalg.matrix_power(A, j) @ v / scipy.special.factorial(j) for j in range(m + 1))
        assert np.linalg.norm(taylor - exact) > 1e-14, m


def test_series_tables_defect_bound_at_the_corner_of_the_numerical_range():
    """The truncation bound behind kh_series_tables_defect must hold at the point of the admissible numerical range
    that is farthest from the imaginary segment: z = -delta - i sqrt(theta^2 - delta^2), an eigenvalue of a NORMAL
    (diagonal) generator with |z| = theta and Re z = -delta.  At the largest admitted defect (0.05) the scalar
    polynomial of every even degree must reproduce exp(z) to the tolerance the tables were built for (2^-53 x a
    small factor for the rounding of the evaluation itself).  (An earlier bound -- the ellipse through the
    semi-minor axis delta alone -- was ~5x over at this point.)"""
    import ctypes

    import numpy as np

    lib = _lib.load()
    for cap in (2.0, 4.0, 6.0):  # (register-tile kernels; cooperative kernels; sparse kernels)
        for defect in (0.05, 5e-3, 3e-4):
            th, ra = (ctypes.c_double * 65)(), (ctypes.c_double * (65 * 65))()
            assert lib.kh_series_tables_defect(0.0, cap, defect, th, ra) == 0
            th, ra = np.array(th), np.array(ra).reshape(65, 65)
            checked = 0
            for m in range(4, 41, 2):
                theta = th[m]
                if not (defect < theta < cap) or theta <= th[m - 2]:
                    continue  # (degree not served by the Chebyshev form at this cap, or no room for the defect)
                c = np.cumprod(np.concatenate([[ra[m, 0], ra[m, 1] / ra[m, 0]], ra[m, 2:m + 1]])).astype(np.longdouble)
                z = np.clongdouble(complex(-defect, -np.sqrt(theta * theta - defect * defect)))
                poly = sum(c[j] * z ** j for j in range(m + 1))
                exact = np.exp(z)
                err = abs(complex(poly - exact))
                # (a) the mathematics behind the threshold theta[m]: truncation error of the Chebyshev series of
                # exp(-i theta x) at x = i z / theta, independent of the tabulated power-form coefficients
                import scipy.special

                x = np.clongdouble(1j) * z / np.longdouble(theta)
                T_prev, T_cur = np.clongdouble(1.0), x
                series = np.clongdouble(scipy.special.jv(0, theta))
                for k in range(1, m + 1):
                    series += 2 * np.clongdouble((-1j) ** k) * np.longdouble(scipy.special.jv(k, theta)) * T_cur
                    T_prev, T_cur = T_cur, 2 * x * T_cur - T_prev
                trunc = abs(complex(series - exact))
                # (jv is double precision: ~2e-16 of noise, more where |T_k(x)| has grown: theta > 4)
                assert trunc < (4.0 if theta <= 4.0 else 6.0) * 2.0 ** -53, (cap, defect, m, theta, trunc)
                # (b) the tabulated polynomial itself: truncation + the rounding of its double-precision coefficient
                # ratios, which grows like e^theta in the power form (the reason for the caps)
                assert err < 2.0 ** -53 * (2.0 + 0.5 * np.exp(theta)), (cap, defect, m, theta, err)
                checked += 1
            assert checked >= 3, (cap, defect)


def _ell_layout(ops, N):
    """kh_ell_layout on scipy.sparse operators (None: absent) -> (E, Ec, off [E][S], vals [n][E][S]), S = kh_ell_rows_of(N)."""
    import ctypes

    import numpy as np

    lib = _lib.load()
    arr = (_lib.kh_csr * len(ops))()
    keep = []
    for o, m in enumerate(ops):
        if m is None:
            continue
        indptr = np.ascontiguousarray(m.indptr, dtype=np.int32)
        indices = np.ascontiguousarray(m.indices, dtype=np.int32)
        data = np.ascontiguousarray(m.data, dtype=np.complex128)
        keep.append((indptr, indices, data))
        arr[o].nnz = len(data)
        arr[o].indptr, arr[o].indices, arr[o].data = indptr.ctypes.data, indices.ctypes.data, data.ctypes.data
    E, Ec = ctypes.c_int32(), ctypes.c_int32()
    rc = lib.kh_ell_layout(N, len(ops), arr, ctypes.byref(E), ctypes.byref(Ec), None, None, 0)
    if rc != 0:
        return rc, None, None, None
    S = lib.kh_ell_rows_of(N)
    assert S >= N and S in (512, 768, 1024, 1536, 2048)
    off = np.zeros((E.value, S), dtype=np.int32)
    vals = np.zeros((len(ops), E.value, S), dtype=np.complex128)
    assert lib.kh_ell_layout(N, len(ops), arr, ctypes.byref(E), ctypes.byref(Ec), off.ctypes.data, vals.ctypes.data, E.value) == 0
    return E.value, Ec.value, off, vals


def test_sparse_row_form_reproduces_the_operators():
    """kh_ell_layout (host code of the library, no GPU): the padded row form the sparse kernels keep in registers.  For
    a drift and two controls with different patterns (one absent in a second list), unsorted column indices and
    duplicate entries: sum_e vals[o][e][r] x[off[e][r] / 16] must be (A_o x)[r] for every operator, the entries the
    controls touch sit in the first Ec slots of EVERY row, E and Ec are multiples of four, rows are padded with their own
    row and value zero; too wide rows and N > 2048 are refused (those engines run the generic CSR kernels)."""
    import numpy as np
    import scipy.sparse as sp

    rng = np.random.default_rng(3)
    N = 300

    def rand_sparse(density):
        m = sp.random(N, N, density=density, random_state=np.random.RandomState(int(rng.integers(1 << 30))), format='coo')
        m = sp.coo_matrix((rng.standard_normal(m.nnz) + 1j * rng.standard_normal(m.nnz), (m.row, m.col)), shape=(N, N))
        return m

    A0, A1, A2 = rand_sparse(0.02), rand_sparse(0.008), rand_sparse(0.004)
    dup = sp.coo_matrix((np.concatenate([A0.data, A0.data[:50]]), (np.concatenate([A0.row, A0.row[:50]]),
                                                                   np.concatenate([A0.col, A0.col[:50]]))), shape=(N, N))
    unsorted = sp.csr_matrix(dup)  # duplicates summed by scipy ...
    raw = sp.csr_matrix(A0)
    raw.indices = raw.indices.copy()
    raw.has_sorted_indices = False
    for r in range(N):  # ... and column indices handed over in reverse order
        lo, hi = raw.indptr[r], raw.indptr[r + 1]
        raw.indices[lo:hi] = raw.indices[lo:hi][::-1].copy()
        raw.data[lo:hi] = raw.data[lo:hi][::-1].copy()
    for ops in ([raw, sp.csr_matrix(A1), sp.csr_matrix(A2)], [unsorted, None, sp.csr_matrix(A2)]):
        E, Ec, off, vals = _ell_layout(ops, N)
        assert E % 4 == 0 and Ec % 4 == 0 and 0 < Ec <= E <= 32
        x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
        cols = off[:, :N] // 16
        assert (off % 16 == 0).all() and cols.min() >= 0 and cols.max() < N
        assert (cols[:, :] == np.arange(N)[None, :])[np.abs(vals[:, :, :N]).sum(axis=0) == 0].all()  # padding: own row
        assert np.abs(vals[:, :, N:]).max() == 0.0
        for o, m in enumerate(ops):
            want = np.zeros(N, dtype=complex) if m is None else sp.csr_matrix(m) @ x
            got = (vals[o][:, :N] * x[cols]).sum(axis=0)
            assert np.abs(got - want).max() < 1e-13
        # nothing a control touches lies beyond slot Ec
        for o in range(1, len(ops)):
            assert np.abs(vals[o][Ec:]).max() == 0.0 if E > Ec else True
        # every column appears once per row among the non-padding slots
        for r in (0, 17, N - 1):
            live = cols[:, r][np.abs(vals[:, :, r]).sum(axis=0) > 0]
            assert len(set(live.tolist())) == len(live)
    dense_rows = sp.csr_matrix(np.ones((40, 40), dtype=complex))
    assert _ell_layout([dense_rows], 40)[0] == _lib.KH_ERR_UNSUPPORTED
    assert b'wider' in _lib.load().kh_last_error()
    wide16 = sp.csr_matrix(sp.random(600, 600, density=0.04, random_state=np.random.RandomState(1), format='csr') + sp.eye(600))
    assert _ell_layout([wide16.astype(complex)], 600)[0] == _lib.KH_ERR_UNSUPPORTED  # > 16 entries per row, N > 512
    # 1024 < N <= 2048: three / four rows per lane, at most 8 entries per row
    E, Ec, off, vals = _ell_layout([sp.eye(1025, format='csr', dtype=complex)], 1025)
    assert (E, off.shape) == (4, (4, 1536))
    E, Ec, off, vals = _ell_layout([sp.diags([1.0, 2.0, 3.0], [-1, 0, 40], shape=(1600, 1600), format='csr', dtype=complex)], 1600)
    assert (E, off.shape) == (4, (4, 2048)) and np.abs(vals[0, :, :1600].sum(axis=0) - (np.r_[0, np.ones(1599)] + 2 + np.r_[3 * np.ones(1560), np.zeros(40)])).max() == 0
    wide9 = sp.diags([1.0] * 9, list(range(9)), shape=(1100, 1100), format='csr', dtype=complex)
    assert _ell_layout([wide9], 1100)[0] == _lib.KH_ERR_UNSUPPORTED  # > 8 entries per row, N > 1024
    assert _ell_layout([sp.eye(2049, format='csr', dtype=complex)], 2049)[0] == _lib.KH_ERR_UNSUPPORTED
    assert _lib.load().kh_ell_rows_of(2049) == 0
