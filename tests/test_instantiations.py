"""One oracle comparison for every sweep-kernel template instantiation the other GPU tests do not reach by themselves
(VERDICT r4 item 1): each case names the instantiations it must dispatch to, the library's own registry
(``kh_debug_launched``) confirms that they ran, and both sweeps are compared with the oracle at 1e-12 (1e-11 on the
stiff Liouvillians).  tests/test_zz_kernel_coverage.py then demands that NO dispatchable instantiation is left without
such a test.  Reference: optimize.py:444-508 (update sweep), :849-886 (backward sweep)."""
import numpy as np
import pytest

from helpers import oracle_controls, spec_to_oracle
from krotov_amd import configs
from oracle import krotov_oracle as ko

pytestmark = pytest.mark.gpu


def _banded(N, bands, nt, K=2):
    from test_hip_parity import _banded as make

    return make(N, bands, nt, K=K)


def _drop_controls(spec, which):
    """Objectives `which` lose their control operator: a zero matrix for the oracle, ``None`` (absent) for the engine."""
    for k in which:
        spec.Hc[k] = [np.zeros_like(spec.Hc[k][0])]
    spec.absent_controls = set(which)
    return spec


def _shared(K, N, L):
    return configs.config_shared(K=K, N=N, nt=4, L=L)


CASES = {}


def case(name, build, expect, env=None, so=False, sparse=False):
    CASES[name] = (build, tuple(expect), dict(env or {}), so, sparse)


# ---- register-generator kernels, 64 < N <= 128 (kh_tilen.h): <elements per lane, [second order,] H1 in registers>
case('tn_n70', lambda: configs.config_c5(K=3, N=70, nt=9, L=1), ['kh_tn_sweep_store<20, true>', 'kh_tn_forward_update<20, false, true>'])
case('tn_n70_so', lambda: configs.config_c5(K=3, N=70, nt=9, L=1), ['kh_tn_forward_update<20, true, true>'], so=True)
case('tn_n90_L2', lambda: configs.config_c5(K=3, N=90, nt=9, L=2), ['kh_tn_sweep_store<24, false>', 'kh_tn_forward_update<24, false, false>'])
case('tn_n90_L2_so', lambda: configs.config_c5(K=3, N=90, nt=9, L=2), ['kh_tn_forward_update<24, true, false>'], so=True)
case('tn_n128_so', lambda: configs.config_c5(K=2, N=128, nt=7, L=1), ['kh_tn_forward_update<32, true, false>'], so=True)
case('tn_n110_so', lambda: configs.config_c5(K=2, N=110, nt=7, L=1), ['kh_tn_forward_update<28, true, false>'], so=True)

# ---- sparse operators in registers (kh_ell.h): <threads, rows per lane, widest row[, second order]>
case('ell_e11', lambda: _banded(40, 11, nt=9), ['kh_ell_sweep_store<512, 1, 12, false>', 'kh_ell_forward_update<512, 1, 12, false, false>'], sparse=True)
case('ell_e11_so', lambda: _banded(40, 11, nt=9), ['kh_ell_forward_update<512, 1, 12, true, false>'], sparse=True, so=True)
case('ell_n600_e11', lambda: _banded(600, 11, nt=4, K=1), ['kh_ell_sweep_store<768, 1, 12, false>', 'kh_ell_forward_update<768, 1, 12, false, false>'], sparse=True)
case('ell_e15_so', lambda: _banded(40, 15, nt=9), ['kh_ell_forward_update<512, 1, 16, true, false>'], sparse=True, so=True)
case('ell_e21', lambda: _banded(40, 21, nt=9), ['kh_ell_sweep_store<512, 1, 24, false>', 'kh_ell_forward_update<512, 1, 24, false, false>'], sparse=True)
case('ell_e21_so', lambda: _banded(40, 21, nt=9), ['kh_ell_forward_update<512, 1, 24, true, false>'], sparse=True, so=True)
case('ell_e29', lambda: _banded(48, 29, nt=9), ['kh_ell_sweep_store<512, 1, 32, false>', 'kh_ell_forward_update<512, 1, 32, false, false>'], sparse=True)
case('ell_e29_so', lambda: _banded(48, 29, nt=9), ['kh_ell_forward_update<512, 1, 32, true, false>'], sparse=True, so=True)
case('ell_n800_e15', lambda: _banded(800, 15, nt=4, K=1), ['kh_ell_sweep_store<512, 2, 16, false>', 'kh_ell_forward_update<512, 2, 16, false, false>'], sparse=True)
case('ell_n800_e15_so', lambda: _banded(800, 15, nt=4, K=1), ['kh_ell_forward_update<512, 2, 16, true, false>'], sparse=True, so=True)
case('ell_n800_e11_so', lambda: _banded(800, 11, nt=4, K=1), ['kh_ell_forward_update<512, 2, 12, true, false>'], sparse=True, so=True)
case('ell_n600_e11_so', lambda: _banded(600, 11, nt=4, K=1), ['kh_ell_forward_update<768, 1, 12, true, false>'], sparse=True, so=True)
case('ell_n600_e15', lambda: _banded(600, 15, nt=4, K=1), ['kh_ell_sweep_store<768, 1, 16, false>', 'kh_ell_forward_update<768, 1, 16, false, false>'], sparse=True)
case('ell_n600_e15_so', lambda: _banded(600, 15, nt=4, K=1), ['kh_ell_forward_update<768, 1, 16, true, false>'], sparse=True, so=True)
case('ell_n625_so', lambda: configs.config_sparse_lindblad(d=25, nt=5, K=2), ['kh_ell_forward_update<768, 1, 8, true, false>'], sparse=True, so=True)
case('ell_n900_so', lambda: configs.config_sparse_lindblad(d=30, nt=4, K=1), ['kh_ell_forward_update<1024, 1, 8, true, false>'], sparse=True, so=True)

# 1024 < N <= 2048 with at most 8 entries per row: three / four rows per lane (a d = 40 Lindbladian: N = 1600)
case('ell_n1089', lambda: configs.config_sparse_lindblad(d=33, nt=3, K=1), ['kh_ell_sweep_store<512, 3, 8, false>', 'kh_ell_forward_update<512, 3, 8, false, false>'], sparse=True)
case('ell_n1089_so', lambda: configs.config_sparse_lindblad(d=33, nt=3, K=1), ['kh_ell_forward_update<512, 3, 8, true, false>'], sparse=True, so=True)
case('ell_n1600', lambda: configs.config_sparse_lindblad(d=40, nt=3, K=1), ['kh_ell_sweep_store<512, 4, 8, false>', 'kh_ell_forward_update<512, 4, 8, false, false>'], sparse=True)
case('ell_n1600_so', lambda: configs.config_sparse_lindblad(d=40, nt=3, K=1), ['kh_ell_forward_update<512, 4, 8, true, false>'], sparse=True, so=True)

# the STREAMED form of the same kernels (nothing resident: rows that do not fit the registers, N up to 4096)
case('ells_n600_e21', lambda: _banded(600, 21, nt=4, K=1), ['kh_ell_sweep_store<512, 8, 4, true>', 'kh_ell_forward_update<512, 8, 4, false, true>'], sparse=True)
case('ells_n600_e21_so', lambda: _banded(600, 21, nt=4, K=1), ['kh_ell_forward_update<512, 8, 4, true, true>'], sparse=True, so=True)
case('ells_forced_lindblad', lambda: configs.config_sparse_lindblad(d=12, nt=21, K=3), ['kh_ell_sweep_store<512, 8, 4, true>', 'kh_ell_forward_update<512, 8, 4, false, true>'],
     env={'KH_KERNEL': 'ellstream'}, sparse=True)
case('ells_forced_L3_so', lambda: configs.config_c5(K=5, N=12, nt=31, L=3, distinct=True), ['kh_ell_forward_update<512, 8, 4, true, true>'],
     env={'KH_KERNEL': 'ellstream'}, sparse=True, so=True)
case('ells_forced_n900', lambda: configs.config_sparse_lindblad(d=30, nt=4, K=2), ['kh_ell_sweep_store<512, 8, 4, true>', 'kh_ell_forward_update<512, 8, 4, false, true>'],
     env={'KH_KERNEL': 'ellstream'}, sparse=True)  # two rows per lane in use (N = 4096: test_sparse_liouvillian_of_dimension_4096)

# ---- streaming register-tile kernel (kh_tile64s.h): <controls, second order, N == 64>
_k520 = lambda: configs.config_c5(K=520, N=64, nt=6, distinct=True)  # noqa: E731  (three objectives per workgroup)
case('stream_L1_n64_so', _k520, ['kh_stream_forward_update<1, true, true>'], so=True)
# ... with a ragged tail (the last workgroups own one objective less) and objectives without the control operator
case('stream_L1_n64_ragged', lambda: _drop_controls(configs.config_c5(K=600, N=64, nt=5, distinct=True), (0, 257, 599)),
     ['kh_stream_forward_update<1, false, true>'], env={'KH_STREAM_G': '256'})  # 256 x 2 + 88: three and two per workgroup
case('stream_L3_n64', lambda: configs.config_c5(K=260, N=64, nt=4, L=3, distinct=True), ['kh_stream_forward_update<3, false, true>'])
case('stream_L3_n64_so', lambda: configs.config_c5(K=260, N=64, nt=4, L=3, distinct=True), ['kh_stream_forward_update<3, true, true>'], so=True)
case('stream_L3_n6_so', lambda: configs.config_c5(K=1100, N=6, nt=5, L=3), ['kh_stream_forward_update<3, true, false>'], so=True)
case('stream_L4_n64', lambda: configs.config_c5(K=260, N=64, nt=4, L=4, distinct=True), ['kh_stream_forward_update<4, false, true>'])
case('stream_L4_n64_so', lambda: configs.config_c5(K=260, N=64, nt=4, L=4, distinct=True), ['kh_stream_forward_update<4, true, true>'], so=True)
case('stream_L4_n8', lambda: configs.config_c5(K=270, N=8, nt=6, L=4), ['kh_stream_forward_update<4, false, false>'])
case('stream_L4_n8_so', lambda: configs.config_c5(K=270, N=8, nt=6, L=4), ['kh_stream_forward_update<4, true, false>'], so=True)

# ---- two-terms-per-phase kernels (kh_tile64q2.h): <second order, sums on the adjoint side, single GPU>; the forms with the
# cross-GPU stage run on one GPU with KH_Q2_SINGLE=0 (and across ranks in test_two_ranks_sharded_on_one_gpu)
case('q2_so_p2p_form', lambda: configs.config_c5(K=8, N=64, nt=21), ['kh_q2_forward_update<true, false, false>'], env={'KH_Q2_SINGLE': '0'}, so=True)
case('q2_fwd_side_p2p_form', lambda: configs.config_c5(K=8, N=64, nt=21), ['kh_q2_forward_update<false, false, false>'],
     env={'KH_Q2_SINGLE': '0', 'KH_NO_ADJ': '1'})

# ---- one-term-per-phase kernels (kh_tile64.h): <rows per thread, controls, second order, single GPU>
_L4 = lambda: configs.config_c5(K=4, N=64, nt=21, L=4, distinct=True)  # noqa: E731
case('tile512_L4_so', _L4, ['kh_tile_forward_update<1, 4, true, true>'], so=True)
case('tile512_L4_so_p2p_form', _L4, ['kh_tile_forward_update<1, 4, true, false>'], env={'KH_TILE_SINGLE': '0'}, so=True)

# ---- cooperative matrix-core kernels (kh_coop.h): <operator slots per lane, objectives per workgroup, second order, sums
# on the adjoint side, A^2 chain, cross-GPU stage>; slots 16: N > 256
for _ks, _N in ((8, 96), (16, 272)):
    for _cols, _K in ((2, 4), (4, 6), (16, 20)):
        _e = {'KH_COOP_COLS': str(_cols)}
        _t = 'kh_coop_forward_update<%d, %d, ' % (_ks, _cols)
        _n = 'coop%d_c%d_' % (_ks, _cols)
        _one = lambda K=_K, N=_N: _shared(K, N, 1)  # noqa: E731
        _two = lambda K=_K, N=_N: _shared(K, N, 2)  # noqa: E731
        if (_ks, _cols) == (16, 16):
            # the A^2 chain is not staged for this shape (its second resident fragment does not fit the registers): one
            # control runs the forms two controls run
            case(_n + 'L1', _one, [_t + 'false, false, false, true>', 'kh_coop_sweep_store<16, 16, false>'], env=_e)
            case(_n + 'L1_so', _one, [_t + 'true, false, false, true>'], env=_e, so=True)
        else:
            case(_n + 'adj', _one, [_t + 'false, true, true, false>', 'kh_coop_sweep_store<%d, %d, true>' % (_ks, _cols)], env=_e)
            case(_n + 'adj_p2p_form', _one, [_t + 'false, true, true, true>'], env=dict(_e, KH_COOP_SINGLE='0'))
            case(_n + 'so', _one, [_t + 'true, false, true, true>'], env=_e, so=True)
            case(_n + 'no_adj', _one, [_t + 'false, false, true, true>'], env=dict(_e, KH_COOP_NO_ADJ='1'))
        case(_n + 'L2', _two, [_t + 'false, false, false, true>', 'kh_coop_sweep_store<%d, %d, false>' % (_ks, _cols)], env=_e)
        case(_n + 'L2_so', _two, [_t + 'true, false, false, true>'], env=_e, so=True)


@pytest.mark.parametrize('name', sorted(CASES))
def test_instantiation_vs_oracle(name, monkeypatch):
    import torch

    from krotov_amd import _lib
    from krotov_amd.engine import HipKrotovEngine

    build, expect, env, so, sparse = CASES[name]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    spec = build()
    prob = spec_to_oracle(spec)
    gp, S, lam = oracle_controls(spec)
    pulses = np.array(gp)
    if sparse:
        ops = configs.sparse_ops(spec)
    else:
        absent = getattr(spec, 'absent_controls', ())
        ops = [[spec.H0[k]] + [None if k in absent else spec.Hc[k][l] for l in range(spec.L)] for k in range(spec.K)]
    eng = HipKrotovEngine(ops, np.diff(spec.tlist), is_super=spec.is_super)
    _lib.forget_launched_kernels()
    rng = np.random.default_rng(17)
    chi_T = spec.target / np.linalg.norm(spec.target, axis=1)[:, None]
    # (hundreds of objectives add up; ||H_l|| ~ 1e2 .. 1e3 with lambda_a = 2 under the shared-operator problems: keep the
    # updated pulses O(1) so that the problem stays well conditioned -- as test_second_order_update_sweep does)
    damp = min(1.0, 8.0 / spec.K) * (0.02 if spec.name.startswith('shared') else 1.0)
    norms = (0.2 + rng.random(spec.K)) * damp
    ref_chi = ko.backward_sweep(prob, chi_T, gp)
    chi = eng.backward(chi_T, pulses)
    kw = {}
    if so:
        older = [p * (1.0 + 0.2 * rng.standard_normal(p.shape)) for p in gp]  # the "previous iteration"
        _, prev = ko.forward_propagation(prob, older, store=True)
        sigma_vals = -(1.0 + rng.random(len(spec.tlist) - 1)) * min(1.0, 8.0 / spec.K) * (1e-3 if spec.name.startswith('shared') else 1.0)
        kw = dict(sigma_vals=sigma_vals, fw_prev=prev, store=True)
        store = torch.full((spec.K, len(spec.tlist), spec.N), float('nan'), dtype=torch.complex128, device=eng.device)
        eng.set_second_order(prev, store, sigma_vals)
    ref = ko.forward_update_sweep(prob, ref_chi, norms, gp, S, lam, **kw)
    opt, psi_T, g_a = eng.forward_update(chi, norms, spec.init, pulses, np.array(S), np.array(lam))
    eng.check()
    launched = _lib.kernel_instantiations(launched_only=True)
    tol = 1e-11 if spec.is_super else 1e-12
    scale = max(1.0, np.abs(np.array(ref[0])).max())
    assert np.abs(chi.cpu().numpy() - ref_chi).max() < tol
    assert np.abs(opt.cpu().numpy() - np.array(ref[0])).max() < tol * scale
    assert np.abs(psi_T.cpu().numpy() - ref[1]).max() < tol
    assert np.abs(g_a.cpu().numpy() - ref[2]).max() < tol * max(1.0, np.abs(ref[2]).max())
    if so:
        assert np.abs(store.cpu().numpy() - ref[3]).max() < tol
    for want in expect:
        assert want in launched, (want, launched)
    eng.close()
