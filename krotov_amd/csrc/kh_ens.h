// krotov_amd/csrc/kh_ens.h -- update sweep for ENSEMBLES on the fp64 matrix cores (N <= 64, one control)
//
// The reference's ensemble_objectives (objectives.py:1054-1094; BASELINE config 5) are K copies of ONE control problem
// whose control operator is scaled: A_k(eps) = H0 + (s_k eps) H1.  The register-tile kernels treat them as K unrelated
// operator lists -- one objective per CU, and with more objectives than CUs the operators are streamed for every
// objective and interval (kh_tile64s.h: 128 MiB per interval at K = 1024).  Here the operators live ONCE per CU and the
// ensemble index is the second matrix dimension: one series term of the c objectives of a workgroup is
//
//     [Y0; Y1] = [H0; H1] X      (128 x 64)(64 x c), complex        T' = coef (Y0 + diag(s_k eps) Y1)
//
// on v_mfma_f64_4x4x4_4b.  Real arithmetic: with the columns of X as (re, im) pairs and X' = (-im, re),
// Y = M_re X + M_im X' -- a real (128 x 128)(128 x 2c) product, no wasted flops.  A wave owns rows [8w, 8w+8) of H0
// (blocks 0, 1 of the instruction) AND of H1 (blocks 2, 3): its 16 x 128 slice of [M_re | M_im] is 32 A operands = 64
// VGPRs per lane for the whole sweep; Y0 and Y1 of a row meet by one row rotation (DPP), the complex coefficient by one
// quad permutation.  The vectors ping-pong through LDS in B-operand order (8 KiB per pair of objectives and buffer), one
// barrier per term.  The first product of an interval, [H0; H1] phi(t_n), gives BOTH the update's <chi|H1 phi> and --
// once the pulse value is known -- the first term of the step.
//
//   optimize.py:444-508 for objectives that share a drift and a control operator up to a real scale, K > #CUs.
//
// Workgroup w owns the objectives [w CPW, (w + 1) CPW), CPW = 2 NCG (NCG column groups of two objectives = four real
// columns); NCG = 1, 2, 4, 8 serves K <= 512, 1024, 2048, 4096 with at most 256 co-resident workgroups.
#pragma once
#include "kh_tile64.h"

#define KH_ENS_THREADS 512
#define KH_ENS_WAVES 8
#define KH_ENS_MAXCG 8

struct KhEnsArgs {
    const cplx *H0;       // the shared drift, row-major N x N
    const cplx *H1;       // the reference control operator: objective k's is scale[k] * H1
    const double *scale;  // [K]
};

// vectors in LDS: xf[buffer][kk][pair][q][W] doubles; kk < 64: X (row kk), kk >= 64: X' (row kk - 64); q: real column
// within a column group (objective q >> 1, component q & 1); pairs of column groups side by side so that one
// ds_read_b128 fetches a lane's B operands of two groups
__host__ __device__ constexpr int kh_ens_np(int ncg) { return ncg == 1 ? 1 : ncg / 2; }
__host__ __device__ constexpr int kh_ens_w(int ncg) { return ncg == 1 ? 1 : 2; }
__host__ __device__ constexpr int kh_ens_buf(int ncg) { return 512 * ncg; }  // doubles per buffer
__host__ inline size_t kh_ens_lds_bytes(int ncg) { return (size_t)2 * kh_ens_buf(ncg) * sizeof(double); }
__host__ inline size_t kh_ens2_lds_bytes(int ncg) {  // + the vector s, + the [H0 | H1] A operands of the 8 waves
    return ((size_t)3 * kh_ens_buf(ncg) + (size_t)KH_ENS_WAVES * 32 * 64) * sizeof(double);
}

// Is objective k's operator list (H0, s_k H1_ref)?  flags[0]: some drift differs from objective 0's; flags[1]: some
// control operator is not a real multiple of objective 0's, element for element, to a few units in the last place
// (what "mu[k] * H1" in double precision produces -- configs.py config_c5, the reference's notebook 08: both operators
// are rounded products, and so is the ratio s taken from their largest element: up to 5 roundings, 2^-53 each).
__global__ void __launch_bounds__(256)
kh_ens_detect_kernel(const cplx *const *ops, int K, int N, int ref_idx, int ref_comp, double *scale, int *flags)
#if KH_DEFINES(KH_TU_MAIN)
{
    const int k = blockIdx.x;
    const cplx *H0 = ops[(size_t)k * 2], *H1 = ops[(size_t)k * 2 + 1], *R0 = ops[0], *R1 = ops[1];
    if (H1 == nullptr || R1 == nullptr) {
        if (threadIdx.x == 0) flags[1] = 1;
        return;
    }
    const double num = ref_comp ? H1[ref_idx].y : H1[ref_idx].x, den = ref_comp ? R1[ref_idx].y : R1[ref_idx].x;
    const double s = num / den;
    bool bad0 = false, bad1 = false;
    const double tol = 2e-15;
    for (int i = threadIdx.x; i < N * N; i += blockDim.x) {
        if (H0 != R0) {
            const cplx a = H0[i], b = R0[i];
            bad0 = bad0 || a.x != b.x || a.y != b.y;
        }
        const cplx a = H1[i], b = R1[i];
        const double ex = s * b.x, ey = s * b.y;
        bad1 = bad1 || !(fabs(a.x - ex) <= tol * fabs(ex)) || !(fabs(a.y - ey) <= tol * fabs(ey));
    }
    if (bad0) flags[0] = 1;
    if (bad1 || !(fabs(s) < 1e300)) flags[1] = 1;
    if (threadIdx.x == 0) scale[k] = s;
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif

template <int NCG, bool SO>
__global__ void __launch_bounds__(KH_ENS_THREADS, 2)
kh_ens_forward_update(KhSweepArgs p, KhEnsArgs en, KhUpdateArgs u, KhExchange ex) {
    constexpr int NP = kh_ens_np(NCG), W = kh_ens_w(NCG), CPW = 2 * NCG, BUF = kh_ens_buf(NCG);
    double *xf = (double *)kh_tile_dyn_lds;  // [2][BUF]
    __shared__ __attribute__((aligned(16))) double red[KH_ENS_WAVES];
    __shared__ __attribute__((aligned(16))) double D_sh[2][2];
    __shared__ __attribute__((aligned(16))) double ok_sh[2];
    __shared__ __attribute__((aligned(16))) double inv_sh[KH_MAX_DEGREE + 2];
    __shared__ __attribute__((aligned(16))) double deg_sh[KH_MAX_DEGREE + 2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int k4 = lane >> 4, blk = (lane >> 2) & 3, q = lane & 3;
    if (tid <= KH_MAX_DEGREE) deg_sh[tid] = p.q2_theta[tid];
    const int N = p.N, nt = p.nt, K = p.K;
    const int w = blockIdx.x;

    // ---- this wave's slice of [M_re | M_im] in A-operand order: lane (k4, blk, q) holds, for k-step ks, the element
    // (row 8 wave + 4 (blk & 1) + q of H0 (blk < 2) or H1, column kk = 4 ks + k4; kk >= 64: the imaginary parts)
    double af[32];
    {
        const cplx *M = blk < 2 ? en.H0 : en.H1;
        const int row = 8 * wave + 4 * (blk & 1) + q;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            const int kk = 4 * ks + k4, col = kk & 63;
            cplx v = c_make(0.0, 0.0);
            if (row < N && col < N) v = M[(size_t)row * N + col];
            af[ks] = kk >= 64 ? v.y : v.x;
        }
    }
    // ---- output side (D layout: lane = 16 row + 4 blk + column): this lane's element is component p = q & 1 of row r
    // of objective k_of(cg) = w CPW + 2 cg + (q >> 1) in Y0 (blk < 2: "owner" lanes) or Y1 (blk >= 2)
    const int r = 8 * wave + 4 * (blk & 1) + k4, pc = q & 1;
    const bool owner = blk < 2;
    double es[NCG], wgt[NCG], S[NCG];
    double n0 = 0.0, n1 = 0.0;
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) {
        const int k = w * CPW + 2 * cg + (q >> 1);
        const bool valid = k < K;
        es[cg] = valid ? en.scale[k] : 0.0;
        wgt[cg] = (valid && owner && r < N) ? u.chi_norms[k] * es[cg] : 0.0;
        S[cg] = (valid && r < N) ? ((const double *)u.phi)[((size_t)k * N + r) * 2 + pc] : 0.0;
    }
    for (int j = 0; j < CPW; ++j) {  // (uniform) the series' degree serves the workgroup's largest generator
        const int k = w * CPW + j;
        if (k < K) {
            n0 = fmax(n0, kh_uniform(p.op_norms[(size_t)k * 2]));
            n1 = fmax(n1, kh_uniform(p.op_norms[(size_t)k * 2 + 1]));
        }
    }
    const int lane_off = (k4 * NP * 4 + q) * W;
    // element (kk, cg, q) of a buffer
    auto xidx = [&](int kk, int cg, int qq) { return ((kk * NP + (cg >> 1)) * 4 + qq) * W + (cg & 1); };
    // owner lanes: t -> X (row r) and its rotated copy X' (row 64 + r; (re, im) -> (-im, re)) of buffer b
    auto write_x = [&](int b, const double (&t)[NCG]) {
        if (owner) {
            double *base = xf + b * BUF;
            if constexpr (W == 2) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const double2 v = make_double2(t[2 * j], t[2 * j + 1]);
                    *(double2 *)&base[xidx(r, 2 * j, q)] = v;
                    *(double2 *)&base[xidx(64 + r, 2 * j, q ^ 1)] = pc ? make_double2(-v.x, -v.y) : v;
                }
            } else {
                base[xidx(r, 0, q)] = t[0];
                base[xidx(64 + r, 0, q ^ 1)] = pc ? -t[0] : t[0];
            }
        }
    };
    // y[cg] <- this lane's element of [H0; H1] X for the vectors in buffer b (32 k-steps on the matrix core)
    auto pass = [&](int b, double (&y)[NCG]) {
        const double *src = xf + b * BUF + lane_off;
        if constexpr (W == 2) {  // (NCG independent accumulator chains per wave, two waves per SIMD: the pipe stays fed)
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) y[cg] = 0.0;
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const double2 v = *(const double2 *)(src + ks * 16 * NP * W + j * 4 * W);
                    y[2 * j] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[ks], v.x, y[2 * j], 0, 0, 0);
                    y[2 * j + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[ks], v.y, y[2 * j + 1], 0, 0, 0);
                }
            }
        } else {  // one column group: two chains per wave (a dependent 4x4x4 issues every 47 cycles, an independent one every 19)
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int ks = 0; ks < 32; ks += 2) {
                a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(af[ks], src[ks * 16], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(af[ks + 1], src[(ks + 1) * 16], a1, 0, 0, 0);
            }
            y[0] = a0 + a1;
        }
    };
    // t = coef (Y0 + s_k eps Y1), coef = (cr, ci) complex; cis = +-ci by component
    auto combine = [&](const double (&y)[NCG], double eps, double cr, double cis, double (&t)[NCG]) {
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
            const double y1 = dpp_move<KH_DPP_ROR8>(y[cg]);
            const double a = fma(es[cg] * eps, y1, y[cg]);
            const double other = dpp_move<KH_DPP_XOR1>(a);
            t[cg] = fma(cis, other, cr * a);
        }
    };

    // Im(mu conj(bra) y) (mu = 1) = bra.x y_im - bra.y y_re, Re(conj(bra) y) (mu = i) = bra.x y_re + bra.y y_im: this lane holds
    // component pc of y, so it needs ONE component of the bra chi_k(t_n)[r] (second order: and of the state under the
    // guess pulses), fetched one interval ahead: component gc, with sign gs
    const bool mu_real = u.mu_im == 0.0;
    const int gc = mu_real ? 1 - pc : pc;
    const double gs = mu_real ? (pc ? u.mu_re : -u.mu_re) : u.mu_im;
    double chi[NCG], prev[NCG], icn[NCG];
    auto load_chi = [&](int n) {
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
            const int k = w * CPW + 2 * cg + (q >> 1);
            const bool in = owner && k < K && r < N;
            chi[cg] = in ? ((const double *)u.chi_store)[(((size_t)k * nt + n) * N + r) * 2 + gc] : 0.0;
            if constexpr (SO) prev[cg] = in ? ((const double *)u.fw_prev)[(((size_t)k * nt + n) * N + r) * 2 + gc] : 0.0;
        }
    };
    if constexpr (SO) {
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
            const int k = w * CPW + 2 * cg + (q >> 1);
            icn[cg] = (owner && k < K) ? 1.0 / u.chi_norms[k] : 0.0;
        }
    }

    int cur = 0;
    write_x(0, S);
    if (u.n_begin < nt - 1) load_chi(u.n_begin);
    __syncthreads();

    double matvecs = 0.0, g_a_loc = 0.0;
    const int n_valid = (K - w * CPW) < CPW ? (K - w * CPW > 0 ? K - w * CPW : 0) : CPW;
    int m_loaded = -1;
    const double cis_sign = pc ? 1.0 : -1.0;

    for (int n = u.n_begin; n < u.n_end; ++n) {
        const int par = n & 1;
        double y[NCG];
        pass(cur, y);  // [H0; H1] phi(t_n)
        matvecs += 2.0 * n_valid;
        // ---- this workgroup's piece of sum_k ||chi_k|| Im(mu <chi_k(t_n) | s_k H1 phi_k(t_n)>)  (optimize.py:466-470)
        {
            double v = 0.0;
            double hs_n = 0.0;
            if constexpr (SO) hs_n = 0.5 * kh_uniform(u.sigma[n]);
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) {
                const double y1 = dpp_move<KH_DPP_ROR8>(y[cg]);
                double bra = chi[cg];
                if constexpr (SO) {  // chi + 0.5 sigma / ||chi|| (phi - phi_prev)  (optimize.py:468-469)
                    const double other = dpp_move<KH_DPP_XOR1>(S[cg]);
                    bra = fma(hs_n * icn[cg], (gc == pc ? S[cg] : other) - prev[cg], bra);
                }
                v = fma(wgt[cg] * (gs * bra), y1, v);
            }
            const double s = sum64_mfma(v);
            if (lane == 0) red[wave] = s;
        }
        if constexpr (SO) {  // phi_k(t_n) -> the stored trajectory
            if (owner && r < N) {
#pragma unroll
                for (int cg = 0; cg < NCG; ++cg) {
                    const int k = w * CPW + 2 * cg + (q >> 1);
                    if (k < K) ((double *)u.fw_store)[(((size_t)k * nt + n) * N + r) * 2 + pc] = S[cg];
                }
            }
        }
        __syncthreads();
        // ---- cross-objective sum (optimize.py:470) ----
        if (wave == 0) {
            double part[1], D[1];
            part[0] = 0.0;
#pragma unroll
            for (int ww = 0; ww < KH_ENS_WAVES; ++ww) part[0] += red[ww];
            const bool ok = kh_exchange<1, KH_GATHER_CHUNKS, true>(ex, n, w, 1, lane, part, D);
            if (lane == 0) {
                D_sh[par][0] = D[0];
                ok_sh[par] = ok ? 1.0 : 0.0;
            }
        }
        const double dt = kh_uniform(p.dt[n]);
        const double guess = kh_uniform(u.guess[n]);
        const double stp = kh_uniform(u.shape[n]) / kh_uniform(u.lambda[0]);
        if (n + 1 < nt - 1) load_chi(n + 1);  // lands while the series runs
        __syncthreads();
        if (ok_sh[par] == 0.0) return;
        // ---- pulse update (optimize.py:471-477) ----
        const double d1 = D_sh[par][0];
        const double eps = kh_uniform(guess + stp * d1);
        g_a_loc = kh_uniform(g_a_loc + stp * (d1 * d1) * dt);
        if (w == 0 && tid == 0) u.opt[n] = eps;
        // ---- the workgroup's objectives over interval n with the updated pulse (optimize.py:479-491) ----
        int nsub, m;
        kh_degree_lookup((n0 + fabs(eps) * n1) * dt, deg_sh, p.theta_max, p.inv_theta_max, m_loaded < 1 ? 12 : m_loaded, &nsub, &m);
        if (m != m_loaded) kh_tile_load_ratios(p, inv_sh, m, tid);  // (workgroup-uniform, rare)
        m_loaded = m;
        const double h = nsub == 1 ? dt : dt / nsub;
        for (int sub = 0; sub < nsub; ++sub) {
            if (sub > 0) {
                pass(cur, y);
                matvecs += 2.0 * n_valid;
            }
            const double c0 = inv_sh[0];
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) S[cg] *= c0;
            for (int j = 1; j <= m; ++j) {
                if (j > 1) {
                    pass(cur, y);
                    matvecs += 2.0 * n_valid;
                }
                const double hj = h * inv_sh[j];
                double t[NCG];
                combine(y, eps, p.fre * hj, cis_sign * p.fim * hj, t);
#pragma unroll
                for (int cg = 0; cg < NCG; ++cg) S[cg] += t[cg];
                if (j == m)
                    write_x(cur ^ 1, S);
                else
                    write_x(cur ^ 1, t);
                __syncthreads();
                cur ^= 1;
            }
        }
    }
    // running states back to the engine workspace
    if (owner && r < N) {
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
            const int k = w * CPW + 2 * cg + (q >> 1);
            if (k < K) {
                ((double *)u.phi)[((size_t)k * N + r) * 2 + pc] = S[cg];
                if constexpr (SO) ((double *)u.fw_store)[(((size_t)k * nt + u.n_end) * N + r) * 2 + pc] = S[cg];
            }
        }
    }
    if (w == 0 && tid == 0) u.g_a[0] = g_a_loc;
    if (tid == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}

// ---------------------------------------------------------------------------
// The same sweep on the A^2 chain (round 6; first order; dispatched for NCG = 2: 512 < K <= 1024)
// ---------------------------------------------------------------------------
// A_k(eps)^2 = P0 + (s_k eps) P1 + (s_k eps)^2 P2 with P0 = H0 H0, P1 = H0 H1 + H1 H0, P2 = H1 H1 of the ensemble's
// reference pair (the q2 kernels' tables of objective 0).  One pass of the even chain is [Z0; Z1; Z2] = [P0; P1; P2] X --
// 1.5 x the MFMAs of an [H0; H1] pass, for TWO terms of the series: t_{2p+2} = c2_p (Z0 + s eps Z1 + (s eps)^2 Z2) with a
// REAL coefficient (f^2 = -+1: no complex rotation); the odd terms by linearity, sum_p t_{2p+1} = f A s with
// s = sum_p h r1_p t_{2p}: ONE [H0; H1] pass next to the last even one (kh_tile64q2.h, kh_q2_expm_action).  A step of
// degree 12 is 6 x 1.5 + 1 = 10 pass units on 6 barriers instead of 12 on 12.
//   * the wave's rows [8w, 8w+8) of P0 | P1 (blocks 0,1 | 2,3: as H0 | H1 above), of P2 with the k range split over the
//     block pairs (blocks 0,1: k-steps 0..15, blocks 2,3: 16..31 -- half the instructions; the halves meet by the same row
//     rotation that brings Z1 to Z0's lanes), and of H0 | H1: 32 + 16 + 32 A operands = 160 VGPRs, whole sweep;
//   * the update sums on the adjoint side: <chi_k | s_k H1 phi_k> = <(s_k H1)^+ chi_k | phi_k> with the left factor
//     formed for the whole co-state store in front of the sweep (kh_gen_adjoint_side with the objectives' own adjoint
//     operators: u.adj_store, [K][nt][N]) -- a dot product with the state the lanes hold anyway, taken in the last phase's
//     shadow; no [H0; H1] phi pass at all;
//   * the interval's first even pass, [P0; P1; P2] phi(t_n), does not depend on the pulse: it runs WHILE the sums are
//     exchanged (wave 0 publishes, multiplies, collects), only its combination waits for eps(t_n).
template <int NCG>
__global__ void __launch_bounds__(KH_ENS_THREADS)
kh_ens2_forward_update(KhSweepArgs p, KhEnsArgs en, const cplx *const *__restrict__ sq /*P0, P1, P2 of the reference pair*/,
                       KhUpdateArgs u, KhExchange ex) {
    static_assert(NCG == 1 || NCG == 2, "more column groups: kh_ens_forward_update");
    constexpr int NP = kh_ens_np(NCG), W = kh_ens_w(NCG), CPW = 2 * NCG, BUF = kh_ens_buf(NCG);
    double *xf = (double *)kh_tile_dyn_lds;  // [3][BUF]: the term vector's two buffers, then s
    double *xs = xf + 2 * BUF;
    double *afc_lds = xf + 3 * BUF + (size_t)(threadIdx.x >> 6) * 32 * 64 + (threadIdx.x & 63);  // [wave][ks][lane]: this lane's column
    __shared__ __attribute__((aligned(16))) double red[2][KH_ENS_WAVES];
    __shared__ __attribute__((aligned(16))) double D_sh[2][2];  // [parity][value, ok]
    __shared__ __attribute__((aligned(16))) double2 rows_sh[KH_Q2_ROWS + 1];  // {r1_p, r2_p} of the degree in use; [ROWS].x: c_0
    __shared__ __attribute__((aligned(16))) double deg_sh[KH_MAX_DEGREE + 2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int k4 = lane >> 4, blk = (lane >> 2) & 3, q = lane & 3;
    if (tid <= KH_MAX_DEGREE) deg_sh[tid] = p.q2_theta[tid];
    const int N = p.N, nt = p.nt, K = p.K;
    const int w = blockIdx.x;

    // ---- A operands (layout as in kh_ens_forward_update): [P0 | P1], P2 split in k, [H0 | H1]
    // ([H0 | H1] is needed once per step: its operands wait in LDS, lane-linear, instead of in 64 more registers -- with
    // all three sets resident the kernel sat at 256 VGPRs and the passes' B-operand loads were no longer in flight together)
    double afA[32], afB[16];
    {
        const int row = 8 * wave + 4 * (blk & 1) + q;
        const cplx *MA = blk < 2 ? sq[0] : sq[1], *MB = sq[2], *MC = blk < 2 ? en.H0 : en.H1;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            const int kk = 4 * ks + k4, col = kk & 63;
            cplx va = c_make(0.0, 0.0), vc = c_make(0.0, 0.0);
            if (row < N && col < N) {
                va = MA[(size_t)row * N + col];
                vc = MC[(size_t)row * N + col];
            }
            afA[ks] = kk >= 64 ? va.y : va.x;
            afc_lds[ks * 64] = kk >= 64 ? vc.y : vc.x;
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int kk = 4 * (ks + (blk >= 2 ? 16 : 0)) + k4, col = kk & 63;
            cplx vb = c_make(0.0, 0.0);
            if (row < N && col < N) vb = MB[(size_t)row * N + col];
            afB[ks] = kk >= 64 ? vb.y : vb.x;
        }
    }
    const int r = 8 * wave + 4 * (blk & 1) + k4, pc = q & 1;
    const bool owner = blk < 2;
    double es[NCG], wgt[NCG], S[NCG];
    double n0 = 0.0, n1 = 0.0;
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) {
        const int k = w * CPW + 2 * cg + (q >> 1);
        const bool valid = k < K;
        es[cg] = valid ? en.scale[k] : 0.0;
        wgt[cg] = (valid && owner && r < N) ? u.chi_norms[k] : 0.0;  // (s_k sits in the adjoint-side store)
        S[cg] = (valid && r < N) ? ((const double *)u.phi)[((size_t)k * N + r) * 2 + pc] : 0.0;
    }
    for (int j = 0; j < CPW; ++j) {  // (uniform) the series' degree serves the workgroup's largest generator
        const int k = w * CPW + j;
        if (k < K) {
            n0 = fmax(n0, kh_uniform(p.op_norms[(size_t)k * 2]));
            n1 = fmax(n1, kh_uniform(p.op_norms[(size_t)k * 2 + 1]));
        }
    }
    const int lane_off = (k4 * NP * 4 + q) * W;
    auto xidx = [&](int kk, int cg, int qq) { return ((kk * NP + (cg >> 1)) * 4 + qq) * W + (cg & 1); };
    // owner lanes: t -> X (row r) and its rotated copy X' (row 64 + r; (re, im) -> (-im, re)) of the buffer at `base`
    auto write_x = [&](double *base, const double (&t)[NCG]) {
        if (owner) {
            if constexpr (W == 2) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const double2 v = make_double2(t[2 * j], t[2 * j + 1]);
                    *(double2 *)&base[xidx(r, 2 * j, q)] = v;
                    *(double2 *)&base[xidx(64 + r, 2 * j, q ^ 1)] = pc ? make_double2(-v.x, -v.y) : v;
                }
            } else {
                base[xidx(r, 0, q)] = t[0];
                base[xidx(64 + r, 0, q ^ 1)] = pc ? -t[0] : t[0];
            }
        }
    };
    // one pass over the k-steps of the A operands af (32, or 16: P2): y[cg] <- this lane's element of the product
    auto pass = [&](const double *src, const auto &af, double (&y)[NCG]) {
        constexpr int nk = (int)(sizeof(af) / sizeof(double));
        if constexpr (W == 2) {
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) y[cg] = 0.0;
#pragma unroll
            for (int ks = 0; ks < nk; ++ks) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const double2 v = *(const double2 *)(src + ks * 16 * NP * W + j * 4 * W);
                    y[2 * j] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[ks], v.x, y[2 * j], 0, 0, 0);
                    y[2 * j + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[ks], v.y, y[2 * j + 1], 0, 0, 0);
                }
            }
        } else {
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int ks = 0; ks < nk; ks += 2) {
                a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(af[ks], src[ks * 16], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(af[ks + 1], src[(ks + 1) * 16], a1, 0, 0, 0);
            }
            y[0] = a0 + a1;
        }
    };

    // (-DKH_ENS2_FUSED, measured and not the default:) the even pass [P0; P1; P2] X in ONE loop over 16 k-step pairs
    // (ks, ks + 16): the two halves of the [P0 | P1] product
    // and the P2 product are independent accumulator chains (a dependent 4x4x4 issues every 47 cycles, an independent one
    // every 19), and P2's B operand is the vector element one of the two [P0 | P1] loads has fetched already -- k-step ks
    // for the block pair 0,1, ks + 16 for 2,3: a select instead of a third LDS read (the reads, 8 bytes per lane and MFMA,
    // are what bounds the term-by-term kernel).
    const bool upper = blk >= 2;
    auto pass3 = [&](const double *src, double (&zA)[NCG], double (&zB)[NCG]) {
#ifndef KH_ENS2_FUSED  // default: the two products as two loops, P2's operands read from LDS again (K = 1024: 15.9 us
        // per interval against 16.4 with the fused loop below, 16.95 term by term; K = 512, one column group: 11.4 /
        // 10.7 / 10.5 -- which is why only NCG = 2 is dispatched to this kernel: profiles/r06/exp_ens2.txt)
        pass(src, afA, zA);
        pass(src + (upper ? 16 * 16 * NP * W : 0), afB, zB);
        return;
#endif
        if constexpr (W == 2) {
            double zA2[NCG];
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) zA[cg] = zA2[cg] = zB[cg] = 0.0;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const double2 v0 = *(const double2 *)(src + ks * 16 * NP * W + j * 4 * W);
                    const double2 v1 = *(const double2 *)(src + (ks + 16) * 16 * NP * W + j * 4 * W);
                    const double2 vb = upper ? v1 : v0;
                    zA[2 * j] = __builtin_amdgcn_mfma_f64_4x4x4f64(afA[ks], v0.x, zA[2 * j], 0, 0, 0);
                    zA[2 * j + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(afA[ks], v0.y, zA[2 * j + 1], 0, 0, 0);
                    zA2[2 * j] = __builtin_amdgcn_mfma_f64_4x4x4f64(afA[ks + 16], v1.x, zA2[2 * j], 0, 0, 0);
                    zA2[2 * j + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(afA[ks + 16], v1.y, zA2[2 * j + 1], 0, 0, 0);
                    zB[2 * j] = __builtin_amdgcn_mfma_f64_4x4x4f64(afB[ks], vb.x, zB[2 * j], 0, 0, 0);
                    zB[2 * j + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(afB[ks], vb.y, zB[2 * j + 1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) zA[cg] += zA2[cg];
        } else {
            double a0 = 0.0, a1 = 0.0, b0 = 0.0;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const double v0 = src[ks * 16], v1 = src[(ks + 16) * 16];
                a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(afA[ks], v0, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(afA[ks + 16], v1, a1, 0, 0, 0);
                b0 = __builtin_amdgcn_mfma_f64_4x4x4f64(afB[ks], upper ? v1 : v0, b0, 0, 0, 0);
            }
            zA[0] = a0 + a1;
            zB[0] = b0;
        }
    };

    // the bra's component this lane needs (see kh_ens_forward_update) -- of V_k(t_n) = (s_k H1)^+ chi_k(t_n)
    const bool mu_real = u.mu_im == 0.0;
    const int gc = mu_real ? 1 - pc : pc;
    const double gs = mu_real ? (pc ? u.mu_re : -u.mu_re) : u.mu_im;
    double vb[NCG];
    auto load_bra = [&](int n) {
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
            const int k = w * CPW + 2 * cg + (q >> 1);
            const bool in = owner && k < K && r < N;
            vb[cg] = in ? ((const double *)u.adj_store)[(((size_t)k * nt + n) * N + r) * 2 + gc] : 0.0;
        }
    };
    // this workgroup's piece of sum_k ||chi_k|| Im(mu <V_k(t_n) | phi_k(t_n)>) -> red[par][wave]  (optimize.py:466-470)
    auto sums = [&](int par) {
        double v = 0.0;
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) v = fma(wgt[cg] * (gs * vb[cg]), S[cg], v);
        const double sv = sum64_mfma(v);
        if (lane == 0) red[par][wave] = sv;
    };

    int cur = 0;
    write_x(xf, S);
    if (u.n_begin < nt - 1) {
        load_bra(u.n_begin);
        sums(u.n_begin & 1);
    }
    __syncthreads();

    double matvecs = 0.0, g_a_loc = 0.0;
    const int n_valid = (K - w * CPW) < CPW ? (K - w * CPW > 0 ? K - w * CPW : 0) : CPW;
    int m_loaded = -1;
    const double cis_sign = pc ? 1.0 : -1.0;
    const double f2 = p.fre * p.fre - p.fim * p.fim;  // f is purely real or purely imaginary

    for (int n = u.n_begin; n < u.n_end; ++n) {
        const int par = n & 1;
        // ---- cross-objective sum (optimize.py:470), with the step's first even pass in its shadow ----
        double part[1] = {0.0};
        if (wave == 0) {
#pragma unroll
            for (int ww = 0; ww < KH_ENS_WAVES; ++ww) part[0] += red[par][ww];
            kh_exchange_publish(ex, n, w, 1, lane, part);
        }
        double zA[NCG], zB[NCG];
        pass3(xf + cur * BUF + lane_off, zA, zB);  // [P0; P1; P2] phi(t_n)
        matvecs += 3.0 * n_valid;
        if (wave == 0) {
            double D[1];
            const bool ok = kh_exchange_collect<1, KH_GATHER_CHUNKS, true>(ex, n, w, 1, lane, part, D);
            if (lane == 0) {
                D_sh[par][0] = D[0];
                D_sh[par][1] = ok ? 1.0 : 0.0;
            }
        }
        const double dt = kh_uniform(p.dt[n]);
        const double guess = kh_uniform(u.guess[n]);
        const double stp = kh_uniform(u.shape[n]) / kh_uniform(u.lambda[0]);
        if (n + 1 < nt - 1) load_bra(n + 1);  // lands while the series runs
        __syncthreads();
        const double2 sums_n = *(const double2 *)D_sh[par];
        if (sums_n.y == 0.0) return;
        // ---- pulse update (optimize.py:471-477) ----
        const double d1 = sums_n.x;
        const double eps = kh_uniform(guess + stp * d1);
        g_a_loc = kh_uniform(g_a_loc + stp * (d1 * d1) * dt);
        if (w == 0 && tid == 0) u.opt[n] = eps;
        // ---- the workgroup's objectives over interval n with the updated pulse (optimize.py:479-491) ----
        int nsub, m;
        kh_degree_lookup((n0 + fabs(eps) * n1) * dt, deg_sh, p.theta_max, p.inv_theta_max, m_loaded < 1 ? 12 : m_loaded, &nsub, &m);
        if (m != m_loaded) {  // (workgroup-uniform, rare) the series' rows of degree m
            __syncthreads();
            if (tid < KH_Q2_ROWS) {
                const double *rr = p.q2_rows + ((size_t)m * KH_Q2_ROWS + tid) * 2;
                rows_sh[tid] = make_double2(rr[0], rr[1]);
            }
            if (tid == KH_Q2_ROWS) rows_sh[KH_Q2_ROWS] = make_double2(p.q2_c0[m], 0.0);
            __syncthreads();
        }
        m_loaded = m;
        const double h = nsub == 1 ? dt : dt / nsub;
        const double f2h2 = f2 * h * h;
        const int phases = (m + 1) >> 1;
        for (int sub = 0; sub < nsub; ++sub) {
            if (sub > 0) {
                pass3(xf + cur * BUF + lane_off, zA, zB);
                matvecs += 3.0 * n_valid;
            }
            // s = sum_p h r1_p T_2p (T_0 = c_0 v; r1_0 relative to v itself), the state sum starts from c_0 v
            const double hr0 = h * rows_sh[0].x, c0 = rows_sh[KH_Q2_ROWS].x;
            double sacc[NCG];
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) {
                sacc[cg] = hr0 * S[cg];
                S[cg] *= c0;
            }
            if (phases == 1) {  // (degree <= 2) s is final already
                write_x(xs, sacc);
                __syncthreads();
            }
            for (int ph = 0; ph < phases; ++ph) {
                if (ph > 0) {
                    pass3(xf + cur * BUF + lane_off, zA, zB);
                    matvecs += 3.0 * n_valid;
                }
                const double c2 = f2h2 * rows_sh[ph].y;
                const bool last = ph + 1 == phases;
                double t2[NCG];
#pragma unroll
                for (int cg = 0; cg < NCG; ++cg) {
                    const double se = es[cg] * eps;
                    const double z1 = dpp_move<KH_DPP_ROR8>(zA[cg]);
                    const double z2 = zB[cg] + dpp_move<KH_DPP_ROR8>(zB[cg]);
                    t2[cg] = c2 * fma(se, fma(se, z2, z1), zA[cg]);
                    S[cg] += t2[cg];
                }
                if (!last) {
                    const double hn = h * rows_sh[ph + 1].x;
#pragma unroll
                    for (int cg = 0; cg < NCG; ++cg) sacc[cg] = fma(hn, t2[cg], sacc[cg]);
                    write_x(xf + (cur ^ 1) * BUF, t2);
                    if (ph + 2 == phases) write_x(xs, sacc);  // s is complete: the next phase multiplies it by A
                } else {
                    double yC[NCG];
                    double afC[32];
#pragma unroll
                    for (int ks = 0; ks < 32; ++ks) afC[ks] = afc_lds[ks * 64];
                    pass(xs + lane_off, afC, yC);  // [H0; H1] s
                    matvecs += 2.0 * n_valid;
#pragma unroll
                    for (int cg = 0; cg < NCG; ++cg) {
                        const double a = fma(es[cg] * eps, dpp_move<KH_DPP_ROR8>(yC[cg]), yC[cg]);
                        const double other = dpp_move<KH_DPP_XOR1>(a);
                        S[cg] += fma(cis_sign * p.fim, other, p.fre * a);  // f A s
                    }
                    write_x(xf + (cur ^ 1) * BUF, S);
                    if (sub + 1 == nsub && n + 1 < nt - 1) sums((n + 1) & 1);  // rides this phase's barrier
                }
                __syncthreads();
                cur ^= 1;
            }
        }
    }
    // running states back to the engine workspace
    if (owner && r < N) {
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
            const int k = w * CPW + 2 * cg + (q >> 1);
            if (k < K) ((double *)u.phi)[((size_t)k * N + r) * 2 + pc] = S[cg];
        }
    }
    if (w == 0 && tid == 0) u.g_a[0] = g_a_loc;
    if (tid == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}
