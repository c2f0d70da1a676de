// Register-resident sweep kernels for state dimension N <= 64 (CDNA4 / gfx950).
//
// One workgroup per objective.  The step generator A = H0 + sum_l eps_l H_l and
// the operators it is rebuilt from stay in VGPRs for the whole sweep (a 64x64
// complex128 operator is 64 KiB = 64 dwords/lane over 256 lanes, 32 over 512);
// only the Taylor-term vector goes through LDS (1 KiB, double-buffered), read as
// conflict-free broadcast ds_read_b128s.  Each lane owns RPT rows x 8 columns
// (columns cg, cg+8, ... so the 8 column-group lanes of a row hit 8 consecutive
// 16-byte LDS slots); row sums are all-reduced over the 8 adjacent lanes with
// DPP (quad_perm, quad_perm, row_half_mirror), never through LDS.
//
//   RPT = 2 : 256 threads, 4 waves, 16 rows per wave
//   RPT = 1 : 512 threads, 8 waves,  8 rows per wave (half the registers/lane)
//
// Stored states are written as one coalesced 1 KiB store per interval.
#pragma once

#include "kh_common.h"
#include "kh_generic.h"

#define KH_TILE_N 64

// Lane roles inside a wave (8 rows x 8 column groups), in two flavours:
//   KhLanes<false>: the row's 8 column groups are 8 adjacent lanes; row sums are a three-step DPP butterfly
//     (quad_perm, quad_perm, row_half_mirror: 18 VALU instructions for a complex value).
//   KhLanes<true>: the matrix core adds them.  v_mfma_f64_4x4x4 with a constant B operand sums its A operand
//     over the four lanes 16 k + i (k = 0..3), so the column group's low bits are lane >> 4; the result for
//     i = 4 blk + r4 lands on lanes 16 r4 + 4 blk + (0..3), and ONE DPP half-mirror step adds the blk-parity
//     partner: 2 MFMA + 6 VALU per complex value.  Pays in the two-terms-per-phase kernels (VALU-issue
//     bound: -3 % on the backward sweep); the one-term-per-phase kernels are latency-bound and keep DPP
//     (measured +1 % with the MFMA form).
//   input side  (operator tiles, products):  column group cg(lane),  row row_in(lane)
//   output side (row sums, state, co-state): row row_out(lane), replicated over 8 adjacent lanes; lane & 7 == 0
//     writes
template <bool MFMA>
struct KhLanes;
template <>
struct KhLanes<true> {
    static __device__ __forceinline__ int cg(int lane) { return (lane >> 4) + 4 * ((lane >> 2) & 1); }
    static __device__ __forceinline__ int row_in(int lane) { return (lane & 3) + 4 * ((lane >> 3) & 1); }
    static __device__ __forceinline__ int row_out(int lane) { return (lane >> 4) + 4 * ((lane >> 3) & 1); }
    // scale * (sum of v over the 8 column groups of a row), on the row's output lanes
    static __device__ __forceinline__ double rowsum(double v, double scale) {
        const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(v, scale, 0.0, 0, 0, 0);
        return d + dpp_move<KH_DPP_HALF_MIRROR>(d);
    }
    // sum over the wave of a value that is zero except on the 8 writer lanes (lane & 7 == 0); valid on lane 0.
    // The same MFMA adds lanes {0,16,32,48} into lane 0 and {8,24,40,56} into lane 8; one row rotation joins
    // them: 1 MFMA + 3 VALU instead of the ~30 dependent instructions of a full 64-lane butterfly.
    static __device__ __forceinline__ double writers_sum(double v) {
        const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
        return d + dpp_move<KH_DPP_ROR8>(d);
    }
};
template <>
struct KhLanes<false> {
    static __device__ __forceinline__ int cg(int lane) { return lane & 7; }
    static __device__ __forceinline__ int row_in(int lane) { return lane >> 3; }
    static __device__ __forceinline__ int row_out(int lane) { return lane >> 3; }
    static __device__ __forceinline__ double rowsum(double v, double scale) { return scale * sum8(v); }
    static __device__ __forceinline__ double writers_sum(double v) { return sum64(v); }
};
typedef KhLanes<false> KhTileLanes;

template <int RPT>
struct KhTile {
    static constexpr int THREADS = 512 / RPT;
    static constexpr int WAVES = THREADS / 64;
    // rows of (wave, lane) for r in [0, RPT): whose operator elements it holds / whose sums and state it holds
    static __device__ __forceinline__ int row_in(int wave, int lane, int r) {
        return wave * (8 * RPT) + r * 8 + KhTileLanes::row_in(lane);
    }
    static __device__ __forceinline__ int row(int wave, int lane, int r) {
        return wave * (8 * RPT) + r * 8 + KhTileLanes::row_out(lane);
    }
    static __device__ __forceinline__ bool writer(int lane) { return (lane & 7) == 0; }
};

template <int RPT>
__device__ __forceinline__ void kh_tile_load_op(const cplx *op, int N, int wave, int lane, cplx (&a)[RPT][8]) {
    const int cg = KhTileLanes::cg(lane);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int row = KhTile<RPT>::row_in(wave, lane, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = cg + 8 * j;
            a[r][j] = (op != nullptr && row < N && col < N) ? op[(size_t)row * N + col] : c_make(0.0, 0.0);
        }
    }
}

// y[r] = sum_c a[r][c] x[c], summed over the row's 8 column groups (on the row's 8 output lanes)
template <int RPT>
__device__ __forceinline__ void kh_tile_matvec(const cplx (&a)[RPT][8], const cplx *x, int cg, cplx (&y)[RPT]) {
    cplx xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = x[cg + 8 * j];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        cplx acc = c_make(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < 8; ++j) c_fma(acc, a[r][j], xv[j]);
        y[r].x = KhTileLanes::rowsum(acc.x, 1.0);
        y[r].y = KhTileLanes::rowsum(acc.y, 1.0);
    }
}

// state (in registers of every lane of the row group) <- exp(f A dt) state.
// buf: LDS ping-pong vectors.  On entry buf[cur] holds the state (all rows
// written, barrier passed); on exit buf[cur] (updated) holds the NEW state, so
// consecutive intervals chain without an extra write + barrier: the last
// Taylor term is never needed in LDS, its slot receives the state instead.
// One barrier per matrix-vector product.  Returns the matvec count.
template <int RPT>
__device__ __forceinline__ int kh_tile_expm_action(const cplx (&a)[RPT][8], cplx (&state)[RPT],
                                                   cplx (*buf)[KH_TILE_N], const double *inv, int &cur, double fre,
                                                   double fim,
                                                   double dt, int nsub, int m, int wave, int lane) {
    const int cg = KhTileLanes::cg(lane);
    const bool writer = KhTile<RPT>::writer(lane);
    const double h = nsub == 1 ? dt : dt / nsub;
    for (int sub = 0; sub < nsub; ++sub) {
        const double c0 = inv[0];  // T_0 = c_0 v (1 for Taylor); ratio 1 is relative to v itself
#pragma unroll
        for (int r = 0; r < RPT; ++r) state[r] = c_make(c0 * state[r].x, c0 * state[r].y);
        for (int j = 1; j <= m; ++j) {
            const double hj = h * inv[j];  // series ratio of term j (Taylor: 1/j); LDS: no SMEM loads in the loop
            const cplx coef = c_make(fre * hj, fim * hj);
            cplx y[RPT];
            kh_tile_matvec<RPT>(a, buf[cur], cg, y);
            const bool last = (j == m);
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const cplx t = c_mul(coef, y[r]);
                state[r].x += t.x;
                state[r].y += t.y;
                const double wx = last ? state[r].x : t.x, wy = last ? state[r].y : t.y;
                if (writer) buf[cur ^ 1][KhTile<RPT>::row(wave, lane, r)] = c_make(wx, wy);
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    return nsub * m;
}

// The 1+LT operator tiles of one objective.  With more than three operators they do not all fit
// in the register file next to the generator tile (32 VGPRs each at RPT = 1): the first NL of them
// are parked in LDS instead, each lane's 8 elements at stride THREADS so a wave reads 64
// consecutive 16-byte slots (no bank conflicts), and only the lane itself ever reads its slots
// (no barrier needed after the fill).
template <int RPT, int LT, int NL>
struct KhTileOps {
    static constexpr int THREADS = 512 / RPT;
    cplx reg[1 + LT - NL][RPT][8];
    cplx *lds;  // this lane's column of the LDS-resident tiles

    __device__ __forceinline__ cplx at(int o, int r, int j) const {
        if (o < NL) return lds[(size_t)((o * RPT + r) * 8 + j) * THREADS];
        return reg[o < NL ? 0 : o - NL][r][j];
    }
    __device__ __forceinline__ void load(const cplx *const *ops_k, int N, int wave, int lane, cplx *lds_base,
                                         int tid) {
        lds = lds_base + tid;
#pragma unroll
        for (int o = 0; o <= LT; ++o) {
            if (o < NL) {
                cplx t[RPT][8];
                kh_tile_load_op<RPT>(ops_k[o], N, wave, lane, t);
#pragma unroll
                for (int r = 0; r < RPT; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) lds[(size_t)((o * RPT + r) * 8 + j) * THREADS] = t[r][j];
            } else {
                kh_tile_load_op<RPT>(ops_k[o], N, wave, lane, reg[o < NL ? 0 : o - NL]);
            }
        }
    }
    // a = op[0] + sum_l eps_l op[1+l]
    __device__ __forceinline__ void build(const double *eps, cplx (&a)[RPT][8]) const {
#pragma unroll
        for (int r = 0; r < RPT; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                cplx v = at(0, r, j);
#pragma unroll
                for (int l = 0; l < LT; ++l) {
                    const cplx hl = at(1 + l, r, j);
                    v.x = fma(eps[l], hl.x, v.x);
                    v.y = fma(eps[l], hl.y, v.y);
                }
                a[r][j] = v;
            }
    }
    // this lane's share of (op[o] x)[row]: its 8 columns only, NOT summed over the row's lanes; xv = x[cg + 8 j]
    __device__ __forceinline__ void matvec_part(int o, const cplx (&xv)[8], cplx (&y)[RPT]) const {
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            cplx acc = c_make(0.0, 0.0);
#pragma unroll
            for (int j = 0; j < 8; ++j) c_fma(acc, at(o, r, j), xv[j]);
            y[r] = acc;
        }
    }
    // y = op[o] x, reduced over the row's 8 lanes
    __device__ __forceinline__ void matvec(int o, const cplx *x, int cg, cplx (&y)[RPT]) const {
        cplx xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = x[cg + 8 * j];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            cplx acc = c_make(0.0, 0.0);
#pragma unroll
            for (int j = 0; j < 8; ++j) c_fma(acc, at(o, r, j), xv[j]);
            y[r].x = KhTileLanes::rowsum(acc.x, 1.0);
            y[r].y = KhTileLanes::rowsum(acc.y, 1.0);
        }
    }
};

// operators parked in LDS: none up to three operators; for L = 3, 4 the update sweep parks L - 2
// (it also keeps chi, the partial sums and, second order, phi_prev), the plain sweep one for L = 4
template <int RPT, int LT>
struct KhTileLds {
    // RPT = 2 (256-thread workgroups, two per CU, K > #CUs): a tile is 64 VGPRs per lane there, so with one control the
    // drift is parked as well (64 KiB per workgroup, 128 KiB per CU) -- generator + control + broadcast vector then fit
    static constexpr int UPDATE = (RPT == 1 && LT >= 3) ? LT - 2 : (RPT == 2 && LT == 1) ? 1 : 0;
    static constexpr int STORE = (RPT == 1 && LT >= 4) ? 1 : (RPT == 2 && LT == 1) ? 1 : 0;
    // the update kernel's single-GPU instantiation has registers to spare (kh_tile_forward_update: SINGLE): three controls
    // park nothing there
    static constexpr int UPDATE_SINGLE = (RPT == 1 && LT == 3) ? 0 : UPDATE;
    static constexpr size_t bytes(int nl) { return (size_t)nl * KH_TILE_N * KH_TILE_N * sizeof(cplx); }
};

extern __shared__ __attribute__((aligned(16))) cplx kh_tile_dyn_lds[];

// the series' ratios of degree m -> LDS (workgroup-uniform m; between intervals; contains barriers)
__device__ __forceinline__ void kh_tile_load_ratios(const KhSweepArgs &p, double *inv_sh, int m, int tid) {
    __syncthreads();  // (no phase is still reading the previous ratios)
    if (tid <= KH_MAX_DEGREE) inv_sh[tid] = p.ratios[(size_t)m * KH_RATIO_STRIDE + tid];
    __syncthreads();
}

// ---------------------------------------------------------------------------
// plain propagation with storage (backward sweep / iteration-0 forward sweep)
// ---------------------------------------------------------------------------
template <int RPT, int LT>
__global__ void __launch_bounds__(512 / RPT, 2)
kh_tile_sweep_store(KhSweepArgs p, const double *__restrict__ pulses, const cplx *__restrict__ state_in,
                    cplx *__restrict__ store, cplx *__restrict__ state_out, int direction) {
    __shared__ __attribute__((aligned(16))) cplx buf[2][KH_TILE_N];
    __shared__ __attribute__((aligned(16))) double inv_sh[KH_MAX_DEGREE + 2];
    __shared__ __attribute__((aligned(16))) double deg_sh[KH_MAX_DEGREE + 2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool writer = KhTile<RPT>::writer(lane);
    if (tid <= KH_MAX_DEGREE) {
        deg_sh[tid] = p.q2_theta[tid];
    }
    const int N = p.N, nt = p.nt;
    double matvecs = 0.0;
    for (int k = blockIdx.x; k < p.K; k += gridDim.x) {
        const cplx *const *ops_k = p.ops + (size_t)k * (1 + LT);
        const double *norms_k = p.op_norms + (size_t)k * (1 + LT);
        KhTileOps<RPT, LT, KhTileLds<RPT, LT>::STORE> h;
        h.load(ops_k, N, wave, lane, kh_tile_dyn_lds, tid);
        double nrm[1 + LT];
#pragma unroll
        for (int o = 0; o <= LT; ++o) nrm[o] = norms_k[o];

        cplx state[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int row = KhTile<RPT>::row(wave, lane, r);
            state[r] = row < N ? state_in[(size_t)k * N + row] : c_make(0.0, 0.0);
        }
        __syncthreads();  // previous objective's readers are done with buf
        int cur = 0;
        if (writer) {
#pragma unroll
            for (int r = 0; r < RPT; ++r) buf[0][KhTile<RPT>::row(wave, lane, r)] = state[r];
        }
        __syncthreads();
        if (store != nullptr && wave == 0 && lane < N)
            store[((size_t)k * nt + (direction > 0 ? 0 : nt - 1)) * N + lane] = buf[0][lane];

        // per-interval scalars are fetched one interval ahead (their load
        // latency would otherwise sit on the critical path of every interval)
        const int n0 = direction > 0 ? 0 : nt - 2;
        double eps_next[LT], dt_next = p.dt[n0];
#pragma unroll
        for (int l = 0; l < LT; ++l) eps_next[l] = pulses[(size_t)l * (nt - 1) + n0];
        int m_hint = 12;
        for (int step = 0; step < nt - 1; ++step) {
            const int n = direction > 0 ? step : nt - 2 - step;
            double eps[LT];
            double theta = nrm[0];
            const double dt = dt_next;
#pragma unroll
            for (int l = 0; l < LT; ++l) {
                eps[l] = eps_next[l];
                theta += fabs(eps[l]) * nrm[1 + l];
            }
            if (step + 1 < nt - 1) {
                const int nn = direction > 0 ? n + 1 : n - 1;
                dt_next = p.dt[nn];
#pragma unroll
                for (int l = 0; l < LT; ++l) eps_next[l] = pulses[(size_t)l * (nt - 1) + nn];
            }
            int nsub, m;
            kh_degree_lookup(theta * dt, deg_sh, p.theta_max, p.inv_theta_max, m_hint, &nsub, &m);
            if (m != m_hint || step == 0) kh_tile_load_ratios(p, inv_sh, m, tid);  // (workgroup-uniform, rare)
            m_hint = m;
            cplx a[RPT][8];
            h.build(eps, a);
            matvecs += kh_tile_expm_action<RPT>(a, state, buf, inv_sh, cur, p.fre, p.fim, dt, nsub, m, wave, lane);
            // buf[cur] now holds the new state: stream it to HBM, one coalesced 1 KiB store
            if (store != nullptr && wave == 0 && lane < N)
                store[((size_t)k * nt + (direction > 0 ? n + 1 : n)) * N + lane] = buf[cur][lane];
        }
        if (state_out != nullptr && wave == 0 && lane < N) state_out[(size_t)k * N + lane] = buf[cur][lane];
    }
    if (tid == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}

// ---------------------------------------------------------------------------
// forward sweep with sequential pulse update (optimize.py:444-508)
// ---------------------------------------------------------------------------
// Requires gridDim.x == K (one resident workgroup per objective): operators
// and the running state never leave the registers / LDS of their workgroup.
// Barriers per interval: 1 (partial sums) + 1 (broadcast of the reduced sums)
// + one per Taylor term.
// SO: second-order update, compiled separately.  SINGLE: launched on ONE GPU as one launch over the sweep with more than one
// workgroup (ex.world == 1, ex.G > 1, u.internal_exchange): the sums' exchange without the cross-GPU stage and without the
// forms it does not take -- an instantiation of its own because the kernel sits at the register limit
template <int RPT, int LT, bool SO, bool SINGLE = false>
__global__ void __launch_bounds__(512 / RPT, 2)
kh_tile_forward_update(KhSweepArgs p, KhUpdateArgs u, KhExchange ex) {
    if (u.n_dev != nullptr) {  // graph-replayed stepwise mode: interval index from device memory
        u.n_begin = *u.n_dev;
        u.n_end = u.n_begin + 1;
        if (u.n_begin >= p.nt - 1) return;
    }
    constexpr int WAVES = KhTile<RPT>::WAVES;
    __shared__ __attribute__((aligned(16))) cplx buf[2][KH_TILE_N];
    __shared__ __attribute__((aligned(16))) double red[2][WAVES][LT][2];  // double-buffered on interval parity
    __shared__ __attribute__((aligned(16))) double D_sh[2][LT + 1];       // the reduced sums, by interval parity
    __shared__ __attribute__((aligned(16))) double ok_sh[2][LT];          // ... and whether their gather came back
    __shared__ __attribute__((aligned(16))) double inv_sh[KH_MAX_DEGREE + 2];
    __shared__ __attribute__((aligned(16))) double deg_sh[KH_MAX_DEGREE + 2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, cg = KhTileLanes::cg(lane);
    const bool writer = KhTile<RPT>::writer(lane);
    if (tid <= KH_MAX_DEGREE) {
        deg_sh[tid] = p.q2_theta[tid];
    }
    const int N = p.N, nt = p.nt;
    const int k = blockIdx.x;
    double matvecs = 0.0;

    const cplx *const *ops_k = p.ops + (size_t)k * (1 + LT);
    const double *norms_k = p.op_norms + (size_t)k * (1 + LT);
    KhTileOps<RPT, LT, SINGLE ? KhTileLds<RPT, LT>::UPDATE_SINGLE : KhTileLds<RPT, LT>::UPDATE> h;
    h.load(ops_k, N, wave, lane, kh_tile_dyn_lds, tid);
    double nrm[1 + LT];
#pragma unroll
    for (int o = 0; o <= LT; ++o) nrm[o] = kh_uniform(norms_k[o]);
    // dH/d eps_l is the forward control operator itself (mu.py:123-134): h[1+l] serves both
    const double chi_norm = kh_uniform(u.chi_norms[k]);

    cplx state[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int row = KhTile<RPT>::row(wave, lane, r);
        state[r] = row < N ? u.phi[(size_t)k * N + row] : c_make(0.0, 0.0);
    }
    int cur = 0;
    if (writer) {
#pragma unroll
        for (int r = 0; r < RPT; ++r) buf[0][KhTile<RPT>::row(wave, lane, r)] = state[r];
    }
    __syncthreads();

    double g_a_loc[LT];
#pragma unroll
    for (int l = 0; l < LT; ++l) g_a_loc[l] = 0.0;

    // chi_k(t_n) rows for this lane, fetched one interval ahead; second order: also the rows of the
    // state propagated under the guess pulses and 0.5 sigma_n / ||chi|| (optimize.py:468-469)
    cplx chi[RPT], prev[RPT];
    double hs = 0.0;
#pragma unroll
    for (int r = 0; r < RPT; ++r) prev[r] = c_make(0.0, 0.0);
    auto load_chi = [&](int n) {
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int row = KhTile<RPT>::row(wave, lane, r);
            chi[r] = row < N ? u.chi_store[((size_t)k * nt + n) * N + row] : c_make(0.0, 0.0);
            if constexpr (SO) prev[r] = row < N ? u.fw_prev[((size_t)k * nt + n) * N + row] : c_make(0.0, 0.0);
        }
        if constexpr (SO) hs = 0.5 * u.sigma[n] / chi_norm;
    };

    // wave-level pieces of  chi_norm * Im(mu <chi(t_n) | H_l phi>)  -> red[par][wave]; phi in buf[cur]
    // The broadcast vector is read ONCE for all L controls, and no product is summed over its row: chi (and the
    // second-order bra) is replicated over the row's 8 lanes, so every lane multiplies its own 8-column share with the
    // bra and ONE 64-lane sum per control (on the matrix core) adds shares and rows together.
    auto partial_pieces = [&](int par) {
        cplx xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = buf[cur][cg + 8 * j];
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            cplx y[RPT];
            h.matvec_part(1 + l, xv, y);
            cplx ov = c_make(0.0, 0.0);
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                // second order: the bra is chi + hs (phi - phi_prev)
                cplx bra = chi[r];
                if constexpr (SO)
                    bra = c_make(fma(hs, state[r].x - prev[r].x, chi[r].x),
                                 fma(hs, state[r].y - prev[r].y, chi[r].y));
                c_fma_conj(ov, bra, y[r]);
            }
            // Im(mu <bra | H_l phi>) needs only one real combination: reduce that, not both parts
            const double v = sum64_mfma(u.mu_re * ov.y + u.mu_im * ov.x);
            if (lane == 0) red[par][wave][l][0] = v;
        }
        matvecs += LT;
    };
    // after a barrier: the workgroup's partial sums, same value in every thread
    auto partial_total = [&](int par, double (&part)[LT]) {
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            double acc = 0.0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) acc += red[par][w][l][0];
            part[l] = chi_norm * acc;
        }
    };

    const bool emit_only = (!u.internal_exchange && u.n_begin == u.n_end);
    if ((u.internal_exchange || emit_only) && u.n_begin < nt - 1) {
        load_chi(u.n_begin);
        partial_pieces(u.n_begin & 1);
    }
    __syncthreads();
    if (emit_only) {
        double part[LT];
        partial_total(u.n_begin & 1, part);
        if (tid == 0)
            for (int l = 0; l < LT; ++l) u.wg_partial[(size_t)k * LT + l] = part[l];
        return;
    }

    // per-interval scalars, fetched one interval ahead
    // (wave-uniform values are kept in scalar registers: with L = 3, 4 the per-control scalars
    // would otherwise cost ~60 VGPRs next to the operator tiles)
    double dt_next = kh_uniform(p.dt[u.n_begin]), guess_next[LT], shape_next[LT], lam[LT];
#pragma unroll
    for (int l = 0; l < LT; ++l) {
        guess_next[l] = kh_uniform(u.guess[(size_t)l * (nt - 1) + u.n_begin]);
        shape_next[l] = kh_uniform(u.shape[(size_t)l * (nt - 1) + u.n_begin]);
        lam[l] = kh_uniform(u.lambda[l]);
    }
    int m_hint = 12;

    for (int n = u.n_begin; n < u.n_end; ++n) {
        const int par = n & 1;
        if (n + 1 < nt - 1) load_chi(n + 1);  // lands while this interval is processed
        // ---- cross-objective sum (optimize.py:470) ----
        if (SINGLE || u.internal_exchange) {
            constexpr int CH = RPT == 2 ? KH_GATHER_CHUNKS_WIDE : KH_GATHER_CHUNKS;  // (256-thread workgroups run two per CU)
            if (LT > 1 && LT <= WAVES && (SINGLE || (ex.world == 1 && ex.G > 1))) {
                // several controls, one GPU: wave 0 publishes all of them, wave l gathers control l -- L polling
                // rounds side by side instead of one wave polling 8 L granules per lane (measured: the exchange cost
                // 2.4 us per interval at L = 2 and 4.1 us at L = 4 against 1.25 us with one control)
                if (wave == 0) {
                    double part[LT];
                    partial_total(par, part);
                    kh_exchange_publish(ex, n, k, LT, lane, part);
                }
                if (wave < LT) {
                    double Dl = 0.0;
                    const bool ok = kh_gather_one<CH>(ex, par, LT, wave, (unsigned)(n + 1), lane, Dl);
                    if (lane == 0) {
                        D_sh[par][wave] = Dl;
                        ok_sh[par][wave] = ok ? 1.0 : 0.0;
                    }
                }
            } else if (wave == 0) {
                double part[LT];
                partial_total(par, part);
                double D[LT];
                const bool ok = kh_exchange<LT, CH, !SINGLE>(ex, n, k, LT, lane, part, D);
                if (lane == 0) {
#pragma unroll
                    for (int l = 0; l < LT; ++l) {
                        D_sh[par][l] = D[l];
                        ok_sh[par][l] = ok ? 1.0 : 0.0;
                    }
                }
            }
        } else if (tid == 0) {
            for (int l = 0; l < LT; ++l) {
                D_sh[par][l] = u.D_in[l];
                ok_sh[par][l] = 1.0;
            }
        }
        // scalars of this interval (prefetched) and prefetch of the next
        const double dt = dt_next;
        double guess[LT], shape[LT];
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            guess[l] = guess_next[l];
            shape[l] = shape_next[l];
        }
        double dt_ld = 0.0, guess_ld[LT], shape_ld[LT];  // in flight across the barrier
        if (n + 1 < nt - 1) {
            dt_ld = p.dt[n + 1];
#pragma unroll
            for (int l = 0; l < LT; ++l) {
                guess_ld[l] = u.guess[(size_t)l * (nt - 1) + n + 1];
                shape_ld[l] = u.shape[(size_t)l * (nt - 1) + n + 1];
            }
        }
        __syncthreads();
        {
            bool all_ok = true;
#pragma unroll
            for (int l = 0; l < LT; ++l) all_ok = all_ok && ok_sh[par][l] != 0.0;
            if (!all_ok) return;
        }
        // ---- pulse update (optimize.py:471-477) ----
        double eps[LT];
        double theta = nrm[0];
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            const double d1 = D_sh[par][l];
            const double step = shape[l] / lam[l];
            eps[l] = kh_uniform(guess[l] + step * d1);
            g_a_loc[l] = kh_uniform(g_a_loc[l] + step * (d1 * d1) * dt);
            theta += fabs(eps[l]) * nrm[1 + l];
        }
        if (k == 0 && tid == 0) {
#pragma unroll
            for (int l = 0; l < LT; ++l) u.opt[(size_t)l * (nt - 1) + n] = eps[l];
        }
        // ---- propagate over interval n with the updated pulse (optimize.py:479-491) ----
        int nsub, m;
        kh_degree_lookup(theta * dt, deg_sh, p.theta_max, p.inv_theta_max, m_hint, &nsub, &m);
        if (m != m_hint || n == u.n_begin) kh_tile_load_ratios(p, inv_sh, m, tid);  // (workgroup-uniform, rare)
        m_hint = m;
        cplx a[RPT][8];
        h.build(eps, a);
        if constexpr (SO) {
            if (wave == 0 && lane < N) u.fw_store[((size_t)k * nt + n) * N + lane] = buf[cur][lane];
        }
        matvecs += kh_tile_expm_action<RPT>(a, state, buf, inv_sh, cur, p.fre, p.fim, dt, nsub, m, wave, lane);
        if (n + 1 < nt - 1) {
            dt_next = kh_uniform(dt_ld);
#pragma unroll
            for (int l = 0; l < LT; ++l) {
                guess_next[l] = kh_uniform(guess_ld[l]);
                shape_next[l] = kh_uniform(shape_ld[l]);
            }
        }
        // ---- partial sums of the next interval (state is in buf[cur], barrier passed) ----
        if (n + 1 < nt - 1) {
            partial_pieces((n + 1) & 1);
            __syncthreads();  // every wave's piece is in red[] before wave 0 totals it
        }
    }
    // running state back to the engine workspace (final states / next launch)
    if (wave == 0 && lane < N) {
        u.phi[(size_t)k * N + lane] = buf[cur][lane];
        if constexpr (SO) u.fw_store[((size_t)k * nt + u.n_end) * N + lane] = buf[cur][lane];
    }
    if (!u.internal_exchange && u.n_end < nt - 1) {
        double part[LT];
        partial_total(u.n_end & 1, part);
        if (tid == 0)
            for (int l = 0; l < LT; ++l) u.wg_partial[(size_t)k * LT + l] = part[l];
    }
    if (k == 0 && tid == 0)
        for (int l = 0; l < LT; ++l) u.g_a[l] = (u.internal_exchange ? 0.0 : u.g_a[l]) + g_a_loc[l];
    if (tid == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}
