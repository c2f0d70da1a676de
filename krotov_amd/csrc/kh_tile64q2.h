// Two-terms-per-phase register-tile kernels: N <= 64, exactly one control (L = 1).
//
// The Taylor series of exp(c A) v is a chain of dependent matrix-vector
// products; on one CU each product costs a fixed LDS-write -> barrier ->
// LDS-read round trip (~250 cycles) and a cross-lane reduction besides its
// 256 cycles of fp64 FMAs.  With a single control the generator is
// A(eps) = H0 + eps H1, so its square is a quadratic polynomial with three
// FIXED matrices,  A^2 = P0 + eps P1 + eps^2 P2,  P0 = H0 H0,
// P1 = H0 H1 + H1 H0,  P2 = H1 H1  (staged once by kh_engine_create).  Keeping
// both A and B = A^2 in registers, every phase produces TWO Taylor terms from
// one broadcast of the input vector:
//     t_{2p+1} = c/(2p+1) A t_{2p},   t_{2p+2} = c^2/((2p+1)(2p+2)) B t_{2p},
// halving the number of barriers/LDS round trips and LDS vector reads per FMA.
// Flop count is unchanged (one matvec per term).
//
// Register budget (512 threads, 2 waves per SIMD, 256 VGPRs): H1, P1, P2, A, B
// tiles (5 x 32 dwords) + the broadcast vector (32).  H0 and P0 tiles live in
// LDS (2 x 64 KiB, lane-linear: conflict-free ds_read_b128) and are re-read
// once per interval when A and B are rebuilt.
#pragma once

#include "kh_common.h"
#include "kh_generic.h"
#include "kh_tile64.h"

#define KH_Q2_THREADS 512
#ifdef KH_Q2_DPP_REDUCE  // all-VALU row sums, for A/B timing (scripts/ab_q2_reduce.sh)
typedef KhLanes<false> KhQ2Lanes;
#else
typedef KhLanes<true> KhQ2Lanes;  // row sums on the matrix core (kh_tile64.h, "Lane roles")
#endif
#define KH_Q2_TILE_ELEMS (8 * KH_Q2_THREADS)  // complex elements of one 64x64 operator, lane-linear

struct KhQ2Lds {
    cplx *h0;    // [8][512]
    cplx *p0;    // [8][512]
    cplx (*buf)[KH_TILE_N];  // [2][64]
    cplx *chib;  // [64] chi(t_{n+1}) for the adjoint-side partial sums
    cplx *sbuf;  // [64] the vector s of kh_q2_expm_action
    double *red; // [2][8 waves][2]
    double *D;   // [2][2]
    double2 *inv2;  // [KH_Q2_ROWS] the series' rows {r1_p, r2_p} of the current degree (Taylor: {1/(2p+1),
                    // 1/((2p+1)(2p+2))}); LDS: no SMEM loads in the phase loop.  [KH_Q2_ROWS]: c_0 (in .x)
    double *deg;    // [KH_MAX_DEGREE+1] copy of the degree-threshold table
};

__host__ __device__ inline size_t kh_q2_lds_bytes() {
    return (size_t)2 * KH_Q2_TILE_ELEMS * sizeof(cplx) + 4 * KH_TILE_N * sizeof(cplx) + (2 * 8 * 2 + 4) * sizeof(double) +
           (KH_MAX_DEGREE / 2 + 1) * sizeof(double2) + (KH_MAX_DEGREE + 2) * sizeof(double);
}

__device__ __forceinline__ KhQ2Lds kh_q2_carve(char *smem) {
    KhQ2Lds s;
    s.h0 = (cplx *)smem;
    s.p0 = s.h0 + KH_Q2_TILE_ELEMS;
    s.buf = (cplx(*)[KH_TILE_N])(s.p0 + KH_Q2_TILE_ELEMS);
    s.chib = (cplx *)(s.buf + 2);
    s.sbuf = s.chib + KH_TILE_N;
    s.red = (double *)(s.sbuf + KH_TILE_N);
    s.D = s.red + 2 * 8 * 2;
    s.inv2 = (double2 *)(s.D + 4);
    s.deg = (double *)(s.inv2 + KH_MAX_DEGREE / 2 + 1);
    return s;
}

// this lane's 8 elements of an operator (row wave*8 + row_in, columns cg + 8 j)
__device__ __forceinline__ void kh_q2_load_tile(const cplx *op, int N, int wave, int lane, cplx (&t)[8]) {
    const int row = wave * 8 + KhQ2Lanes::row_in(lane), cg = KhQ2Lanes::cg(lane);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int col = cg + 8 * j;
        t[j] = (op != nullptr && row < N && col < N) ? op[(size_t)row * N + col] : c_make(0.0, 0.0);
    }
}

__device__ __forceinline__ void kh_q2_stage_tile(const cplx *op, int N, int wave, int lane, int tid, cplx *dst) {
    cplx t[8];
    kh_q2_load_tile(op, N, wave, lane, t);
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[j * KH_Q2_THREADS + tid] = t[j];
}

// the series' rows of degree m -> LDS (workgroup-uniform m; called between intervals, contains a barrier)
__device__ __forceinline__ void kh_q2_load_rows(const KhSweepArgs &p, const KhQ2Lds &s, int m, int tid) {
    __syncthreads();  // (no phase is still reading the previous rows)
    if (tid < KH_Q2_ROWS) {
        const double *r = p.q2_rows + ((size_t)m * KH_Q2_ROWS + tid) * 2;
        s.inv2[tid] = make_double2(r[0], r[1]);
    }
    if (tid == KH_Q2_ROWS) s.inv2[KH_Q2_ROWS] = make_double2(p.q2_c0[m], 0.0);
    __syncthreads();
}

// A = H0 + eps H1,  B = P0 + eps P1 + eps^2 P2  (H0, P0 from LDS)
__device__ __forceinline__ void kh_q2_build(const KhQ2Lds &s, int tid, double eps, const cplx (&h1)[8],
                                            const cplx (&p1)[8], const cplx (&p2)[8], cplx (&a)[8], cplx (&b)[8]) {
    const double eps2 = eps * eps;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const cplx h0 = s.h0[j * KH_Q2_THREADS + tid];
        const cplx q0 = s.p0[j * KH_Q2_THREADS + tid];
        a[j].x = fma(eps, h1[j].x, h0.x);
        a[j].y = fma(eps, h1[j].y, h0.y);
        b[j].x = fma(eps2, p2[j].x, fma(eps, p1[j].x, q0.x));
        b[j].y = fma(eps2, p2[j].y, fma(eps, p1[j].y, q0.y));
    }
}

#define KH_Q2_REFRESH 64
// The plain sweeps advance resident tiles too, entirely in registers (H1, P1, P2 are there already):
//   A += (eps - eps') H1,   B += (eps - eps') P1 + (eps^2 - eps'^2) P2
// and restart from H0, P0 in LDS every KH_Q2_REFRESH intervals: no LDS traffic per interval instead of the 128 KiB
// a workgroup read to rebuild its two tiles (1 024 cycles of the LDS pipe in front of the first phase).
__device__ __forceinline__ void kh_q2_restart_lds(const KhQ2Lds &s, int tid, cplx (&a)[8], cplx (&b)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = s.h0[j * KH_Q2_THREADS + tid];
        b[j] = s.p0[j * KH_Q2_THREADS + tid];
    }
}
__device__ __forceinline__ void kh_q2_advance_reg(double eps, double eps_prev, const cplx (&h1)[8], const cplx (&p1)[8],
                                                  const cplx (&p2)[8], cplx (&a)[8], cplx (&b)[8]) {
    const double e1 = eps - eps_prev, e2 = e1 * (eps + eps_prev);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j].x = fma(e1, h1[j].x, a[j].x);
        a[j].y = fma(e1, h1[j].y, a[j].y);
        b[j].x = fma(e2, p2[j].x, fma(e1, p1[j].x, b[j].x));
        b[j].y = fma(e2, p2[j].y, fma(e1, p1[j].y, b[j].y));
    }
}

// The update kernels keep A and B resident and advance them from interval to interval instead:
//   A += (eps - eps') H1,   B += (eps - eps') P1 + (eps^2 - eps'^2) P2      (P1, P2 from LDS)
// Only H1, A and B stay in registers (96 VGPRs instead of the 160 of H1, P1, P2, A, B): no spills.  Every
// KH_Q2_REFRESH intervals A and B restart from H0 and P0 in global memory (eps' = 0), so rounding cannot drift
// (64 updates: a few ulp of the tiles).
__device__ __forceinline__ void kh_q2_advance(const KhQ2Lds &s, int tid, double eps, double eps_prev, const cplx (&h1)[8],
                                              cplx (&a)[8], cplx (&b)[8]) {
    const double e1 = eps - eps_prev, e2 = e1 * (eps + eps_prev);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const cplx q1 = s.h0[j * KH_Q2_THREADS + tid];  // (the update kernels stage P1, P2 where the plain sweep has H0, P0)
        const cplx q2 = s.p0[j * KH_Q2_THREADS + tid];
        a[j].x = fma(e1, h1[j].x, a[j].x);
        a[j].y = fma(e1, h1[j].y, a[j].y);
        b[j].x = fma(e2, q2.x, fma(e1, q1.x, b[j].x));
        b[j].y = fma(e2, q2.y, fma(e1, q1.y, b[j].y));
    }
}

// state <- exp(f A dt) state with two Taylor terms per phase.  On entry
// buf[cur] holds the state; on exit buf[cur] holds the new state.  f*f is real
// (-1 in Hilbert space, +1 for Liouvillians): c2 = f^2 h^2 / (j1 j2).
//
// The even terms are a chain of products with B = A^2:  t_{2p+2} = c2_p B t_{2p}.
// The odd terms  t_{2p+1} = f h/(2p+1) A t_{2p}  only enter the state sum, and A is
// linear:  sum_p t_{2p+1} = f A s  with  s = sum_p h/(2p+1) t_{2p}.  So each lane
// accumulates its row of s while the even terms go by (two FMAs per phase), the
// phase that produces the last input t_{2(P-1)} also writes s to LDS, and the
// LAST phase does the one A product (on s) next to its B product: P + 1
// matrix-vector products per step instead of 2 P, on the same critical path of
// P phases.  (With m = 14: 8 products instead of 14.)
// `epilogue()` runs once, when the new state is complete in `state` and before the last barrier: work that
// depends on the new state and must be visible after that barrier rides on it instead of a barrier of its own.
// Returns the number of matrix-vector products issued.
template <class Epilogue>
__device__ __forceinline__ int kh_q2_expm_action(const cplx (&a)[8], const cplx (&b)[8], cplx &state,
                                                 cplx (*buf)[KH_TILE_N], cplx *sbuf, const double2 *inv2, int &cur,
                                                 cplx *store_in, int N, double fre, double fim,
                                                 double dt, int nsub, int m, int wave, int lane,
                                                 Epilogue epilogue) {
    const int cg = KhQ2Lanes::cg(lane), row = wave * 8 + KhQ2Lanes::row_out(lane);
    const bool writer = (lane & 7) == 0;
    const double h = nsub == 1 ? dt : dt / nsub;
    const double f2h2 = (fre * fre - fim * fim) * h * h;  // f is purely real or purely imaginary
    const int phases = (m + 1) >> 1;
    if (store_in != nullptr && wave == 0 && lane < N) {
        // the interval's incoming state goes to HBM from here (one LDS read + one fire-and-forget
        // coalesced store, outside the phase loop so the loop carries no exec-mask juggling for it)
        store_in[lane] = buf[cur][lane];
    }
    for (int sub = 0; sub < nsub; ++sub) {
        // this row of s = sum_p r1_p h T_2p (T_0 = c_0 v, r1_0 relative to the incoming state v; Taylor: 1/(2p+1))
        const double hr = h * inv2[0].x, c0 = inv2[KH_Q2_ROWS].x;
        cplx sacc = c_make(hr * state.x, hr * state.y);
        state = c_make(c0 * state.x, c0 * state.y);
        if (phases == 1) {  // (degree <= 2) s = h t_0 is final already: one extra barrier in this rare case
            if (writer) sbuf[row] = sacc;
            __syncthreads();
        }
        for (int ph = 0; ph < phases; ++ph) {
            const double c2 = f2h2 * inv2[ph].y;  // inv2[p] = {r1_p, r2_p}
            cplx xv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = buf[cur][cg + 8 * j];
            const bool last = (ph + 1 == phases);
            cplx yb = c_make(0.0, 0.0);
#pragma unroll
            for (int j = 0; j < 8; ++j) c_fma(yb, b[j], xv[j]);
            const double t2x = KhQ2Lanes::rowsum(yb.x, c2), t2y = KhQ2Lanes::rowsum(yb.y, c2);
            state.x += t2x;
            state.y += t2y;
            if (!last) {
                const double hn = h * inv2[ph + 1].x;
                sacc.x = fma(hn, t2x, sacc.x);
                sacc.y = fma(hn, t2y, sacc.y);
                if (writer) {
                    buf[cur ^ 1][row] = c_make(t2x, t2y);
                    if (ph + 2 == phases) sbuf[row] = sacc;  // s is complete: next phase multiplies it by A
                }
            } else {
                // (the B product is finished before s is fetched: both vectors at once do not fit next to the tiles)
                __builtin_amdgcn_sched_barrier(0);
                cplx sv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) sv[j] = sbuf[cg + 8 * j];
                cplx ya = c_make(0.0, 0.0);
#pragma unroll
                for (int j = 0; j < 8; ++j) c_fma(ya, a[j], sv[j]);
                const cplx odd = c_mul(c_make(fre, fim), c_make(KhQ2Lanes::rowsum(ya.x, 1.0), KhQ2Lanes::rowsum(ya.y, 1.0)));
                state.x += odd.x;
                state.y += odd.y;
                if (writer) buf[cur ^ 1][row] = c_make(state.x, state.y);
                if (sub + 1 == nsub) epilogue();
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    return nsub * (phases + 1);
}

// ---------------------------------------------------------------------------
// plain propagation with storage (backward sweep / iteration-0 forward sweep)
// ---------------------------------------------------------------------------
// sq: [K*3] pointers to P0, P1, P2 of this direction's operators
__global__ void __launch_bounds__(KH_Q2_THREADS)
kh_q2_sweep_store(KhSweepArgs p, const cplx *const *__restrict__ sq, const double *__restrict__ pulses,
                  const cplx *__restrict__ state_in, cplx *__restrict__ store, cplx *__restrict__ state_out,
                  int direction)
#if KH_DEFINES(KH_TU_Q2)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KhQ2Lds s = kh_q2_carve(smem);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid <= KH_MAX_DEGREE) s.deg[tid] = p.q2_theta[tid];
    const int row = wave * 8 + KhQ2Lanes::row_out(lane);  // the row whose sums/state this lane holds
    const bool writer = (lane & 7) == 0;
    const int N = p.N, nt = p.nt;
    double matvecs = 0.0;
    for (int k = blockIdx.x; k < p.K; k += gridDim.x) {
        const cplx *const *ops_k = p.ops + (size_t)k * 2;
        const cplx *const *sq_k = sq + (size_t)k * 3;
        __syncthreads();  // previous objective's readers are done with LDS
        kh_q2_stage_tile(ops_k[0], N, wave, lane, tid, s.h0);
        kh_q2_stage_tile(sq_k[0], N, wave, lane, tid, s.p0);
        cplx h1[8], p1[8], p2[8];
        kh_q2_load_tile(ops_k[1], N, wave, lane, h1);
        kh_q2_load_tile(sq_k[1], N, wave, lane, p1);
        kh_q2_load_tile(sq_k[2], N, wave, lane, p2);
        const double nrm0 = p.op_norms[(size_t)k * 2], nrm1 = p.op_norms[(size_t)k * 2 + 1];

        cplx state = row < N ? state_in[(size_t)k * N + row] : c_make(0.0, 0.0);
        int cur = 0;
        if (writer) s.buf[0][row] = state;
        __syncthreads();
        // the state entering interval `step` is stored from inside its first phase
        // (index n for the forward direction, n+1 for the backward one); the last
        // state is stored after the loop.
        const int n0 = direction > 0 ? 0 : nt - 2;
        double eps_next = pulses[n0], dt_next = p.dt[n0];
        KhDegreeCache dc = {12, 1.0, 0.0};
        int m_rows = -1;
#ifdef KH_TIMING
        long long t_build = 0, t_phases = 0;
#endif
        cplx a[8], b[8];
        double eps_prev = 0.0;
        for (int step0 = 0; step0 < nt - 1; step0 += KH_Q2_REFRESH) {
        // (restart outside the interval loop: its body keeps ONE definition of the tiles, see kh_q2_forward_update)
        kh_q2_restart_lds(s, tid, a, b);
        eps_prev = 0.0;
        const int step_stop = step0 + KH_Q2_REFRESH < nt - 1 ? step0 + KH_Q2_REFRESH : nt - 1;
        for (int step = step0; step < step_stop; ++step) {
#ifdef KH_TIMING
            const long long tq0 = clock64();
#endif
            const int n = direction > 0 ? step : nt - 2 - step;
            const double eps = eps_next, dt = dt_next;
            if (step + 1 < nt - 1) {
                const int nn = direction > 0 ? n + 1 : n - 1;
                dt_next = p.dt[nn];
                eps_next = pulses[nn];
            }
            int nsub, m;
            kh_degree_cached((nrm0 + fabs(eps) * nrm1) * dt, s.deg, p.theta_max, p.inv_theta_max, dc, &nsub, &m);
            if (m != m_rows) {  // (rare along a smooth pulse)
                kh_q2_load_rows(p, s, m, tid);
                m_rows = m;
            }
            kh_q2_advance_reg(eps, eps_prev, h1, p1, p2, a, b);
            eps_prev = eps;
            cplx *store_in =
                store == nullptr ? nullptr : store + ((size_t)k * nt + (direction > 0 ? n : n + 1)) * N;
#ifdef KH_TIMING
            const long long tq1 = clock64();
            t_build += tq1 - tq0;
#endif
            matvecs += kh_q2_expm_action(a, b, state, s.buf, s.sbuf, s.inv2, cur, store_in, N, p.fre, p.fim, dt, nsub, m,
                                         wave, lane, [] {});
#ifdef KH_TIMING
            t_phases += clock64() - tq1;
#endif
        }
        }
#ifdef KH_TIMING
        if (tid == 0 && k == 0 && p.stats != nullptr) {
            p.stats[1] = (double)t_build;   // scalars, degree, tile rebuild issue
            p.stats[2] = (double)t_phases;  // (the rebuild's LDS latency lands here)
        }
#endif
        if (store != nullptr && wave == 0 && lane < N)
            store[((size_t)k * nt + (direction > 0 ? nt - 1 : 0)) * N + lane] = s.buf[cur][lane];
        if (state_out != nullptr && wave == 0 && lane < N) state_out[(size_t)k * N + lane] = s.buf[cur][lane];
    }
    if (tid == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif

// ---------------------------------------------------------------------------
// forward sweep with sequential pulse update (optimize.py:444-508), grid == K
// ---------------------------------------------------------------------------
// SO: second-order update (compiled separately so first order keeps its register budget).
// ADJ (first order, in-kernel exchange, control operator equal to +/- its adjoint): the partial sum of interval
// n+1 is taken as <w|phi(t_{n+1})> with w = (sign H1) chi(t_{n+1}).  w does not depend on phi, so its
// matrix-vector product runs in the shadow of the exchange of interval n (the other seven waves idle there, and
// wave 0 has ~1 us between its store and the first useful poll), and what is left after the last Taylor phase
// -- one complex multiply per row, the wave sum, one LDS write -- rides on that phase's barrier: the separate
// "partial sums" step (8 LDS reads, 32 FMAs, row sums, a barrier: 0.46 us of 6 per interval) disappears.
// SINGLE: launched on one GPU (ex.world == 1): the sums' exchange without the cross-GPU stage -- its code, inlined by a
// run-time branch otherwise, costs this kernel the handful of registers that spill (256 VGPRs, 6 spilled -> 253, none)
template <bool SO, bool ADJ, bool SINGLE = false>
__global__ void __launch_bounds__(KH_Q2_THREADS)
kh_q2_forward_update(KhSweepArgs p, const cplx *const *__restrict__ sq, KhUpdateArgs u, KhExchange ex) {
    static_assert(!(SO && ADJ), "the second-order bra depends on the new state");
#ifdef KH_Q2_NO_PREFETCH  // (A/B build)
    constexpr bool PREFETCH = false;
#else
    constexpr bool PREFETCH = !SO;
#endif
    if (u.n_dev != nullptr) {  // graph-replayed stepwise mode: interval index from device memory
        u.n_begin = *u.n_dev;
        u.n_end = u.n_begin + 1;
        if (u.n_begin >= p.nt - 1) return;
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KhQ2Lds s = kh_q2_carve(smem);
    double(*red)[8][2] = (double(*)[8][2])s.red;  // [parity][wave][re, im]
    double(*D_sh)[2] = (double(*)[2])s.D;         // [parity][value, ok]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, cg = KhQ2Lanes::cg(lane);
    if (tid <= KH_MAX_DEGREE) s.deg[tid] = p.q2_theta[tid];
    const int row = wave * 8 + KhQ2Lanes::row_out(lane);  // the row whose sums/state/co-state this lane holds
    // one wave of each SIMD's pair (wave 0, which runs the exchange, among them) issues first when both are
    // ready: the pair's latency gaps interleave instead of coinciding (measured -1.7 % on the update sweep)
    if (wave < 4) __builtin_amdgcn_s_setprio(1);
    const bool writer = (lane & 7) == 0;
    const int N = p.N, nt = p.nt;
    const int k = blockIdx.x;
    double matvecs = 0.0;

    const cplx *const *ops_k = p.ops + (size_t)k * 2;
    const cplx *const *sq_k = sq + (size_t)k * 3;
    kh_q2_stage_tile(sq_k[1], N, wave, lane, tid, s.h0);  // P1, P2 (kh_q2_advance)
    kh_q2_stage_tile(sq_k[2], N, wave, lane, tid, s.p0);
    cplx h1[8], a[8], b[8];
    kh_q2_load_tile(ops_k[1], N, wave, lane, h1);  // also dH/d eps (mu.py:123-134)
    double eps_prev = 0.0;
    // (wave-uniform scalars live in SGPRs: the VGPR file is full of operator tiles)
    const double nrm0 = kh_uniform(p.op_norms[(size_t)k * 2]), nrm1 = kh_uniform(p.op_norms[(size_t)k * 2 + 1]);
    const double chi_norm = kh_uniform(u.chi_norms[k]);

    cplx state = row < N ? u.phi[(size_t)k * N + row] : c_make(0.0, 0.0);
    int cur = 0;
    if (writer) s.buf[0][row] = state;
    __syncthreads();

    double g_a_loc = 0.0;
    // chi_k(t_n) row of this lane, fetched one interval ahead; second order: also the row of the
    // state propagated under the guess pulses and 0.5 sigma_n / ||chi|| (optimize.py:468-469)
    cplx chi = c_make(0.0, 0.0), prev = c_make(0.0, 0.0);
    double hs = 0.0;
    auto load_chi = [&](int n) {
        chi = row < N ? u.chi_store[((size_t)k * nt + n) * N + row] : c_make(0.0, 0.0);
        if constexpr (SO) {
            prev = row < N ? u.fw_prev[((size_t)k * nt + n) * N + row] : c_make(0.0, 0.0);
            hs = 0.5 * u.sigma[n] / chi_norm;
        }
    };

    // wave-level pieces of <chi(t_n) | H1 phi> -> red[par][wave]; phi in buf[cur]
    auto partial_pieces = [&](int par) {
        cplx xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = s.buf[cur][cg + 8 * j];
        cplx y = c_make(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < 8; ++j) c_fma(y, h1[j], xv[j]);
        y.x = KhQ2Lanes::rowsum(y.x, 1.0);
        y.y = KhQ2Lanes::rowsum(y.y, 1.0);
        cplx ov = c_make(0.0, 0.0);
        // <chi + hs (phi - phi_prev) | H1 phi>: the second-order bra folded into the co-state
        cplx bra = chi;
        if constexpr (SO) bra = c_make(fma(hs, state.x - prev.x, chi.x), fma(hs, state.y - prev.y, chi.y));
        if (writer) c_fma_conj(ov, bra, y);
        // Im(mu <chi|H1 phi>) needs only one real combination: reduce that, not both parts
        const double v = KhQ2Lanes::writers_sum(u.mu_re * ov.y + u.mu_im * ov.x);
        if (lane == 0) red[par][wave][0] = v;
        matvecs += 1.0;
    };
    auto partial_total = [&](int par) {
        double acc = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) acc += red[par][w][0];
        return chi_norm * acc;
    };

    // ADJ: w = sign * H1 chi(t_{n+1}) (chi broadcast from LDS) on the row's output lanes
    cplx w = c_make(0.0, 0.0);
    auto adjoint_side = [&]() {
        cplx xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = s.chib[cg + 8 * j];
        cplx y = c_make(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < 8; ++j) c_fma(y, h1[j], xv[j]);
        w.x = KhQ2Lanes::rowsum(y.x, u.adj_sign);
        w.y = KhQ2Lanes::rowsum(y.y, u.adj_sign);
        matvecs += 1.0;
    };

    // (this kernel is only launched as ONE launch over the sweep, sums exchanged in-kernel: the per-interval form of the
    // sharded RCCL path runs kh_tile_forward_update -- krotov_hip.hip:launch_update; the branches for it are gone)
    if (u.n_begin < nt - 1) {
        load_chi(u.n_begin);
        partial_pieces(u.n_begin & 1);
    }
    if constexpr (ADJ) {
        if (u.n_begin + 1 < nt - 1) {
            load_chi(u.n_begin + 1);
            if (writer) s.chib[row] = chi;
        }
    }
    __syncthreads();

    double dt_next = kh_uniform(p.dt[u.n_begin]), guess_next = kh_uniform(u.guess[u.n_begin]),
           shape_next = kh_uniform(u.shape[u.n_begin]);
    const double lam = kh_uniform(u.lambda[0]);
    int m_rows = -1;
    // S/lambda of the coming interval is formed one interval ahead: the fp64 division (~25 dependent
    // instructions) otherwise sits between the exchange and the rebuild of the tiles
    double stepw_next = kh_uniform(shape_next / lam);
    KhDegreeCache dc = {12, 1.0, 0.0};

#ifdef KH_TIMING
    long long t_ex = 0, t_prop = 0, t_part = 0, t_coll = 0;
    const long long t_all0 = clock64();
#endif
    for (int nr = u.n_begin, n_stop; nr < u.n_end; nr = n_stop) {
    // restart A = H0, B = P0 from global memory (outside the interval loop: the loop body keeps one definition of
    // the tiles, which the register allocator needs to keep them in place).  Restart points are ABSOLUTE interval
    // indices (multiples of KH_Q2_REFRESH, plus the launch's first interval), so a sweep cut into several launches
    // restarts where the single launch does.  Rounding: an advanced tile carries an absolute error of a few
    // ulp(max |eps|^2 |P2|) over the window -- relative to the LARGEST pulse value of the window, not the current one.
    kh_q2_load_tile(ops_k[0], N, wave, lane, a);
    kh_q2_load_tile(sq_k[0], N, wave, lane, b);
    eps_prev = 0.0;
    n_stop = (nr / KH_Q2_REFRESH + 1) * KH_Q2_REFRESH;
    n_stop = n_stop < u.n_end ? n_stop : u.n_end;
    for (int n = nr; n < n_stop; ++n) {
        const int par = n & 1;
        if constexpr (!ADJ) {
            if (n + 1 < nt - 1) load_chi(n + 1);
        }
#ifdef KH_TIMING
        const long long tq0 = clock64();
#endif
        // ---- cross-objective sum (optimize.py:470) ----
        cplx q1[8], q2[8];
        {
            double part[1] = {0.0};
            if (wave == 0) {
                part[0] = partial_total(par);
#ifndef KH_Q2_X_NOEXCH
                kh_exchange_publish(ex, n, k, 1, lane, part);
#endif
            }
            // The P1 / P2 elements of the coming tile advance leave LDS now, while the sums cross the GPU (the
            // registers are free here: no product is in flight), instead of behind the exchange in front of the
            // first phase (16 ds_read_b128 per lane = 128 KiB per workgroup and interval: ~1000 cycles of LDS pipe).
            // Not in the second-order instantiations: they carry phi_prev, sigma and the forward-side partial sums
            // besides, and the 64 landing registers pushed 19-26 values into scratch -- one of them reloaded in every
            // phase (second-order sweep 6.8 us per interval against 4.8 first order; without the prefetch: see
            // docs/HISTORY.md R4.8)
            if constexpr (PREFETCH) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    q1[j] = s.h0[j * KH_Q2_THREADS + tid];
                    q2[j] = s.p0[j * KH_Q2_THREADS + tid];
                }
            }
            if constexpr (ADJ) {
                if (n + 1 < nt - 1) adjoint_side();  // chib holds chi(t_{n+1})
            }
            if (wave == 0) {
                double D[1];
#ifdef KH_Q2_X_NOEXCH  // (timing experiment: wrong results) the workgroup's own sum instead of everybody's
                const bool ok = true;
                D[0] = part[0];
#else
#ifdef KH_TIMING
                const long long tc0 = clock64();
#endif
                const bool ok = kh_exchange_collect<1, KH_GATHER_CHUNKS, !SINGLE>(ex, n, k, 1, lane, part, D);
#ifdef KH_TIMING
                t_coll += clock64() - tc0;
#endif
#endif
                if (lane == 0) {
                    D_sh[par][0] = D[0];
                    D_sh[par][1] = ok ? 1.0 : 0.0;
                }
            }
        }
        // (issued here, not before the exchange: measured 21.3 vs 22.1 ms per sweep)
        const double dt = dt_next, guess = guess_next, stepw = stepw_next;
        double dt_ld = 0.0, guess_ld = 0.0, shape_ld = 0.0;  // in flight across the barrier
        if (n + 1 < nt - 1) {
            dt_ld = p.dt[n + 1];
            guess_ld = u.guess[n + 1];
            shape_ld = u.shape[n + 1];
        }
        __syncthreads();
#ifdef KH_TIMING
        const long long tq1 = clock64();
        t_ex += tq1 - tq0;
#endif
        if (D_sh[par][1] == 0.0) return;
        // ---- pulse update (optimize.py:471-477) ----
        const double d1 = D_sh[par][0];
        const double eps = kh_uniform(guess + stepw * d1);
        g_a_loc = kh_uniform(g_a_loc + stepw * (d1 * d1) * dt);
        dt_next = kh_uniform(dt_ld);
        guess_next = kh_uniform(guess_ld);
        shape_next = kh_uniform(shape_ld);
        if (k == 0 && tid == 0) u.opt[n] = eps;
        // ---- propagate over interval n with the updated pulse (optimize.py:479-491) ----
        int nsub, m;
        kh_degree_cached((nrm0 + fabs(eps) * nrm1) * dt, s.deg, p.theta_max, p.inv_theta_max, dc, &nsub, &m);
        if (m != m_rows) {  // (rare along a smooth pulse)
            kh_q2_load_rows(p, s, m, tid);
            m_rows = m;
        }
        if constexpr (PREFETCH)
            kh_q2_advance_reg(eps, eps_prev, h1, q1, q2, a, b);
        else
            kh_q2_advance(s, tid, eps, eps_prev, h1, a, b);
        eps_prev = eps;
        stepw_next = kh_uniform(shape_next / lam);  // (under the LDS latency of the tile reads)
        cplx *fw_out = nullptr;
        if constexpr (SO) fw_out = u.fw_store + ((size_t)k * nt + n) * N;
        if constexpr (ADJ) {
            if (n + 2 < nt - 1) load_chi(n + 2);  // lands during the phases; goes to LDS in the epilogue
            auto epilogue = [&] {
                if (n + 1 < nt - 1) {
                    cplx ov = c_make(0.0, 0.0);
                    if (writer) c_fma_conj(ov, w, state);
                    const double v = KhQ2Lanes::writers_sum(u.mu_re * ov.y + u.mu_im * ov.x);
                    if (lane == 0) red[(n + 1) & 1][wave][0] = v;
                    if (n + 2 < nt - 1 && writer) s.chib[row] = chi;
                }
            };
#ifdef KH_Q2_X_NOPHASES  // (timing experiment: wrong results) the exchange alone: no propagation
            epilogue();
            __syncthreads();
#else
            matvecs += kh_q2_expm_action(a, b, state, s.buf, s.sbuf, s.inv2, cur, fw_out, N, p.fre, p.fim, dt, nsub, m, wave,
                                         lane, epilogue);
#endif
        } else {
            matvecs += kh_q2_expm_action(a, b, state, s.buf, s.sbuf, s.inv2, cur, fw_out, N, p.fre, p.fim, dt, nsub, m, wave,
                                         lane, [] {});
        }
#ifdef KH_TIMING
        const long long tq2 = clock64();
        t_prop += tq2 - tq1;
#endif
        if constexpr (!ADJ) {
            if (n + 1 < nt - 1) {
                partial_pieces((n + 1) & 1);
                __syncthreads();
            }
        }
#ifdef KH_TIMING
        t_part += clock64() - tq2;
#endif
    }
    }
#ifdef KH_TIMING
    if (tid == 0 && k == 0 && p.stats != nullptr) {
        p.stats[1] = (double)t_ex;
        p.stats[2] = (double)t_prop;
        p.stats[3] = (double)t_coll;  // (wave 0: first poll issued -> sums complete)
    }
#endif
    if (wave == 0 && lane < N) {
        u.phi[(size_t)k * N + lane] = s.buf[cur][lane];
        if constexpr (SO) u.fw_store[((size_t)k * nt + u.n_end) * N + lane] = s.buf[cur][lane];
    }
    if (k == 0 && tid == 0) u.g_a[0] = g_a_loc;
    if (tid == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}

// C = X Y (+ Y X if symmetrize) for N <= 64; grid-stride over the N*N outputs (engine set-up only)
__global__ void kh_q2_product(const cplx *__restrict__ X, const cplx *__restrict__ Y, cplx *__restrict__ C, int N,
                              int symmetrize)
#if KH_DEFINES(KH_TU_MAIN)
{
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < N * N; idx += gridDim.x * blockDim.x) {
        const int i = idx / N, j = idx % N;
        cplx acc = c_make(0.0, 0.0);
        for (int q = 0; q < N; ++q) c_fma(acc, X[(size_t)i * N + q], Y[(size_t)q * N + j]);
        if (symmetrize)
            for (int q = 0; q < N; ++q) c_fma(acc, Y[(size_t)i * N + q], X[(size_t)q * N + j]);
        C[idx] = acc;
    }
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif
