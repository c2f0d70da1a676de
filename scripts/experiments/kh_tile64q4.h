// EXPERIMENT (round 3): the two-terms-per-phase plain sweep of kh_tile64q2.h with 1024-thread
// workgroups -- 16 waves, four per SIMD, at most 128 VGPRs -- instead of 512.  A lane owns 1 row x 4 columns of
// every tile (16 dwords), so H1, P1, P2, A and B need 80 VGPRs; a wave owns 4 rows; the row sum runs over 16 lanes:
// v_mfma_f64_4x4x4 adds the four lanes 16 k + i, two row rotations add the four 4-lane blocks.
// Question asked: do four waves per SIMD hide the read -> FMA -> reduce -> write -> barrier latency of a phase that
// two cannot?  Selected with KH_Q4=1 for the plain sweeps only (backward sweep / iteration-0 forward sweep); the
// answer and the numbers are in DESIGN.md.
#pragma once

// (included from krotov_hip.hip in a -DKH_WITH_Q4 build, after the kernel headers of krotov_amd/csrc)

#define KH_Q4_THREADS 1024
#define KH_Q4_TILE_ELEMS (4 * KH_Q4_THREADS)  // complex elements of one 64x64 operator, lane-linear

struct KhQ4Lanes {
    static __device__ __forceinline__ int cg(int lane) { return (lane >> 4) + 4 * ((lane >> 2) & 3); }
    static __device__ __forceinline__ int row_in(int lane) { return lane & 3; }
    static __device__ __forceinline__ int row_out(int lane) { return lane >> 4; }
    // scale * (sum of v over the 16 column groups of a row), on all 16 lanes of the row's output group
    static __device__ __forceinline__ double rowsum(double v, double scale) {
        double d = __builtin_amdgcn_mfma_f64_4x4x4f64(v, scale, 0.0, 0, 0, 0);
        d += dpp_move<KH_DPP_ROR4>(d);
        return d + dpp_move<KH_DPP_ROR8>(d);
    }
};

struct KhQ4Lds {
    cplx *h0;    // [4][512]
    cplx *p0;    // [4][512]
    cplx (*buf)[KH_TILE_N];  // [2][64]
    cplx *chib;  // [64] chi(t_{n+1}) for the adjoint-side partial sums
    cplx *sbuf;  // [64] the vector s of kh_q4_expm_action
    double *red; // [2][16 waves][2]
    double *D;   // [2][2]
    double2 *inv2;  // [KH_Q2_ROWS] the series' rows {r1_p, r2_p} of the current degree (Taylor: {1/(2p+1),
                    // 1/((2p+1)(2p+2))}); LDS: no SMEM loads in the phase loop.  [KH_Q2_ROWS]: c_0 (in .x)
    double *deg;    // [KH_MAX_DEGREE+1] copy of the degree-threshold table
};

__host__ __device__ inline size_t kh_q4_lds_bytes() {
    return (size_t)2 * KH_Q4_TILE_ELEMS * sizeof(cplx) + 4 * KH_TILE_N * sizeof(cplx) + (2 * 16 * 2 + 4) * sizeof(double) +
           (KH_MAX_DEGREE / 2 + 1) * sizeof(double2) + (KH_MAX_DEGREE + 2) * sizeof(double);
}

__device__ __forceinline__ KhQ4Lds kh_q4_carve(char *smem) {
    KhQ4Lds s;
    s.h0 = (cplx *)smem;
    s.p0 = s.h0 + KH_Q4_TILE_ELEMS;
    s.buf = (cplx(*)[KH_TILE_N])(s.p0 + KH_Q4_TILE_ELEMS);
    s.chib = (cplx *)(s.buf + 2);
    s.sbuf = s.chib + KH_TILE_N;
    s.red = (double *)(s.sbuf + KH_TILE_N);
    s.D = s.red + 2 * 16 * 2;
    s.inv2 = (double2 *)(s.D + 4);
    s.deg = (double *)(s.inv2 + KH_MAX_DEGREE / 2 + 1);
    return s;
}

// this lane's 8 elements of an operator (row wave*8 + row_in, columns cg + 8 j)
__device__ __forceinline__ void kh_q4_load_tile(const cplx *op, int N, int wave, int lane, cplx (&t)[4]) {
    const int row = wave * 4 + KhQ4Lanes::row_in(lane), cg = KhQ4Lanes::cg(lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = cg + 16 * j;
        t[j] = (op != nullptr && row < N && col < N) ? op[(size_t)row * N + col] : c_make(0.0, 0.0);
    }
}

__device__ __forceinline__ void kh_q4_stage_tile(const cplx *op, int N, int wave, int lane, int tid, cplx *dst) {
    cplx t[4];
    kh_q4_load_tile(op, N, wave, lane, t);
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j * KH_Q4_THREADS + tid] = t[j];
}

// the series' rows of degree m -> LDS (workgroup-uniform m; called between intervals, contains a barrier)
__device__ __forceinline__ void kh_q4_load_rows(const KhSweepArgs &p, const KhQ4Lds &s, int m, int tid) {
    __syncthreads();  // (no phase is still reading the previous rows)
    if (tid < KH_Q2_ROWS) {
        const double *r = p.q2_rows + ((size_t)m * KH_Q2_ROWS + tid) * 2;
        s.inv2[tid] = make_double2(r[0], r[1]);
    }
    if (tid == KH_Q2_ROWS) s.inv2[KH_Q2_ROWS] = make_double2(p.q2_c0[m], 0.0);
    __syncthreads();
}

// The plain sweeps advance resident tiles too, entirely in registers (H1, P1, P2 are there already):
//   A += (eps - eps') H1,   B += (eps - eps') P1 + (eps^2 - eps'^2) P2
// and restart from H0, P0 in LDS every KH_Q2_REFRESH intervals: no LDS traffic per interval instead of the 128 KiB
// a workgroup read to rebuild its two tiles (1 024 cycles of the LDS pipe in front of the first phase).
__device__ __forceinline__ void kh_q4_restart_lds(const KhQ4Lds &s, int tid, cplx (&a)[4], cplx (&b)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = s.h0[j * KH_Q4_THREADS + tid];
        b[j] = s.p0[j * KH_Q4_THREADS + tid];
    }
}
__device__ __forceinline__ void kh_q4_advance_reg(double eps, double eps_prev, const cplx (&h1)[4], const cplx (&p1)[4],
                                                  const cplx (&p2)[4], cplx (&a)[4], cplx (&b)[4]) {
    const double e1 = eps - eps_prev, e2 = e1 * (eps + eps_prev);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j].x = fma(e1, h1[j].x, a[j].x);
        a[j].y = fma(e1, h1[j].y, a[j].y);
        b[j].x = fma(e2, p2[j].x, fma(e1, p1[j].x, b[j].x));
        b[j].y = fma(e2, p2[j].y, fma(e1, p1[j].y, b[j].y));
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
__device__ __forceinline__ int kh_q4_expm_action(const cplx (&a)[4], const cplx (&b)[4], cplx &state,
                                                 cplx (*buf)[KH_TILE_N], cplx *sbuf, const double2 *inv2, int &cur,
                                                 cplx *store_in, int N, double fre, double fim,
                                                 double dt, int nsub, int m, int wave, int lane,
                                                 Epilogue epilogue) {
    const int cg = KhQ4Lanes::cg(lane), row = wave * 4 + KhQ4Lanes::row_out(lane);
    const bool writer = (lane & 15) == 0;
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
            cplx xv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = buf[cur][cg + 16 * j];
            const bool last = (ph + 1 == phases);
            cplx yb = c_make(0.0, 0.0);
#pragma unroll
            for (int j = 0; j < 4; ++j) c_fma(yb, b[j], xv[j]);
            const double t2x = KhQ4Lanes::rowsum(yb.x, c2), t2y = KhQ4Lanes::rowsum(yb.y, c2);
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
                cplx sv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) sv[j] = sbuf[cg + 16 * j];
                cplx ya = c_make(0.0, 0.0);
#pragma unroll
                for (int j = 0; j < 4; ++j) c_fma(ya, a[j], sv[j]);
                const cplx odd = c_mul(c_make(fre, fim), c_make(KhQ4Lanes::rowsum(ya.x, 1.0), KhQ4Lanes::rowsum(ya.y, 1.0)));
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
__global__ void __launch_bounds__(KH_Q4_THREADS)
kh_q4_sweep_store(KhSweepArgs p, const cplx *const *__restrict__ sq, const double *__restrict__ pulses,
                  const cplx *__restrict__ state_in, cplx *__restrict__ store, cplx *__restrict__ state_out,
                  int direction) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KhQ4Lds s = kh_q4_carve(smem);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid <= KH_MAX_DEGREE) s.deg[tid] = p.q2_theta[tid];
    const int row = wave * 4 + KhQ4Lanes::row_out(lane);  // the row whose sums/state this lane holds
    const bool writer = (lane & 15) == 0;
    const int N = p.N, nt = p.nt;
    double matvecs = 0.0;
    for (int k = blockIdx.x; k < p.K; k += gridDim.x) {
        const cplx *const *ops_k = p.ops + (size_t)k * 2;
        const cplx *const *sq_k = sq + (size_t)k * 3;
        __syncthreads();  // previous objective's readers are done with LDS
        kh_q4_stage_tile(ops_k[0], N, wave, lane, tid, s.h0);
        kh_q4_stage_tile(sq_k[0], N, wave, lane, tid, s.p0);
        cplx h1[4], p1[4], p2[4];
        kh_q4_load_tile(ops_k[1], N, wave, lane, h1);
        kh_q4_load_tile(sq_k[1], N, wave, lane, p1);
        kh_q4_load_tile(sq_k[2], N, wave, lane, p2);
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
        cplx a[4], b[4];
        double eps_prev = 0.0;
        for (int step0 = 0; step0 < nt - 1; step0 += KH_Q2_REFRESH) {
        // (restart outside the interval loop: its body keeps ONE definition of the tiles, see kh_q4_forward_update)
        kh_q4_restart_lds(s, tid, a, b);
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
                kh_q4_load_rows(p, s, m, tid);
                m_rows = m;
            }
            kh_q4_advance_reg(eps, eps_prev, h1, p1, p2, a, b);
            eps_prev = eps;
            cplx *store_in =
                store == nullptr ? nullptr : store + ((size_t)k * nt + (direction > 0 ? n : n + 1)) * N;
#ifdef KH_TIMING
            const long long tq1 = clock64();
            t_build += tq1 - tq0;
#endif
            matvecs += kh_q4_expm_action(a, b, state, s.buf, s.sbuf, s.inv2, cur, store_in, N, p.fre, p.fim, dt, nsub, m,
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

