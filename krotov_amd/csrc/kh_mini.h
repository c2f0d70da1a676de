// One-wave-per-objective kernels for small problems: N <= 16, one control, at most 8 objectives
// (BASELINE configs 1-3: two-level systems, single- and two-qubit gates; the reference's own examples).
//
// The register-tile kernels spend ~0.45 us per Taylor phase on a workgroup-wide LDS round trip and barrier
// whatever N is, and the update sweep another ~1.7 us per interval on the cross-workgroup exchange through the
// memory side.  Below N = 16 neither is needed:
//   * a 16 x 16 complex tile is 4 elements per lane of ONE wave (lane = row * 4 + column quarter); a product
//     is 16 FMAs and a DPP quad reduction, and the vector never leaves the registers: the next product fetches
//     its four elements from the lanes that hold them with ds_bpermute -- no LDS memory, no barrier in the series;
//   * the objectives are the waves of ONE workgroup: the update sums cross through LDS with one
//     __syncthreads per interval instead of a global exchange (objectives sharded over GPUs: the peer-window
//     stage of kh_common.h on top, by wave 0).
// The series is the one of kh_tile64q2.h: even terms by the chain B = A^2 = P0 + eps P1 + eps^2 P2, the odd
// terms as ONE product A s, coefficients from the engine's series tables (Taylor, or the shorter real-spectrum
// series for Hermitian operators).  Same arithmetic to rounding; parity-tested against the oracle and the
// reference goldens like every other kernel.
#pragma once

#include "kh_common.h"
#include "kh_generic.h"

#define KH_MINI_N 16
#define KH_MINI_MAXK 8

struct KhMiniLds {
    double2 rows[KH_MAX_DEGREE + 1][KH_Q2_ROWS];  // {r1_p, r2_p} of every degree (kh_common.h, "Series coefficients")
    double c0[KH_MAX_DEGREE + 1];
    double deg[KH_MAX_DEGREE + 1];
    double part[2][KH_MINI_MAXK];        // the objectives' partial sums, by interval parity
    double D[2][2];                      // cross-GPU total + ok flag, by interval parity
};

// the whole series tables -> LDS (all threads of the block; followed by a __syncthreads in the caller)
__device__ __forceinline__ void kh_mini_stage_tables(const KhSweepArgs &p, KhMiniLds &s, int tid, int nthreads) {
    for (int i = tid; i < (KH_MAX_DEGREE + 1) * KH_Q2_ROWS; i += nthreads) {
        s.rows[i / KH_Q2_ROWS][i % KH_Q2_ROWS] = make_double2(p.q2_rows[2 * (size_t)i], p.q2_rows[2 * (size_t)i + 1]);
    }
    for (int i = tid; i <= KH_MAX_DEGREE; i += nthreads) {
        s.c0[i] = p.q2_c0[i];
        s.deg[i] = p.q2_theta[i];
    }
}

// lane (row r = lane / 4, quarter cq = lane % 4) holds elements [r][4 cq + j], j = 0..3
__device__ __forceinline__ void kh_mini_load_tile(const cplx *op, int N, int lane, cplx (&t)[4]) {
    const int r = lane >> 2, cq = lane & 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * cq + j;
        t[j] = (op != nullptr && r < N && c < N) ? op[(size_t)r * N + c] : c_make(0.0, 0.0);
    }
}

// (tile x vector)[r] on the four lanes of row r.  The vector lives in registers, element c on the four lanes
// 4 c .. 4 c + 3 (where the previous product left it): each lane fetches its four elements with ds_bpermute
// (the LDS crossbar, no LDS memory) -- no write -> wait -> read round trip between the terms of the series.
__device__ __forceinline__ cplx kh_mini_matvec(const cplx (&t)[4], cplx v, int lane) {
    const int cq = lane & 3;
    cplx y = c_make(0.0, 0.0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int src = 4 * (4 * cq + j);
        c_fma(y, t[j], c_make(__shfl(v.x, src), __shfl(v.y, src)));
    }
    return c_make(sum4(y.x), sum4(y.y));
}

// The step's coefficients in registers.  A single wave exposes every LDS latency (~100 cycles per dependent
// read), so nothing is read from the tables while the degree, the step and the sub-step count stay what they
// were in the previous interval -- the usual case along a smooth pulse on a uniform grid.
#define KH_MINI_PHASES 8  // phase counts up to this run from registers; beyond: from the LDS tables
struct KhMiniCoef {
    int m, nsub;
    double dt;
    double c0, hr;                 // T_0 = c_0 v;  s starts as h r1_0 v
    double c2[KH_MINI_PHASES];     // f^2 h^2 r2_p
    double hn[KH_MINI_PHASES];     // h r1_{p+1}
};

__device__ __forceinline__ void kh_mini_coefficients(KhMiniCoef &c, const KhMiniLds &s, int m, int nsub, double dt,
                                                     double fre, double fim) {
    if (c.m == m && c.nsub == nsub && c.dt == dt) return;
    c.m = m;
    c.nsub = nsub;
    c.dt = dt;
    const double h = nsub == 1 ? dt : dt / nsub;
    const double f2h2 = (fre * fre - fim * fim) * h * h;
    const double2 *rows = s.rows[m];
    c.c0 = s.c0[m];
    c.hr = h * rows[0].x;
#pragma unroll
    for (int q = 0; q < KH_MINI_PHASES; ++q) {
        c.c2[q] = f2h2 * rows[q].y;
        c.hn[q] = h * rows[q + 1].x;
    }
}

// state <- series(f A dt) state for this wave's objective (`state`: this lane's row of the vector)
__device__ __forceinline__ int kh_mini_expm_action(const cplx (&a)[4], const cplx (&b)[4], cplx &state,
                                                   const KhMiniLds &s, const KhMiniCoef &c, double fre, double fim,
                                                   double dt, int nsub, int m, int lane) {
    const int phases = (m + 1) >> 1;
    for (int sub = 0; sub < nsub; ++sub) {
        cplx term = state;                                   // T_2p, the chain's current vector
        cplx sacc = c_make(c.hr * state.x, c.hr * state.y);  // s = sum_p r1_p h T_2p
        state = c_make(c.c0 * state.x, c.c0 * state.y);
        if (phases <= KH_MINI_PHASES) {
#pragma unroll
            for (int ph = 0; ph < KH_MINI_PHASES; ++ph) {
                if (ph < phases) {
                    const cplx yb = kh_mini_matvec(b, term, lane);
                    term = c_make(c.c2[ph] * yb.x, c.c2[ph] * yb.y);
                    state.x += term.x;
                    state.y += term.y;
                    if (ph + 1 < phases) {
                        sacc.x = fma(c.hn[ph], term.x, sacc.x);
                        sacc.y = fma(c.hn[ph], term.y, sacc.y);
                    }
                }
            }
        } else {
            const double h = nsub == 1 ? dt : dt / nsub;
            const double f2h2 = (fre * fre - fim * fim) * h * h;
            const double2 *rows = s.rows[m];
            for (int ph = 0; ph < phases; ++ph) {
                const double c2 = f2h2 * rows[ph].y;
                const cplx yb = kh_mini_matvec(b, term, lane);
                term = c_make(c2 * yb.x, c2 * yb.y);
                state.x += term.x;
                state.y += term.y;
                if (ph + 1 < phases) {
                    const double hn = h * rows[ph + 1].x;
                    sacc.x = fma(hn, term.x, sacc.x);
                    sacc.y = fma(hn, term.y, sacc.y);
                }
            }
        }
        const cplx odd = c_mul(c_make(fre, fim), kh_mini_matvec(a, sacc, lane));
        state.x += odd.x;
        state.y += odd.y;
    }
    return nsub * (phases + 1);
}

__device__ __forceinline__ void kh_mini_build(double eps, const cplx (&h0)[4], const cplx (&h1)[4], const cplx (&p0)[4],
                                              const cplx (&p1)[4], const cplx (&p2)[4], cplx (&a)[4], cplx (&b)[4]) {
    const double eps2 = eps * eps;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = c_make(fma(eps, h1[j].x, h0[j].x), fma(eps, h1[j].y, h0[j].y));
        b[j] = c_make(fma(eps2, p2[j].x, fma(eps, p1[j].x, p0[j].x)), fma(eps2, p2[j].y, fma(eps, p1[j].y, p0[j].y)));
    }
}

// ---------------------------------------------------------------------------
// plain propagation with storage: one single-wave workgroup per objective
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
kh_mini_sweep_store(KhSweepArgs p, const cplx *const *__restrict__ sq, const double *__restrict__ pulses,
                    const cplx *__restrict__ state_in, cplx *__restrict__ store, cplx *__restrict__ state_out,
                    int direction)
#if KH_DEFINES(KH_TU_MINI)
{
    __shared__ KhMiniLds s;
    const int lane = threadIdx.x, r = lane >> 2, k = blockIdx.x;
    const bool writer = (lane & 3) == 0;
    const int N = p.N, nt = p.nt;
    kh_mini_stage_tables(p, s, lane, 64);
    cplx h0[4], h1[4], p0[4], p1[4], p2[4];
    kh_mini_load_tile(p.ops[(size_t)k * 2], N, lane, h0);
    kh_mini_load_tile(p.ops[(size_t)k * 2 + 1], N, lane, h1);
    kh_mini_load_tile(sq[(size_t)k * 3], N, lane, p0);
    kh_mini_load_tile(sq[(size_t)k * 3 + 1], N, lane, p1);
    kh_mini_load_tile(sq[(size_t)k * 3 + 2], N, lane, p2);
    const double nrm0 = p.op_norms[(size_t)k * 2], nrm1 = p.op_norms[(size_t)k * 2 + 1];
    cplx state = r < N ? state_in[(size_t)k * N + r] : c_make(0.0, 0.0);
    __syncthreads();  // (the tables)
    const bool stores = store != nullptr && writer && r < N;
    if (stores) store[((size_t)k * nt + (direction > 0 ? 0 : nt - 1)) * N + r] = state;
    double matvecs = 0.0;
    KhDegreeCache dc = {12, 1.0, 0.0};
    KhMiniCoef coef;
    coef.m = -1;
    // Per-interval scalars are fetched one interval ahead, and CONSUMED (moved to scalar registers) before the
    // interval's state is stored: loads and stores share one in-order counter on gfx9, so a wait for a load
    // issued after a store also waits for the store's acknowledgement (~1 us) -- most of a step here.
    const int n0 = direction > 0 ? 0 : nt - 2;
    double eps_next = kh_uniform(pulses[n0]), dt_next = kh_uniform(p.dt[n0]);
    for (int step = 0; step < nt - 1; ++step) {
        const int n = direction > 0 ? step : nt - 2 - step;
        const double eps = eps_next, dt = dt_next;
        double eps_ld = 0.0, dt_ld = 0.0;
        if (step + 1 < nt - 1) {
            const int nn = direction > 0 ? n + 1 : n - 1;
            eps_ld = pulses[nn];
            dt_ld = p.dt[nn];
        }
        int nsub, m;
        kh_degree_cached((nrm0 + fabs(eps) * nrm1) * dt, s.deg, p.theta_max, p.inv_theta_max, dc, &nsub, &m);
        kh_mini_coefficients(coef, s, m, nsub, dt, p.fre, p.fim);
        cplx a[4], b[4];
        kh_mini_build(eps, h0, h1, p0, p1, p2, a, b);
        matvecs += kh_mini_expm_action(a, b, state, s, coef, p.fre, p.fim, dt, nsub, m, lane);
        eps_next = kh_uniform(eps_ld);
        dt_next = kh_uniform(dt_ld);
        if (stores) store[((size_t)k * nt + (direction > 0 ? n + 1 : n)) * N + r] = state;
    }
    if (state_out != nullptr && writer && r < N) state_out[(size_t)k * N + r] = state;
    if (lane == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif

// ---------------------------------------------------------------------------
// forward sweep with sequential pulse update (optimize.py:444-508): ONE workgroup, wave k = objective k
// ---------------------------------------------------------------------------
template <bool SO>
__global__ void __launch_bounds__(64 * KH_MINI_MAXK)
kh_mini_forward_update(KhSweepArgs p, const cplx *const *__restrict__ sq, KhUpdateArgs u, KhExchange ex) {
    __shared__ KhMiniLds s;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane >> 2, k = w;
    const bool writer = (lane & 3) == 0;
    const int N = p.N, nt = p.nt, K = p.K;
    kh_mini_stage_tables(p, s, tid, blockDim.x);
    cplx h0[4], h1[4], p0[4], p1[4], p2[4];
    kh_mini_load_tile(p.ops[(size_t)k * 2], N, lane, h0);
    kh_mini_load_tile(p.ops[(size_t)k * 2 + 1], N, lane, h1);  // also dH/d eps (mu.py:123-134)
    kh_mini_load_tile(sq[(size_t)k * 3], N, lane, p0);
    kh_mini_load_tile(sq[(size_t)k * 3 + 1], N, lane, p1);
    kh_mini_load_tile(sq[(size_t)k * 3 + 2], N, lane, p2);
    const double nrm0 = p.op_norms[(size_t)k * 2], nrm1 = p.op_norms[(size_t)k * 2 + 1];
    const double chi_norm = u.chi_norms[k];
    cplx state = r < N ? u.phi[(size_t)k * N + r] : c_make(0.0, 0.0);
    __syncthreads();  // (the tables)
    double matvecs = 0.0;

    // this objective's  ||chi|| Im(mu <bra(t_n)|H1 phi>)  -> part[n & 1][k]; phi's row of this lane in `state`
    // chi(t_n) (and, second order, phi_prev(t_n), sigma_n) of this lane's row, fetched one interval ahead
    cplx chi = c_make(0.0, 0.0), prev = c_make(0.0, 0.0);
    double sig = 0.0;
    auto load_bra = [&](int n) {
        if (writer && r < N) {
            chi = u.chi_store[((size_t)k * nt + n) * N + r];
            if constexpr (SO) prev = u.fw_prev[((size_t)k * nt + n) * N + r];
        }
        if constexpr (SO) sig = u.sigma[n];
    };
    auto partial = [&](int n) {
        const cplx y = kh_mini_matvec(h1, state, lane);
        cplx bra = chi;
        if constexpr (SO) {  // bra = chi + sigma / (2 ||chi||) (phi - phi_prev)  (optimize.py:468-469)
            if (writer && r < N) {
                const double hs = 0.5 * sig / chi_norm;
                bra = c_make(fma(hs, state.x - prev.x, chi.x), fma(hs, state.y - prev.y, chi.y));
            }
        }
        cplx ov = c_make(0.0, 0.0);
        c_fma_conj(ov, bra, y);
        const double v = sum64(u.mu_re * ov.y + u.mu_im * ov.x);
        if (lane == 0) s.part[n & 1][w] = chi_norm * v;
        matvecs += 1.0;
    };

    if (u.n_begin < nt - 1) {
        load_bra(u.n_begin);
        partial(u.n_begin);
    }
    __syncthreads();
    double g_a_loc = 0.0;
    const double lam = u.lambda[0];
    KhDegreeCache dc = {12, 1.0, 0.0};
    KhMiniCoef coef;
    coef.m = -1;
    double dt_next = p.dt[u.n_begin], guess_next = u.guess[u.n_begin], shape_next = u.shape[u.n_begin];
    for (int n = u.n_begin; n < u.n_end; ++n) {
        const int par = n & 1;
        const double dt = dt_next, guess = guess_next, shape = shape_next;
        if (n + 1 < nt - 1) {  // next interval's scalars and co-state row: in flight during this interval
            dt_next = p.dt[n + 1];
            guess_next = u.guess[n + 1];
            shape_next = u.shape[n + 1];
            load_bra(n + 1);
        }
        // ---- cross-objective sum (optimize.py:470): through LDS, in objective order ----
        double d1 = 0.0;
        for (int q = 0; q < K; ++q) d1 += s.part[par][q];
        if (ex.world > 1) {  // objectives sharded over GPUs: second stage through the peer windows
            if (w == 0) {
                const unsigned int epoch = ex.epoch_base + (unsigned)(n + 1);
                double D[1] = {d1};
                if (n != ex.fail_at) kh_p2p_publish(ex, par, 1, lane, D, epoch);
                const bool ok = kh_p2p_gather<1>(ex, par, 1, epoch, lane, D);
                if (lane == 0) {
                    s.D[par][0] = D[0];
                    s.D[par][1] = ok ? 1.0 : 0.0;
                }
            }
            __syncthreads();
            if (s.D[par][1] == 0.0) return;
            d1 = s.D[par][0];
        }
        // ---- pulse update (optimize.py:471-477) ----
        const double stepw = shape / lam;
        const double eps = guess + stepw * d1;
        g_a_loc += stepw * (d1 * d1) * dt;
        if (tid == 0) u.opt[n] = eps;
        if constexpr (SO) {
            if (writer && r < N) u.fw_store[((size_t)k * nt + n) * N + r] = state;
        }
        // ---- propagate over interval n with the updated pulse (optimize.py:479-491) ----
        int nsub, m;
        kh_degree_cached((nrm0 + fabs(eps) * nrm1) * dt, s.deg, p.theta_max, p.inv_theta_max, dc, &nsub, &m);
        kh_mini_coefficients(coef, s, m, nsub, dt, p.fre, p.fim);
        cplx a[4], b[4];
        kh_mini_build(eps, h0, h1, p0, p1, p2, a, b);
        matvecs += kh_mini_expm_action(a, b, state, s, coef, p.fre, p.fim, dt, nsub, m, lane);
        if (n + 1 < nt - 1) partial(n + 1);
        __syncthreads();  // everybody's partial sum of the next interval is in LDS
    }
    if (writer && r < N) {
        u.phi[(size_t)k * N + r] = state;
        if constexpr (SO) u.fw_store[((size_t)k * nt + u.n_end) * N + r] = state;
    }
    if (tid == 0) u.g_a[0] = g_a_loc;
    if (lane == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}

// ===========================================================================
// N <= 4, at most 4 objectives: the whole problem in ONE wave
// ===========================================================================
// One matrix element per lane: lane = 16 o + 4 r + c holds element [r][c] of objective o's tiles; vector element
// e of objective o sits on lanes 16 o + 4 e .. + 3.  A product is one complex multiply, a DPP quad sum and one
// ds_bpermute fetch per lane; the update sum over rows AND objectives is a single wavefront reduction (no LDS,
// no barrier anywhere; objectives sharded over GPUs: the peer-window stage by the same wave).  The series'
// degree is the largest any of the wave's objectives needs (a higher degree also serves a smaller norm).
#define KH_QUAD_N 4
#define KH_QUAD_MAXK 4

__device__ __forceinline__ cplx kh_quad_load(const cplx *op, int N, int r, int c, bool valid) {
    return (valid && op != nullptr && r < N && c < N) ? op[(size_t)r * N + c] : c_make(0.0, 0.0);
}

// (tile x vector)[r] on the four lanes of row r of every objective
__device__ __forceinline__ cplx kh_quad_matvec(cplx t, cplx v, int lane) {
    const int src = (lane & 48) | ((lane & 3) << 2);  // the lanes holding element c of this lane's objective
    const cplx x = c_make(__shfl(v.x, src), __shfl(v.y, src));
    const cplx y = c_mul(t, x);
    return c_make(sum4(y.x), sum4(y.y));
}

__device__ __forceinline__ int kh_quad_expm_action(cplx a, cplx b, cplx &state, const KhMiniLds &s, const KhMiniCoef &c,
                                                   double fre, double fim, double dt, int nsub, int m, int lane) {
    const int phases = (m + 1) >> 1;
    for (int sub = 0; sub < nsub; ++sub) {
        cplx term = state;
        cplx sacc = c_make(c.hr * state.x, c.hr * state.y);
        state = c_make(c.c0 * state.x, c.c0 * state.y);
        if (phases <= KH_MINI_PHASES) {
#pragma unroll
            for (int ph = 0; ph < KH_MINI_PHASES; ++ph) {
                if (ph < phases) {
                    const cplx yb = kh_quad_matvec(b, term, lane);
                    term = c_make(c.c2[ph] * yb.x, c.c2[ph] * yb.y);
                    state.x += term.x;
                    state.y += term.y;
                    if (ph + 1 < phases) {
                        sacc.x = fma(c.hn[ph], term.x, sacc.x);
                        sacc.y = fma(c.hn[ph], term.y, sacc.y);
                    }
                }
            }
        } else {
            const double h = nsub == 1 ? dt : dt / nsub;
            const double f2h2 = (fre * fre - fim * fim) * h * h;
            const double2 *rows = s.rows[m];
            for (int ph = 0; ph < phases; ++ph) {
                const double c2 = f2h2 * rows[ph].y;
                const cplx yb = kh_quad_matvec(b, term, lane);
                term = c_make(c2 * yb.x, c2 * yb.y);
                state.x += term.x;
                state.y += term.y;
                if (ph + 1 < phases) {
                    const double hn = h * rows[ph + 1].x;
                    sacc.x = fma(hn, term.x, sacc.x);
                    sacc.y = fma(hn, term.y, sacc.y);
                }
            }
        }
        const cplx odd = c_mul(c_make(fre, fim), kh_quad_matvec(a, sacc, lane));
        state.x += odd.x;
        state.y += odd.y;
    }
    return nsub * (phases + 1);
}

// largest value of a lane-varying non-negative double over the wave (same value in every lane)
__device__ __forceinline__ double kh_wave_max(double v) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}

struct KhQuadTiles {
    cplx h0, h1, p0, p1, p2;
    double nrm0, nrm1;
};

__device__ __forceinline__ KhQuadTiles kh_quad_load_tiles(const KhSweepArgs &p, const cplx *const *sq, int k, int r, int c,
                                                          bool valid) {
    KhQuadTiles t;
    const size_t kk = valid ? (size_t)k : 0;
    t.h0 = kh_quad_load(p.ops[kk * 2], p.N, r, c, valid);
    t.h1 = kh_quad_load(p.ops[kk * 2 + 1], p.N, r, c, valid);
    t.p0 = kh_quad_load(sq[kk * 3], p.N, r, c, valid);
    t.p1 = kh_quad_load(sq[kk * 3 + 1], p.N, r, c, valid);
    t.p2 = kh_quad_load(sq[kk * 3 + 2], p.N, r, c, valid);
    t.nrm0 = valid ? p.op_norms[kk * 2] : 0.0;
    t.nrm1 = valid ? p.op_norms[kk * 2 + 1] : 0.0;
    return t;
}

__device__ __forceinline__ void kh_quad_build(const KhQuadTiles &t, double eps, cplx &a, cplx &b) {
    const double eps2 = eps * eps;
    a = c_make(fma(eps, t.h1.x, t.h0.x), fma(eps, t.h1.y, t.h0.y));
    b = c_make(fma(eps2, t.p2.x, fma(eps, t.p1.x, t.p0.x)), fma(eps2, t.p2.y, fma(eps, t.p1.y, t.p0.y)));
}

__global__ void __launch_bounds__(64)
kh_quad_sweep_store(KhSweepArgs p, const cplx *const *__restrict__ sq, const double *__restrict__ pulses,
                    const cplx *__restrict__ state_in, cplx *__restrict__ store, cplx *__restrict__ state_out,
                    int direction)
#if KH_DEFINES(KH_TU_MINI)
{
    __shared__ KhMiniLds s;
    const int lane = threadIdx.x, k = lane >> 4, r = (lane >> 2) & 3, c = lane & 3;
    const int N = p.N, nt = p.nt;
    const bool valid = k < p.K, owner = valid && c == 0 && r < N;
    kh_mini_stage_tables(p, s, lane, 64);
    const KhQuadTiles t = kh_quad_load_tiles(p, sq, k, r, c, valid);
    cplx state = (valid && r < N) ? state_in[(size_t)k * N + r] : c_make(0.0, 0.0);
    __syncthreads();  // (the tables)
    const bool stores = store != nullptr && owner;
    if (stores) store[((size_t)k * nt + (direction > 0 ? 0 : nt - 1)) * N + r] = state;
    double matvecs = 0.0;
    KhDegreeCache dc = {12, 1.0, 0.0};
    KhMiniCoef coef;
    coef.m = -1;
    const int n0 = direction > 0 ? 0 : nt - 2;
    double eps_next = kh_uniform(pulses[n0]), dt_next = kh_uniform(p.dt[n0]);
    for (int step = 0; step < nt - 1; ++step) {
        const int n = direction > 0 ? step : nt - 2 - step;
        const double eps = eps_next, dt = dt_next;
        double eps_ld = 0.0, dt_ld = 0.0;
        if (step + 1 < nt - 1) {
            const int nn = direction > 0 ? n + 1 : n - 1;
            eps_ld = pulses[nn];
            dt_ld = p.dt[nn];
        }
        int nsub, m;
        const double theta = kh_uniform(kh_wave_max((t.nrm0 + fabs(eps) * t.nrm1) * dt));
        kh_degree_cached(theta, s.deg, p.theta_max, p.inv_theta_max, dc, &nsub, &m);
        kh_mini_coefficients(coef, s, m, nsub, dt, p.fre, p.fim);
        cplx a, b;
        kh_quad_build(t, eps, a, b);
        matvecs += kh_quad_expm_action(a, b, state, s, coef, p.fre, p.fim, dt, nsub, m, lane);
        eps_next = kh_uniform(eps_ld);
        dt_next = kh_uniform(dt_ld);
        if (stores) store[((size_t)k * nt + (direction > 0 ? n + 1 : n)) * N + r] = state;
    }
    if (state_out != nullptr && owner) state_out[(size_t)k * N + r] = state;
    if (lane == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs * p.K);
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif

template <bool SO>
__global__ void __launch_bounds__(64)
kh_quad_forward_update(KhSweepArgs p, const cplx *const *__restrict__ sq, KhUpdateArgs u, KhExchange ex) {
    __shared__ KhMiniLds s;
    const int lane = threadIdx.x, k = lane >> 4, r = (lane >> 2) & 3, c = lane & 3;
    const int N = p.N, nt = p.nt;
    const bool valid = k < p.K, owner = valid && c == 0 && r < N;
    kh_mini_stage_tables(p, s, lane, 64);
    const KhQuadTiles t = kh_quad_load_tiles(p, sq, k, r, c, valid);
    const double chi_norm = valid ? u.chi_norms[k] : 0.0;
    cplx state = (valid && r < N) ? u.phi[(size_t)k * N + r] : c_make(0.0, 0.0);
    __syncthreads();  // (the tables)
    double matvecs = 0.0;

    cplx chi = c_make(0.0, 0.0), prev = c_make(0.0, 0.0);
    double sig = 0.0;
    auto load_bra = [&](int n) {
        if (owner) {
            chi = u.chi_store[((size_t)k * nt + n) * N + r];
            if constexpr (SO) prev = u.fw_prev[((size_t)k * nt + n) * N + r];
        }
        if constexpr (SO) sig = u.sigma[n];
    };
    // sum over rows and objectives of ||chi_k|| Im(mu <bra_k|H1 phi_k>): one wavefront reduction
    auto update_sum = [&]() {
        const cplx y = kh_quad_matvec(t.h1, state, lane);
        cplx bra = chi;
        if constexpr (SO) {
            if (owner) {
                const double hs = 0.5 * sig / chi_norm;
                bra = c_make(fma(hs, state.x - prev.x, chi.x), fma(hs, state.y - prev.y, chi.y));
            }
        }
        cplx ov = c_make(0.0, 0.0);
        if (owner) c_fma_conj(ov, bra, y);
        matvecs += 1.0;
        return sum64(chi_norm * (u.mu_re * ov.y + u.mu_im * ov.x));
    };

    double d_next = 0.0;
    if (u.n_begin < nt - 1) {
        load_bra(u.n_begin);
        d_next = update_sum();
    }
    double g_a_loc = 0.0;
    const double lam = u.lambda[0];
    KhDegreeCache dc = {12, 1.0, 0.0};
    KhMiniCoef coef;
    coef.m = -1;
    double dt_next = p.dt[u.n_begin], guess_next = u.guess[u.n_begin], shape_next = u.shape[u.n_begin];
    for (int n = u.n_begin; n < u.n_end; ++n) {
        const int par = n & 1;
        const double dt = dt_next, guess = guess_next, shape = shape_next;
        if (n + 1 < nt - 1) {
            dt_next = p.dt[n + 1];
            guess_next = u.guess[n + 1];
            shape_next = u.shape[n + 1];
            load_bra(n + 1);
        }
        double d1 = d_next;
        if (ex.world > 1) {  // objectives sharded over GPUs: second stage through the peer windows
            const unsigned int epoch = ex.epoch_base + (unsigned)(n + 1);
            double D[1] = {d1};
            if (n != ex.fail_at) kh_p2p_publish(ex, par, 1, lane, D, epoch);
            if (!kh_p2p_gather<1>(ex, par, 1, epoch, lane, D)) return;
            d1 = D[0];
        }
        const double stepw = shape / lam;
        const double eps = kh_uniform(guess + stepw * d1);
        g_a_loc += stepw * (d1 * d1) * dt;
        if (lane == 0) u.opt[n] = eps;
        if constexpr (SO) {
            if (owner) u.fw_store[((size_t)k * nt + n) * N + r] = state;
        }
        int nsub, m;
        const double theta = kh_uniform(kh_wave_max((t.nrm0 + fabs(eps) * t.nrm1) * dt));
        kh_degree_cached(theta, s.deg, p.theta_max, p.inv_theta_max, dc, &nsub, &m);
        kh_mini_coefficients(coef, s, m, nsub, dt, p.fre, p.fim);
        cplx a, b;
        kh_quad_build(t, eps, a, b);
        matvecs += kh_quad_expm_action(a, b, state, s, coef, p.fre, p.fim, dt, nsub, m, lane);
        if (n + 1 < nt - 1) d_next = update_sum();
    }
    if (owner) {
        u.phi[(size_t)k * N + r] = state;
        if constexpr (SO) u.fw_store[((size_t)k * nt + u.n_end) * N + r] = state;
    }
    if (lane == 0) u.g_a[0] = g_a_loc;
    if (lane == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs * p.K);
}
