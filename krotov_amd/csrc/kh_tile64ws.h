// Wave-specialised two-terms-per-phase kernels: N <= 64, one control (L = 1).
//
// Same mathematics as kh_tile64q2.h (A = H0 + eps H1 and B = A^2 = P0 + eps P1 +
// eps^2 P2 resident on chip, two Taylor terms per phase), but the two halves of
// a phase run on DIFFERENT waves of the same SIMDs:
//
//   waves 0-3 ("B waves", raised priority): hold B.  They own the dependent
//     chain  t_{2p+2} = c2 B t_{2p}:  read t_{2p} (LDS broadcast), 64 FMAs/lane,
//     DPP row reduction, write t_{2p+2}, barrier.  Nothing else sits on the
//     critical path of an interval.
//   waves 4-7 ("A waves"): hold A.  They read the same t_{2p}, form
//     A t_{2p} and accumulate the odd terms  sA += c1 (A t_{2p})  UNREDUCED in
//     registers; one reduction + one LDS hand-over per interval.  Their FMAs
//     fill the issue slots the B wave of the same SIMD leaves while it waits on
//     LDS / DPP latencies, so both fp64 FMA streams overlap instead of adding.
//
// Each wave owns 16 rows x 64 columns of its matrix: lane = (row group lane/8,
// column group lane%8), 2 rows x 8 columns per lane (64 dwords per operator tile).
// B waves: B, P1, P2 tiles (192 VGPRs) + vector (32); A waves: A, H1 (128) + 32.
// H0 / P0 tiles sit in LDS (2 x 64 KiB, lane-linear) and are read once per
// interval when A and B are rebuilt.  Barriers per interval: phases + 1.
#pragma once

#include "kh_common.h"
#include "kh_generic.h"
#include "kh_tile64.h"

#define KH_WS_THREADS 512
#define KH_WS_GROUP 256                      // threads per wave group
#define KH_WS_TILE_ELEMS (16 * KH_WS_GROUP)  // complex elements of one 64x64 operator

struct KhWsLds {
    cplx *h0;                 // [16][256] A waves' drift tile
    cplx *p0;                 // [16][256] B waves' P0 tile
    cplx (*buf)[KH_TILE_N];   // [2][64] Taylor-term ping-pong
    cplx *odd;                // [64] reduced odd-term sum of the interval (A waves -> B waves)
    double *red;              // [2][4]  partial-sum pieces of the A waves
    double *D;                // [2][2]
    double *deg;              // [KH_MAX_DEGREE+1] copy of the degree-threshold table (no SMEM/global loads
                              // on the per-interval critical path)
    double2 *inv2;            // [KH_MAX_DEGREE/2] {1/(2p+1), 1/(2p+2)}: no scalar (SMEM) loads in the phase
                              // loop -- they share lgkmcnt with the LDS traffic and force full drains
};

__host__ __device__ inline size_t kh_ws_lds_bytes() {
    return (size_t)2 * KH_WS_TILE_ELEMS * sizeof(cplx) + 3 * KH_TILE_N * sizeof(cplx) + (2 * 4 + 4) * sizeof(double) +
           (KH_MAX_DEGREE / 2) * sizeof(double2) + (KH_MAX_DEGREE + 2) * sizeof(double);
}

__device__ __forceinline__ KhWsLds kh_ws_carve(char *smem) {
    KhWsLds s;
    s.h0 = (cplx *)smem;
    s.p0 = s.h0 + KH_WS_TILE_ELEMS;
    s.buf = (cplx(*)[KH_TILE_N])(s.p0 + KH_WS_TILE_ELEMS);
    s.odd = (cplx *)(s.buf + 2);
    s.red = (double *)(s.odd + KH_TILE_N);
    s.D = s.red + 2 * 4;
    s.inv2 = (double2 *)(s.D + 4);
    s.deg = (double *)(s.inv2 + KH_MAX_DEGREE / 2);
    return s;
}

// rows owned by a lane: gw = wave within its group (0..3), r = 0, 1
__device__ __forceinline__ int kh_ws_row(int gw, int lane, int r) { return gw * 16 + r * 8 + (lane >> 3); }

__device__ __forceinline__ void kh_ws_load_tile(const cplx *op, int N, int gw, int lane, cplx (&t)[2][8]) {
    const int cg = lane & 7;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = kh_ws_row(gw, lane, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = cg + 8 * j;
            t[r][j] = (op != nullptr && row < N && col < N) ? op[(size_t)row * N + col] : c_make(0.0, 0.0);
        }
    }
}

// gtid = thread index within the wave group (0..255)
__device__ __forceinline__ void kh_ws_stage_tile(const cplx *op, int N, int gw, int lane, int gtid, cplx *dst) {
    cplx t[2][8];
    kh_ws_load_tile(op, N, gw, lane, t);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[(r * 8 + j) * KH_WS_GROUP + gtid] = t[r][j];
}

__device__ __forceinline__ void kh_ws_read_x(const cplx *x, int cg, cplx (&xv)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = x[cg + 8 * j];
}


// Generator rebuilds.  The LDS reads are issued four at a time (a compiler
// fence between chunks): hoisting all 16 ds_read_b128 of a tile ahead of the
// FMAs costs 64 extra live VGPRs and pushes the kernel into scratch spills.
#define KH_WS_FENCE() asm volatile("" ::: "memory")

__device__ __forceinline__ void kh_ws_build_B(const cplx *p0, int gtid, double eps, double eps2,
                                              const cplx (&p1)[2][8], const cplx (&p2)[2][8], cplx (&b)[2][8]) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const cplx q0 = p0[(r * 8 + j) * KH_WS_GROUP + gtid];
            b[r][j].x = fma(eps2, p2[r][j].x, fma(eps, p1[r][j].x, q0.x));
            b[r][j].y = fma(eps2, p2[r][j].y, fma(eps, p1[r][j].y, q0.y));
            if ((j & 3) == 3) KH_WS_FENCE();
        }
}

__device__ __forceinline__ void kh_ws_build_A(const cplx *h0t, int gtid, double eps, const cplx (&h1)[2][8],
                                              cplx (&a)[2][8]) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const cplx h0 = h0t[(r * 8 + j) * KH_WS_GROUP + gtid];
            a[r][j].x = fma(eps, h1[r][j].x, h0.x);
            a[r][j].y = fma(eps, h1[r][j].y, h0.y);
            if ((j & 3) == 3) KH_WS_FENCE();
        }
}

// ---------------------------------------------------------------------------
// one interval of the B waves: even terms, the dependent chain
// ---------------------------------------------------------------------------
// ev[r]: running state rows (reduced; identical in the 8 lanes of a row).
__device__ __forceinline__ void kh_ws_interval_B(const KhWsLds &s, const cplx (&b)[2][8], cplx (&ev)[2], int &cur,
                                                 double f2, double h, int phases, int gw, int lane) {
    const int cg = lane & 7;
    for (int ph = 0; ph < phases; ++ph) {
        const double2 iv = s.inv2[ph];
        const double c2 = f2 * (h * iv.x) * (h * iv.y);
        cplx xv[8];
        kh_ws_read_x(s.buf[cur], cg, xv);
        cplx y0 = c_make(0.0, 0.0), y1 = c_make(0.0, 0.0);
#ifndef KH_DBG_NO_B_FMA
        {
            // 8 independent accumulator chains (even / odd columns): a single wave
            // cannot hide the fp64 FMA latency with only 4
            cplx z0 = c_make(0.0, 0.0), z1 = c_make(0.0, 0.0);
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                c_fma(y0, b[0][j], xv[j]);
                c_fma(y1, b[1][j], xv[j]);
                c_fma(z0, b[0][j + 1], xv[j + 1]);
                c_fma(z1, b[1][j + 1], xv[j + 1]);
            }
            y0.x += z0.x;
            y0.y += z0.y;
            y1.x += z1.x;
            y1.y += z1.y;
        }
#else
        y0 = xv[0]; y1 = xv[1];
#endif
#ifndef KH_DBG_NO_REDUCE
        const double t0x = c2 * sum8(y0.x), t0y = c2 * sum8(y0.y);
        const double t1x = c2 * sum8(y1.x), t1y = c2 * sum8(y1.y);
#else
        const double t0x = c2 * y0.x, t0y = c2 * y0.y, t1x = c2 * y1.x, t1y = c2 * y1.y;
#endif
        ev[0].x += t0x;
        ev[0].y += t0y;
        ev[1].x += t1x;
        ev[1].y += t1y;
        const bool last = (ph + 1 == phases);
        if (!last) {
            if (cg == 0) {
                s.buf[cur ^ 1][kh_ws_row(gw, lane, 0)] = c_make(t0x, t0y);
                s.buf[cur ^ 1][kh_ws_row(gw, lane, 1)] = c_make(t1x, t1y);
            }
            __syncthreads();
        } else {
            __syncthreads();  // the A waves have published the reduced odd-term sum
            const cplx o0 = s.odd[kh_ws_row(gw, lane, 0)], o1 = s.odd[kh_ws_row(gw, lane, 1)];
            ev[0].x += o0.x;
            ev[0].y += o0.y;
            ev[1].x += o1.x;
            ev[1].y += o1.y;
            if (cg == 0) {
                s.buf[cur ^ 1][kh_ws_row(gw, lane, 0)] = ev[0];
                s.buf[cur ^ 1][kh_ws_row(gw, lane, 1)] = ev[1];
            }
            __syncthreads();
        }
        cur ^= 1;
    }
}

// one interval of the A waves: odd terms, off the critical path
__device__ __forceinline__ void kh_ws_interval_A(const KhWsLds &s, const cplx (&a)[2][8], int &cur, double fre,
                                                 double fim, double h, int phases, int gw, int lane) {
    const int cg = lane & 7;
    cplx s0 = c_make(0.0, 0.0), s1 = c_make(0.0, 0.0);  // unreduced odd-term sums of this lane's rows
    for (int ph = 0; ph < phases; ++ph) {
        const double hj1 = h * s.inv2[ph].x;
        const cplx c1 = c_make(fre * hj1, fim * hj1);
        cplx xv[8];
        kh_ws_read_x(s.buf[cur], cg, xv);
        cplx y0 = c_make(0.0, 0.0), y1 = c_make(0.0, 0.0);
#ifndef KH_DBG_NO_A_FMA
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c_fma(y0, a[0][j], xv[j]);
            c_fma(y1, a[1][j], xv[j]);
        }
#else
        y0 = xv[0]; y1 = xv[1];
#endif
        c_fma(s0, c1, y0);
        c_fma(s1, c1, y1);
        const bool last = (ph + 1 == phases);
        if (last) {
            const cplx r0 = c_make(sum8(s0.x), sum8(s0.y)), r1 = c_make(sum8(s1.x), sum8(s1.y));
            if (cg == 0) {
                s.odd[kh_ws_row(gw, lane, 0)] = r0;
                s.odd[kh_ws_row(gw, lane, 1)] = r1;
            }
            __syncthreads();
        }
        __syncthreads();
        cur ^= 1;
    }
}

// ---------------------------------------------------------------------------
// plain propagation with storage (backward sweep / iteration-0 forward sweep)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(KH_WS_THREADS)
kh_ws_sweep_store(KhSweepArgs p, const cplx *const *__restrict__ sq, const double *__restrict__ pulses,
                  const cplx *__restrict__ state_in, cplx *__restrict__ store, cplx *__restrict__ state_out,
                  int direction) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KhWsLds s = kh_ws_carve(smem);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, cg = lane & 7;
    if (tid < KH_MAX_DEGREE / 2) s.inv2[tid] = make_double2(1.0 / (2 * tid + 1), 1.0 / (2 * tid + 2));
    if (tid <= KH_MAX_DEGREE) s.deg[tid] = p.deg_theta[tid];
    const bool is_B = wave < 4;
    const int gw = wave & 3, gtid = tid & (KH_WS_GROUP - 1);
    const int N = p.N, nt = p.nt;
    const double f2 = p.fre * p.fre - p.fim * p.fim;
    if (is_B) __builtin_amdgcn_s_setprio(3);
    double matvecs = 0.0;
    for (int k = blockIdx.x; k < p.K; k += gridDim.x) {
        const cplx *const *ops_k = p.ops + (size_t)k * 2;
        const cplx *const *sq_k = sq + (size_t)k * 3;
        __syncthreads();  // previous objective's readers are done with LDS
        cplx m0[2][8], c1t[2][8], c2t[2][8];  // B waves: B, P1, P2;  A waves: A, H1, (unused)
        if (is_B) {
            kh_ws_stage_tile(sq_k[0], N, gw, lane, gtid, s.p0);
            kh_ws_load_tile(sq_k[1], N, gw, lane, c1t);
            kh_ws_load_tile(sq_k[2], N, gw, lane, c2t);
        } else {
            kh_ws_stage_tile(ops_k[0], N, gw, lane, gtid, s.h0);
            kh_ws_load_tile(ops_k[1], N, gw, lane, c1t);
        }
        const double nrm0 = p.op_norms[(size_t)k * 2], nrm1 = p.op_norms[(size_t)k * 2 + 1];
        cplx ev[2];
        ev[0] = ev[1] = c_make(0.0, 0.0);
        if (is_B) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = kh_ws_row(gw, lane, r);
                ev[r] = row < N ? state_in[(size_t)k * N + row] : c_make(0.0, 0.0);
                if (cg == 0) s.buf[0][row] = ev[r];
            }
        }
        int cur = 0;
        __syncthreads();
        if (store != nullptr && wave == 0 && lane < N)
            store[((size_t)k * nt + (direction > 0 ? 0 : nt - 1)) * N + lane] = s.buf[0][lane];

        const int n0 = direction > 0 ? 0 : nt - 2;
        double eps_next = pulses[n0], dt_next = p.dt[n0];
        int m_hint = 12;
#ifdef KH_TIMING
        long long t_build = 0, t_phase = 0;
        const long long t_all0 = clock64();
#endif
        for (int step = 0; step < nt - 1; ++step) {
            const int n = direction > 0 ? step : nt - 2 - step;
            const double eps = eps_next, dt = dt_next;
            if (step + 1 < nt - 1) {
                const int nn = direction > 0 ? n + 1 : n - 1;
                dt_next = p.dt[nn];
                eps_next = pulses[nn];
            }
            int nsub, m;
            kh_degree_lookup((nrm0 + fabs(eps) * nrm1) * dt, s.deg, p.theta_max, p.inv_theta_max, m_hint,
                             &nsub, &m);
            m_hint = m;
            const int phases = (m + 1) >> 1;
            const double h = nsub == 1 ? dt : dt / nsub;
#ifdef KH_TIMING
            const long long tq0 = clock64();
#endif
            if (is_B) {
                const double eps2 = eps * eps;
                kh_ws_build_B(s.p0, gtid, eps, eps2, c1t, c2t, m0);
#ifdef KH_TIMING
                const long long tq1 = clock64();
                t_build += tq1 - tq0;
#endif
                for (int sub = 0; sub < nsub; ++sub) kh_ws_interval_B(s, m0, ev, cur, f2, h, phases, gw, lane);
#ifdef KH_TIMING
                t_phase += clock64() - tq1;
#endif
            } else {
                kh_ws_build_A(s.h0, gtid, eps, c1t, m0);
                for (int sub = 0; sub < nsub; ++sub) kh_ws_interval_A(s, m0, cur, p.fre, p.fim, h, phases, gw, lane);
            }
            matvecs += nsub * phases;  // per wave group; both groups add -> 2 per phase
            if (store != nullptr && wave == 0 && lane < N)
                store[((size_t)k * nt + (direction > 0 ? n + 1 : n)) * N + lane] = s.buf[cur][lane];
        }
        if (state_out != nullptr && wave == 0 && lane < N) state_out[(size_t)k * N + lane] = s.buf[cur][lane];
#ifdef KH_TIMING
        if (tid == 0 && blockIdx.x == 0 && p.stats != nullptr) {
            p.stats[1] = (double)t_build;
            p.stats[2] = (double)t_phase;
            p.stats[3] = (double)(clock64() - t_all0);
        }
#endif
    }
    if ((tid == 0 || tid == KH_WS_GROUP) && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}

// ---------------------------------------------------------------------------
// forward sweep with sequential pulse update (optimize.py:444-508), grid == K
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(KH_WS_THREADS)
kh_ws_forward_update(KhSweepArgs p, const cplx *const *__restrict__ sq, KhUpdateArgs u, KhExchange ex) {
    if (u.n_dev != nullptr) {  // graph-replayed stepwise mode: interval index from device memory
        u.n_begin = *u.n_dev;
        u.n_end = u.n_begin + 1;
        if (u.n_begin >= p.nt - 1) return;
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KhWsLds s = kh_ws_carve(smem);
    double(*red)[4] = (double(*)[4])s.red;  // [parity][A wave]
    double(*D_sh)[2] = (double(*)[2])s.D;   // [parity][value, ok]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, cg = lane & 7;
    if (tid < KH_MAX_DEGREE / 2) s.inv2[tid] = make_double2(1.0 / (2 * tid + 1), 1.0 / (2 * tid + 2));
    if (tid <= KH_MAX_DEGREE) s.deg[tid] = p.deg_theta[tid];
    const bool is_B = wave < 4;
    const int gw = wave & 3, gtid = tid & (KH_WS_GROUP - 1);
    const int N = p.N, nt = p.nt;
    const int k = blockIdx.x;
    const double f2 = p.fre * p.fre - p.fim * p.fim;
    if (is_B) __builtin_amdgcn_s_setprio(3);
    double matvecs = 0.0;

    const cplx *const *ops_k = p.ops + (size_t)k * 2;
    const cplx *const *sq_k = sq + (size_t)k * 3;
    cplx m0[2][8], c1t[2][8], c2t[2][8];
    if (is_B) {
        kh_ws_stage_tile(sq_k[0], N, gw, lane, gtid, s.p0);
        kh_ws_load_tile(sq_k[1], N, gw, lane, c1t);
        kh_ws_load_tile(sq_k[2], N, gw, lane, c2t);
    } else {
        kh_ws_stage_tile(ops_k[0], N, gw, lane, gtid, s.h0);
        kh_ws_load_tile(ops_k[1], N, gw, lane, c1t);  // H1: also dH/d eps (mu.py:123-134)
    }
    const double nrm0 = p.op_norms[(size_t)k * 2], nrm1 = p.op_norms[(size_t)k * 2 + 1];
    const double chi_norm = u.chi_norms[k];

    cplx ev[2];
    ev[0] = ev[1] = c_make(0.0, 0.0);
    if (is_B) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = kh_ws_row(gw, lane, r);
            ev[r] = row < N ? u.phi[(size_t)k * N + row] : c_make(0.0, 0.0);
            if (cg == 0) s.buf[0][row] = ev[r];
        }
    }
    int cur = 0;
    __syncthreads();

    double g_a_loc = 0.0;
    cplx chi[2];
    chi[0] = chi[1] = c_make(0.0, 0.0);
    auto load_chi = [&](int n) {
        if (!is_B) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = kh_ws_row(gw, lane, r);
                chi[r] = row < N ? u.chi_store[((size_t)k * nt + n) * N + row] : c_make(0.0, 0.0);
            }
        }
    };
    // A waves: pieces of Im(mu <chi(t_n) | H1 phi>) -> red[par][gw]; phi in buf[cur]
    auto partial_pieces = [&](int par) {
        if (!is_B) {
            cplx xv[8];
            kh_ws_read_x(s.buf[cur], cg, xv);
            cplx y0 = c_make(0.0, 0.0), y1 = c_make(0.0, 0.0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                c_fma(y0, c1t[0][j], xv[j]);
                c_fma(y1, c1t[1][j], xv[j]);
            }
            y0 = c_make(sum8(y0.x), sum8(y0.y));
            y1 = c_make(sum8(y1.x), sum8(y1.y));
            cplx ov = c_make(0.0, 0.0);
            if (cg == 0) {
                c_fma_conj(ov, chi[0], y0);
                c_fma_conj(ov, chi[1], y1);
            }
            const double v = sum64(u.mu_re * ov.y + u.mu_im * ov.x);
            if (lane == 0) red[par][gw] = v;
            matvecs += 1.0;
        }
    };
    auto partial_total = [&](int par) {
        return chi_norm * (((red[par][0] + red[par][1]) + red[par][2]) + red[par][3]);
    };

    const bool emit_only = (!u.internal_exchange && u.n_begin == u.n_end);
    if ((u.internal_exchange || emit_only) && u.n_begin < nt - 1) {
        load_chi(u.n_begin);
        partial_pieces(u.n_begin & 1);
    }
    __syncthreads();
    if (emit_only) {
        const double part = partial_total(u.n_begin & 1);
        if (tid == 0) u.wg_partial[k] = part;
        return;
    }

    double dt_next = p.dt[u.n_begin], guess_next = u.guess[u.n_begin], shape_next = u.shape[u.n_begin];
    const double lam = u.lambda[0];
    int m_hint = 12;

    for (int n = u.n_begin; n < u.n_end; ++n) {
        const int par = n & 1;
        if (n + 1 < nt - 1) load_chi(n + 1);
        // ---- cross-objective sum (optimize.py:470) ----
        if (u.internal_exchange) {
            if (wave == 0) {
                double part[1] = {partial_total(par)};
                double D[1];
                const bool ok = kh_exchange<1>(ex, n, k, 1, lane, part, D);
                if (lane == 0) {
                    D_sh[par][0] = D[0];
                    D_sh[par][1] = ok ? 1.0 : 0.0;
                }
            }
        } else if (tid == 0) {
            D_sh[par][0] = u.D_in[0];
            D_sh[par][1] = 1.0;
        }
        const double dt = dt_next, guess = guess_next, shape = shape_next;
        if (n + 1 < nt - 1) {
            dt_next = p.dt[n + 1];
            guess_next = u.guess[n + 1];
            shape_next = u.shape[n + 1];
        }
        __syncthreads();
        if (D_sh[par][1] == 0.0) return;
        // ---- pulse update (optimize.py:471-477) ----
        const double d1 = D_sh[par][0];
        const double stepw = shape / lam;
        const double eps = guess + stepw * d1;
        g_a_loc += stepw * (d1 * d1) * dt;
        if (k == 0 && tid == 0) u.opt[n] = eps;
        // ---- propagate over interval n with the updated pulse (optimize.py:479-491) ----
        int nsub, m;
        kh_degree_lookup((nrm0 + fabs(eps) * nrm1) * dt, s.deg, p.theta_max, p.inv_theta_max, m_hint, &nsub,
                         &m);
        m_hint = m;
        const int phases = (m + 1) >> 1;
        const double h = nsub == 1 ? dt : dt / nsub;
        if (is_B) {
            const double eps2 = eps * eps;
kh_ws_build_B(s.p0, gtid, eps, eps2, c1t, c2t, m0);
            for (int sub = 0; sub < nsub; ++sub) kh_ws_interval_B(s, m0, ev, cur, f2, h, phases, gw, lane);
        } else {
kh_ws_build_A(s.h0, gtid, eps, c1t, m0);
            for (int sub = 0; sub < nsub; ++sub) kh_ws_interval_A(s, m0, cur, p.fre, p.fim, h, phases, gw, lane);
        }
        matvecs += nsub * phases;
        if (n + 1 < nt - 1) {
            partial_pieces((n + 1) & 1);
            __syncthreads();
        }
    }
    if (wave == 0 && lane < N) u.phi[(size_t)k * N + lane] = s.buf[cur][lane];
    if (!u.internal_exchange && u.n_end < nt - 1) {
        const double part = partial_total(u.n_end & 1);
        if (tid == 0) u.wg_partial[k] = part;
    }
    if (k == 0 && tid == 0) u.g_a[0] = (u.internal_exchange ? 0.0 : u.g_a[0]) + g_a_loc;
    if ((tid == 0 || tid == KH_WS_GROUP) && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}
