// Register-resident generator for 64 < N <= 128 with per-objective operators (any number of controls).
//
// kh_tile64.h stops at N = 64 and the cooperative kernels need ONE operator list for all objectives; an ensemble of, say,
// N = 80 fell to the generic kernels, which stream every operator from L2 for every term of the series.  Here, as in the
// N <= 64 kernels, the generator A(eps) = H0 + sum_l eps_l H_l lives in registers for the whole sweep and only the term
// vector goes through LDS:
//   * one 512-thread workgroup per objective; lane = (row tid >> 2, column group tid & 3), columns cg + 4 j: N / 4
//     elements per lane (32 at N = 128: 128 VGPRs);
//   * a term = N / 4 broadcast LDS reads (four distinct addresses per wave-level read), N FMAs per lane, a quad sum by DPP,
//     one LDS write per row, one barrier;
//   * per interval A is ADVANCED, A += sum_l (eps_l - eps_l') H_l, from lane-linear copies of the control operators
//     (kh_tn_permute at engine creation: 1 KiB per wave-level load), and restarts from H0 every 64 intervals;
//   * the update sweep's <chi|H_l phi> stream the same copies once more (no room to keep them across the exchange);
//   * ONE control and N <= 96 (H1REG): the control operator stays in registers next to A (2 x 24 elements per lane), so
//     neither the advance nor the partial sums read memory -- with 256 objectives of N = 96 the two streams are 75 MB per
//     interval, i.e. the sweep ran at the memory system's speed, not the CU's.
// Series: the engine's one-term-per-phase ratios (Taylor, or the Chebyshev form for Hermitian / nearly anti-Hermitian
// generators).  Bound: LDS read (512 lanes x N/4 x 16 B per term) and fp64 FMA issue, about equal.
#pragma once

#include "kh_common.h"
#include "kh_generic.h"

#define KH_TN_THREADS 512
#define KH_TN_NMAX 128
#define KH_TN_REFRESH 64

struct KhTnLds {
    cplx (*buf)[KH_TN_NMAX];  // [2][128]
    double *ratio;            // [KH_RATIO_STRIDE]
    double *deg;              // [KH_MAX_DEGREE + 1]
    double *red;              // [8 waves][KH_MAX_L]
    double *D, *ok, *eps, *eps_prev, *g_a;  // [KH_MAX_L] each
};

__host__ __device__ inline size_t kh_tn_lds_bytes() {
    return (size_t)2 * KH_TN_NMAX * sizeof(cplx) + (KH_RATIO_STRIDE + KH_MAX_DEGREE + 1 + 8 * KH_MAX_L + 5 * KH_MAX_L) * sizeof(double) + 64;
}

__device__ __forceinline__ KhTnLds kh_tn_carve(char *smem) {
    KhTnLds s;
    s.buf = (cplx(*)[KH_TN_NMAX])smem;
    s.ratio = (double *)(s.buf + 2);
    s.deg = s.ratio + KH_RATIO_STRIDE;
    s.red = s.deg + KH_MAX_DEGREE + 1;
    s.D = s.red + 8 * KH_MAX_L;
    s.ok = s.D + KH_MAX_L;
    s.eps = s.ok + KH_MAX_L;
    s.eps_prev = s.eps + KH_MAX_L;
    s.g_a = s.eps_prev + KH_MAX_L;
    return s;
}

// row-major N x N -> lane order: out[j * 512 + tid] = in[row tid >> 2][column (tid & 3) + 4 j], j < 32 (zero beyond N)
__global__ void kh_tn_permute(const cplx *__restrict__ in, cplx *__restrict__ out, int N)
#if KH_DEFINES(KH_TU_MAIN)
{
    const int tid = threadIdx.x, row = tid >> 2, cg = tid & 3;
    for (int j = blockIdx.x; j < KH_TN_NMAX / 4; j += gridDim.x) {
        const int col = cg + 4 * j;
        out[(size_t)j * KH_TN_THREADS + tid] = (row < N && col < N) ? in[(size_t)row * N + col] : c_make(0.0, 0.0);
    }
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif

// The lane-order copies are padded to 128 rows; a lane whose row is beyond N (`active` false) holds zeros and does not
// fetch them -- at N = 81 that is 37 % of an operator's bytes, and the kernels with several controls are bound by exactly
// that stream (round 6: N = 100, L = 6: 48 / 88 -> 39 / 68 us per interval; N = 81, L = 2: 15.1 / 27.2 -> 13.8 / 22.7).
template <int EPL>
__device__ __forceinline__ void kh_tn_load(const cplx *__restrict__ tab, int tid, bool active, cplx (&a)[EPL]) {
    const unsigned t = (unsigned)kh_launder(tid);
    if (active) {
#pragma unroll
        for (int j0 = 0; j0 < EPL; j0 += 4) {
#pragma unroll
            for (int j = j0; j < j0 + 4; ++j) a[j] = (tab + (size_t)j * KH_TN_THREADS)[t];
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int j = 0; j < EPL; ++j) a[j] = c_make(0.0, 0.0);
    }
}

// a += w * (operator in lane order)
template <int EPL>
__device__ __forceinline__ void kh_tn_axpy(const cplx *__restrict__ tab, double w, int tid, bool active, cplx (&a)[EPL]) {
    const unsigned t = (unsigned)kh_launder(tid);
    if (!active) return;
#pragma unroll
    for (int j0 = 0; j0 < EPL; j0 += 4) {
        cplx v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (tab + (size_t)(j0 + q) * KH_TN_THREADS)[t];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[j0 + q].x = fma(w, v[q].x, a[j0 + q].x);
            a[j0 + q].y = fma(w, v[q].y, a[j0 + q].y);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// (M x)[row] on the row's four lanes; M's elements in registers
template <int EPL>
__device__ __forceinline__ cplx kh_tn_row(const cplx (&a)[EPL], const cplx *x, int cg) {
    cplx s = c_make(0.0, 0.0);
#pragma unroll
    for (int j0 = 0; j0 < EPL; j0 += 4) {
        cplx v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = x[cg + 4 * (j0 + q)];
#pragma unroll
        for (int q = 0; q < 4; ++q) c_fma(s, a[j0 + q], v[q]);
    }
    return c_make(sum4(s.x), sum4(s.y));
}

// (M x)[row] with M streamed from its lane-order copy
template <int EPL>
__device__ __forceinline__ cplx kh_tn_row_streamed(const cplx *__restrict__ tab, int tid, const cplx *x, int cg) {
    const unsigned t = (unsigned)kh_launder(tid);
    cplx s = c_make(0.0, 0.0);
#pragma unroll
    for (int j0 = 0; j0 < EPL; j0 += 4) {
        cplx m[4], v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            m[q] = (tab + (size_t)(j0 + q) * KH_TN_THREADS)[t];
            v[q] = x[cg + 4 * (j0 + q)];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) c_fma(s, m[q], v[q]);
        __builtin_amdgcn_sched_barrier(0);
    }
    return c_make(sum4(s.x), sum4(s.y));
}

__device__ __forceinline__ void kh_tn_load_ratios(const KhSweepArgs &p, const KhTnLds &s, int m, int tid) {
    __syncthreads();
    if (tid < KH_RATIO_STRIDE) s.ratio[tid] = p.ratios[(size_t)m * KH_RATIO_STRIDE + tid];
    __syncthreads();
}

// state <- exp(f A dt) state, term by term; `state`: this lane's row (the same on the row's four lanes).
// On entry nobody reads the LDS vectors any more; on exit buf[0] or buf[1] holds the last term (not the state).
template <int EPL>
__device__ __forceinline__ int kh_tn_expm_action(const cplx (&a)[EPL], cplx &state, const KhTnLds &s, double fre, double fim,
                                                 double dt, int nsub, int m, int row, int cg, bool active) {
    const double h = dt / nsub;
    const bool writer = active && cg == 0;
    for (int sub = 0; sub < nsub; ++sub) {
        if (writer) s.buf[0][row] = state;
        const double c0 = s.ratio[0];
        state = c_make(c0 * state.x, c0 * state.y);
        __syncthreads();
        int cur = 0;
        for (int j = 1; j <= m; ++j) {
            if (active) {
                const double hj = h * s.ratio[j];
                const cplx t = c_mul(c_make(fre * hj, fim * hj), kh_tn_row(a, s.buf[cur], cg));
                if (writer) s.buf[cur ^ 1][row] = t;
                state.x += t.x;
                state.y += t.y;
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    return nsub * m;
}

// tabs: [K*(1+L)] lane-order copies of this direction's operators (NULL: control absent)
template <int EPL, bool H1REG>
__global__ void __launch_bounds__(KH_TN_THREADS)
kh_tn_sweep_store(KhSweepArgs p, const cplx *const *__restrict__ tabs, const double *__restrict__ pulses,
                  const cplx *__restrict__ state_in, cplx *__restrict__ store, cplx *__restrict__ state_out, int direction) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KhTnLds s = kh_tn_carve(smem);
    const int tid = threadIdx.x, row = tid >> 2, cg = tid & 3, N = p.N, L = p.L, nt = p.nt;
    const bool active = row < N, writer = active && cg == 0;
    if (tid <= KH_MAX_DEGREE) s.deg[tid] = p.q2_theta[tid];
    // columns N .. 127 of the term vectors meet zero matrix elements in kh_tn_row: they must be finite (0 * NaN would
    // poison every row), and dynamic LDS holds whatever the previous kernel left there
    if (tid >= N && tid < KH_TN_NMAX) s.buf[0][tid] = s.buf[1][tid] = c_make(0.0, 0.0);
    double matvecs = 0.0;
    int m_cur = -1;
    for (int k = blockIdx.x; k < p.K; k += gridDim.x) {
        const cplx *const *tab_k = tabs + (size_t)k * (1 + L);
        const double *norms_k = p.op_norms + (size_t)k * (1 + L);
        cplx a[EPL], h1[H1REG ? EPL : 1];
        if constexpr (H1REG) kh_tn_load<EPL>(tab_k[1], tid, active, h1);  // (L == 1, operator present: checked by the host)
        cplx state = active ? state_in[(size_t)k * N + row] : c_make(0.0, 0.0);
        if (store != nullptr && writer) store[((size_t)k * nt + (direction > 0 ? 0 : nt - 1)) * N + row] = state;
        KhDegreeCache dc = {12, 1.0, 0.0};
        for (int step0 = 0; step0 < nt - 1; step0 += KH_TN_REFRESH) {
            __syncthreads();
            kh_tn_load<EPL>(tab_k[0], tid, active, a);  // restart from the drift: rounding of the advances cannot drift
            if (tid < L) s.eps_prev[tid] = 0.0;
            const int step_stop = step0 + KH_TN_REFRESH < nt - 1 ? step0 + KH_TN_REFRESH : nt - 1;
            for (int step = step0; step < step_stop; ++step) {
                const int n = direction > 0 ? step : nt - 2 - step;
                double theta = norms_k[0];
                for (int l = 0; l < L; ++l) {
                    const double v = pulses[(size_t)l * (nt - 1) + n];
                    if (tid == l) s.eps[l] = v;
                    theta += fabs(v) * norms_k[1 + l];
                }
                const double dt = p.dt[n];
                __syncthreads();
                if constexpr (H1REG) {
                    const double d = s.eps[0] - s.eps_prev[0];
#pragma unroll
                    for (int j = 0; j < EPL; ++j) {
                        a[j].x = fma(d, h1[j].x, a[j].x);
                        a[j].y = fma(d, h1[j].y, a[j].y);
                    }
                } else {
                    for (int l = 0; l < L; ++l) {
                        const double e = s.eps[l], d = e - s.eps_prev[l];
                        if (tab_k[1 + l] != nullptr && d != 0.0) kh_tn_axpy<EPL>(tab_k[1 + l], d, tid, active, a);
                    }
                }
                int nsub, m;
                kh_degree_cached(theta * dt, s.deg, p.theta_max, p.inv_theta_max, dc, &nsub, &m);
                if (m != m_cur) {
                    kh_tn_load_ratios(p, s, m, tid);
                    m_cur = m;
                } else {
                    __syncthreads();  // (eps_prev is rewritten below)
                }
                if (tid < L) s.eps_prev[tid] = s.eps[tid];
                matvecs += kh_tn_expm_action<EPL>(a, state, s, p.fre, p.fim, dt, nsub, m, row, cg, active);
                if (store != nullptr && writer) store[((size_t)k * nt + (direction > 0 ? n + 1 : n)) * N + row] = state;
            }
        }
        if (state_out != nullptr && writer) state_out[(size_t)k * N + row] = state;
    }
    if (tid == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}

// forward sweep with sequential pulse update (optimize.py:444-508): ONE launch, grid == K <= #CUs
template <int EPL, bool SO, bool H1REG>
__global__ void __launch_bounds__(KH_TN_THREADS)
kh_tn_forward_update(KhSweepArgs p, const cplx *const *__restrict__ tabs, KhUpdateArgs u, KhExchange ex) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KhTnLds s = kh_tn_carve(smem);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, row = tid >> 2, cg = tid & 3, N = p.N, L = p.L, nt = p.nt;
    const int k = blockIdx.x;
    const bool active = row < N, writer = active && cg == 0;
    const cplx *const *tab_k = tabs + (size_t)k * (1 + L);
    const double *norms_k = p.op_norms + (size_t)k * (1 + L);
    const double chi_norm = u.chi_norms[k];
    if (tid <= KH_MAX_DEGREE) s.deg[tid] = p.q2_theta[tid];
    if (tid < KH_MAX_L) s.g_a[tid] = 0.0;
    // columns N .. 127 of the term vectors meet zero matrix elements in kh_tn_row: they must be finite (0 * NaN would
    // poison every row), and dynamic LDS holds whatever the previous kernel left there
    if (tid >= N && tid < KH_TN_NMAX) s.buf[0][tid] = s.buf[1][tid] = c_make(0.0, 0.0);
    cplx a[EPL], h1[H1REG ? EPL : 1];
    if constexpr (H1REG) kh_tn_load<EPL>(tab_k[1], tid, active, h1);
    cplx state = active ? u.phi[(size_t)k * N + row] : c_make(0.0, 0.0);
    if (SO && writer) u.fw_store[((size_t)k * nt) * N + row] = state;
    double matvecs = 0.0;
    int m_cur = -1;
    KhDegreeCache dc = {12, 1.0, 0.0};

    // pieces of Im(mu <chi(t_n) + 0.5 sigma/||chi|| (phi - phi_prev) | H_l phi(t_n)>) -> red[wave][l]; phi(t_n) = `state`
    auto partial_pieces = [&](int n) {
        if constexpr (!SO && !H1REG) {
            if (u.adj_store != nullptr) {
                // adjoint side (first order; u.adj_store = [L][K][nt][N] H_lk^+ chi_k(t_n), kh_gen_adjoint_side): a control's
                // piece is conj(V_l[row]) phi[row] on the row's writer lane -- no staging of phi, no operator stream
                for (int l = 0; l < L; ++l) {
                    double v = 0.0;
                    if (writer && tab_k[1 + l] != nullptr) {
                        const cplx vl = u.adj_store[(((size_t)l * p.K + k) * nt + n) * N + row];
                        cplx ov = c_make(0.0, 0.0);
                        c_fma_conj(ov, vl, state);
                        v = u.mu_re * ov.y + u.mu_im * ov.x;
                    }
                    v = sum64(v);
                    if (lane == 0) s.red[wave * KH_MAX_L + l] = v;
                }
                matvecs += (double)L;
                __syncthreads();
                return;
            }
        }
        cplx bra = c_make(0.0, 0.0);
        if (writer) {
            bra = u.chi_store[((size_t)k * nt + n) * N + row];
            if constexpr (SO) {
                const cplx prev = u.fw_prev[((size_t)k * nt + n) * N + row];
                const double hs = 0.5 * u.sigma[n] / chi_norm;
                bra.x = fma(hs, state.x - prev.x, bra.x);
                bra.y = fma(hs, state.y - prev.y, bra.y);
            }
            s.buf[0][row] = state;
        }
        __syncthreads();
        for (int l = 0; l < L; ++l) {
            double v = 0.0;
            if (H1REG || tab_k[1 + l] != nullptr) {
                cplx z = c_make(0.0, 0.0);
                if constexpr (H1REG) {
                    if (active) z = kh_tn_row(h1, s.buf[0], cg);
                } else {
                    if (active) z = kh_tn_row_streamed<EPL>(tab_k[1 + l], tid, s.buf[0], cg);
                }
                if (writer) {
                    cplx ov = c_make(0.0, 0.0);
                    c_fma_conj(ov, bra, z);
                    v = u.mu_re * ov.y + u.mu_im * ov.x;
                }
            }
            v = sum64(v);
            if (lane == 0) s.red[wave * KH_MAX_L + l] = v;
        }
        matvecs += (double)L;
        __syncthreads();
    };

    partial_pieces(0);

    for (int nr = 0, n_stop; nr < nt - 1; nr = n_stop) {
        kh_tn_load<EPL>(tab_k[0], tid, active, a);  // restart A = H0
        if (tid < L) s.eps_prev[tid] = 0.0;
        n_stop = (nr / KH_TN_REFRESH + 1) * KH_TN_REFRESH;
        n_stop = n_stop < nt - 1 ? n_stop : nt - 1;
        for (int n = nr; n < n_stop; ++n) {
            // ---- cross-objective sum (optimize.py:470): wave 0 publishes, wave l gathers control l ----
            if (wave == 0) {
                double part[KH_MAX_L];
                for (int l = 0; l < KH_MAX_L; ++l) {
                    double acc = 0.0;
                    if (l < L)
                        for (int w = 0; w < KH_TN_THREADS / 64; ++w) acc += s.red[w * KH_MAX_L + l];
                    part[l] = chi_norm * acc;
                }
                if (ex.G == 1) {
                    if (lane == 0)
                        for (int l = 0; l < L; ++l) {
                            s.D[l] = part[l];
                            s.ok[l] = 1.0;
                        }
                }
                if (ex.G > 1) kh_publish(ex, n & 1, k, L, lane, part, (unsigned)(n + 1));
            }
            if (ex.G > 1 && wave < L) {
                double Dl = 0.0;
                const bool ok = kh_gather_one<KH_GATHER_CHUNKS>(ex, n & 1, L, wave, (unsigned)(n + 1), lane, Dl);
                if (lane == 0) {
                    s.D[wave] = Dl;
                    s.ok[wave] = ok ? 1.0 : 0.0;
                }
            }
            __syncthreads();
            if (ex.world > 1) {  // objectives sharded over GPUs: the GPUs' sums through the peer windows
                if (wave == 0) {
                    double D[KH_MAX_L];
                    bool ok = true;
                    for (int l = 0; l < KH_MAX_L; ++l) {
                        D[l] = l < L ? s.D[l] : 0.0;
                        ok = ok && (l >= L || s.ok[l] != 0.0);
                    }
                    const unsigned int epoch = ex.epoch_base + (unsigned)(n + 1);
                    if (ok) {
                        if (k == 0 && n != ex.fail_at) kh_p2p_publish(ex, n & 1, L, lane, D, epoch);
                        ok = kh_p2p_gather<KH_MAX_L>(ex, n & 1, L, epoch, lane, D);
                    }
                    if (lane == 0)
                        for (int l = 0; l < L; ++l) {
                            s.D[l] = D[l];
                            s.ok[l] = ok ? 1.0 : 0.0;
                        }
                }
                __syncthreads();
            }
            {
                bool all_ok = true;
                for (int l = 0; l < L; ++l) all_ok = all_ok && s.ok[l] != 0.0;
                if (!all_ok) return;
            }
            // ---- pulse update (optimize.py:471-477) ----
            const double dt = p.dt[n];
            double theta = norms_k[0];
            for (int l = 0; l < L; ++l) {
                const double stepw = u.shape[(size_t)l * (nt - 1) + n] / u.lambda[l];
                const double d1 = s.D[l];
                const double eps = u.guess[(size_t)l * (nt - 1) + n] + stepw * d1;
                if (tid == l) {
                    s.eps[l] = eps;
                    s.g_a[l] += stepw * (d1 * d1) * dt;
                    if (k == 0) u.opt[(size_t)l * (nt - 1) + n] = eps;
                }
                theta += fabs(eps) * norms_k[1 + l];
            }
            __syncthreads();
            // ---- propagate over interval n with the updated pulses (optimize.py:479-491) ----
            if constexpr (H1REG) {
                const double d = s.eps[0] - s.eps_prev[0];
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    a[j].x = fma(d, h1[j].x, a[j].x);
                    a[j].y = fma(d, h1[j].y, a[j].y);
                }
            } else {
                for (int l = 0; l < L; ++l) {
                    const double d = s.eps[l] - s.eps_prev[l];
                    if (tab_k[1 + l] != nullptr && d != 0.0) kh_tn_axpy<EPL>(tab_k[1 + l], d, tid, active, a);
                }
            }
            int nsub, m;
            kh_degree_cached(theta * dt, s.deg, p.theta_max, p.inv_theta_max, dc, &nsub, &m);
            if (m != m_cur) {
                kh_tn_load_ratios(p, s, m, tid);
                m_cur = m;
            } else {
                __syncthreads();
            }
            if (tid < L) s.eps_prev[tid] = s.eps[tid];
            matvecs += kh_tn_expm_action<EPL>(a, state, s, p.fre, p.fim, dt, nsub, m, row, cg, active);
            if (SO && writer) u.fw_store[((size_t)k * nt + n + 1) * N + row] = state;
            // ---- partial sums of the next interval ----
            if (n + 1 < nt - 1) partial_pieces(n + 1);
        }
    }
    if (writer) u.phi[(size_t)k * N + row] = state;
    if (k == 0 && tid < L) u.g_a[tid] = s.g_a[tid];
    if (tid == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}
