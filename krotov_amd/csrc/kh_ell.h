// Sparse operators with the matrix in registers: the regime of the reference's DensityMatrixODEPropagator
// (propagators.py:162-327) -- Liouvillians of dimension N = several hundred with a handful of entries per row.
//
// The generic CSR path (kh_generic.h) gives a row to 16 lanes and re-reads the operator arrays from L2 for every term of
// the series: with ~6 entries per row most lanes idle and a term costs ~40 us.  Here
//   * one 512-thread workgroup per objective (two waves per SIMD, 256 VGPRs), one row per lane for N <= 512, two for
//     N <= 1024 (a 1024-thread form with 128 VGPRs was tried first: the compiler serialised every LDS gather and spilled);
//   * the generator A(eps) = A_0 + sum_l eps_l A_l is kept as ONE sparse matrix on the union of the operators' patterns,
//     in padded "ELL" form: lane r holds the E entries of row r -- values (4 VGPRs each) and the byte offsets of their
//     columns in the LDS vector (1 VGPR each) -- in registers for the whole sweep;
//   * per interval only the entries some control touches are re-formed, a_e = v_0e + sum_l eps_l v_le (they are sorted to
//     the front of every row at engine creation: e < Ec; the per-operator values v_le are read lane-linear from L2);
//   * a term of the series is E LDS gathers (ds_read_b128) and 4 E FMAs per lane, one LDS write and one barrier: the
//     vector ping-pongs between two LDS buffers at compile-time offsets, the state's running sum stays in a register.
// The host builds the ELL arrays from the caller's CSR arrays once per distinct operator list and direction
// (krotov_hip.hip: build_ell_host).  Wider rows (E > 32; 16 for N > 512), N > 1024 and the per-interval (sharded / stepwise) launches stay
// with the generic CSR kernels.
#pragma once

#include "kh_common.h"
#include "kh_generic.h"

#define KH_ELL_THREADS 512  // (default workgroup; the kernels are templates on the thread count: 512, 768, 1024)
#define KH_ELL_NMAX 2048  // rows: one per lane up to 512 / 768 / 1024, else 2, 3 or 4 per lane (tid + 512 i) up to 2048
#define KH_ELL_EMAX 32    // widest padded row with one row per lane; with two rows per lane: KH_ELL_EMAX2
#define KH_ELL_EMAX2 16
#define KH_ELL_EMAX4 8    // ... with three or four rows per lane (1024 < N <= 2048)
// rows of the padded pools (= threads x rows per lane of the instantiation that serves N): 512, 768, 1024, 1536, 2048
__host__ __device__ inline int kh_ell_rows(int N) {
    return N <= 512 ? 512 : N <= 768 ? 768 : N <= 1024 ? 1024 : N <= 1536 ? 1536 : 2048;
}
__host__ __device__ inline int kh_ell_emax(int N) { return N <= 512 ? KH_ELL_EMAX : N <= 1024 ? KH_ELL_EMAX2 : KH_ELL_EMAX4; }
#define KH_ELL_THETA_CAP 6.0  // largest ||A dt|| of one sub-step with the Chebyshev-form series (krotov_hip.hip)

// One distinct operator list, one direction: where its arrays start in the engine's two pools (kernel arguments, so the
// loads are global loads; pointers inside a structure read from memory would make them FLAT ones)
struct KhEll {
    long long off_at;   // int pool:  [E][rows] byte offset (column * 16) of every entry's vector element; padding: own row
    long long vals_at;  // cplx pool: [1 + L][E][rows] values of the drift and of every control operator on the union
                        //            pattern (0: absent)
    int E, Ec;          // entries per (padded) row; the first Ec of every row are the ones some control operator touches
    int rows, pad_;     // kh_ell_rows(N): the pools' row count
};

#define KH_ELL_XB_BYTES (KH_ELL_NMAX * (int)sizeof(cplx))  // second vector buffer at a compile-time offset
// STREAMED form (template flag of the kernels below; krotov_hip.hip: e->ell_stream): operators whose rows do not fit
// the registers -- N up to 4096, up to 32 entries per row -- keep NOTHING resident: a term reads the row's offsets and
// values from the pools (lane = row, so every load is coalesced; L2 / Infinity-Cache traffic of 20 bytes per entry and
// term), the interval's values of the control-touched slots are formed once per interval into a per-workgroup scratch
// plane (row-private: the lane that writes an element is the one that reads it).  Eight rows per lane of 512 threads;
// the two vector buffers take 2 x 64 KiB of LDS.  Several times faster than the generic CSR kernels (which it replaces
// for these shapes, and which cannot hold N > 2540 in LDS at all), several times slower than the register form.
#define KH_ELLS_NMAX 4096
#define KH_ELLS_RPL 8
#define KH_ELLS_XB_BYTES (KH_ELLS_NMAX * (int)sizeof(cplx))

struct KhEllLds {
    double *ratio;  // [KH_RATIO_STRIDE] the series' ratios of the current degree
    double *red;    // [16 waves][KH_MAX_L]
    double *D;      // [KH_MAX_L]
    double *ok;     // [KH_MAX_L + 1]
    double *deg;    // [KH_MAX_DEGREE + 1] copy of the degree-threshold table
    double *g_a;    // [KH_MAX_L] running integrals of g_a (thread l)
    double *eps;    // [KH_MAX_L] the interval's pulse values (read by the tile rebuild: a register array indexed by a
                    // run-time control number would be moved through the VGPR index register)
};

__host__ __device__ inline size_t kh_ell_lds_bytes(bool stream = false) {
    return (size_t)2 * (stream ? KH_ELLS_XB_BYTES : KH_ELL_XB_BYTES) +
           (KH_RATIO_STRIDE + 16 * KH_MAX_L + KH_MAX_L + KH_MAX_L + 1 + KH_MAX_DEGREE + 1 + 2 * KH_MAX_L) * sizeof(double) + 64;
}

template <int XB = KH_ELL_XB_BYTES>
__device__ __forceinline__ KhEllLds kh_ell_carve(char *smem) {
    KhEllLds s;
    s.ratio = (double *)(smem + 2 * XB);
    s.red = s.ratio + KH_RATIO_STRIDE;
    s.D = s.red + 16 * KH_MAX_L;
    s.ok = s.D + KH_MAX_L;
    s.deg = s.ok + KH_MAX_L + 1;
    s.eps = s.deg + KH_MAX_DEGREE + 1;
    s.g_a = s.eps + KH_MAX_L;
    return s;
}

// this lane's rows of an ELL structure: offsets and the drift's values (all E entries); row slot i is row tid + T i.
// (Four entries at a time, with a scheduling barrier between the groups: left to itself the compiler issues every load
// of a row at once and needs a second set of registers for the values in flight -- the kernels then spill.)
template <int T, int RPL, int EMAX>
__device__ __forceinline__ void kh_ell_load(const KhEll &el, const int *__restrict__ offs, const cplx *__restrict__ vals,
                                            int tid, cplx (&a)[RPL][EMAX], int (&off)[RPL][EMAX]) {
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        // (uniform base pointers + an unsigned 32-bit lane index: scalar-base addressing, no 64-bit address per entry)
        const unsigned row = (unsigned)(tid + T * i);
#pragma unroll
        for (int e0 = 0; e0 < EMAX; e0 += 4) {
#pragma unroll
            for (int e = e0; e < e0 + 4; ++e) {
                off[i][e] = (int)row * (int)sizeof(cplx);
                a[i][e] = c_make(0.0, 0.0);
            }
            if (e0 < el.E) {  // (E is a multiple of four: build_ell_host)
#pragma unroll
                for (int e = e0; e < e0 + 4; ++e) {
                    const int *po = offs + (el.off_at + (long long)e * el.rows);
                    const cplx *pv = vals + (el.vals_at + (long long)e * el.rows);
                    off[i][e] = po[row];
                    a[i][e] = pv[row];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// a_e = v_0e + sum_l eps_l v_le for the entries the controls touch (e < Ec, a multiple of four)
template <int T, int RPL, int EMAX>
__device__ __forceinline__ void kh_ell_rebuild(const KhEll &el, const cplx *__restrict__ vals, int tid, int L,
                                               const double *eps, cplx (&a)[RPL][EMAX]) {
    const long long plane = (long long)el.E * el.rows;
    // (kh_launder: the addresses below are loop-invariant per entry; visible to the optimiser they are hoisted out of the
    // interval loop and kept in two VGPRs per entry and operator -- registers the matrix needs)
    const int tid_l = kh_launder(tid);
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const unsigned row = (unsigned)(tid_l + T * i);
#pragma unroll
        for (int e0 = 0; e0 < EMAX; e0 += 4) {
            if (e0 < el.Ec) {
#pragma unroll
                for (int e = e0; e < e0 + 4; ++e) a[i][e] = (vals + (el.vals_at + (long long)e * el.rows))[row];
                for (int l = 0; l < L; ++l) {
                    const double w = eps[l];
#pragma unroll
                    for (int e = e0; e < e0 + 4; ++e) {
                        const cplx v = (vals + (el.vals_at + (1 + l) * plane + (long long)e * el.rows))[row];
                        a[i][e].x = fma(w, v.x, a[i][e].x);
                        a[i][e].y = fma(w, v.y, a[i][e].y);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// one row of A x: gathers in groups of four (their latencies overlap; more in flight would need more landing registers)
template <int EMAX>
__device__ __forceinline__ cplx kh_ell_row(const cplx (&a)[EMAX], const int (&off)[EMAX], const char *x) {
    static_assert(EMAX % 4 == 0, "rows are padded to a multiple of four entries");
    cplx s = c_make(0.0, 0.0);
#pragma unroll
    for (int e0 = 0; e0 < EMAX; e0 += 4) {
        cplx v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *(const cplx *)(x + off[e0 + q]);
#pragma unroll
        for (int q = 0; q < 4; ++q) c_fma(s, a[e0 + q], v[q]);
    }
    return s;
}

// (A_l x)_row for one control operator: its values on the first Ec entries of the row, read from L2
template <int EMAX>
__device__ __forceinline__ cplx kh_ell_control_row(const KhEll &el, const cplx *__restrict__ vals, int l, unsigned row,
                                                   const int (&off)[EMAX], const char *x) {
    const long long base = el.vals_at + (long long)(1 + l) * el.E * el.rows;
    row = (unsigned)kh_launder((int)row);
    cplx s = c_make(0.0, 0.0);
#pragma unroll
    for (int e0 = 0; e0 < EMAX; e0 += 4) {
        if (e0 < el.Ec) {
            cplx w[4], v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                w[q] = (vals + (base + (long long)(e0 + q) * el.rows))[row];
                v[q] = *(const cplx *)(x + off[e0 + q]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) c_fma(s, w[q], v[q]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return s;
}

// ---- streamed form: the same three operations with the matrix in the pools ----
// the interval's values of the control-touched slots (e < Ec) of this lane's rows -> the workgroup's scratch plane
template <int T, int RPL>
__device__ __forceinline__ void kh_ells_rebuild(const KhEll &el, const cplx *__restrict__ vals, cplx *__restrict__ scr, int tid,
                                                int L, const double *eps, int N) {
    const long long plane = (long long)el.E * el.rows;
    for (int i = 0; i < RPL; ++i) {
        const int row = tid + T * i;
        if (row >= N) break;
        for (int e0 = 0; e0 < el.Ec; e0 += 4) {
            cplx v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (vals + (el.vals_at + (long long)(e0 + q) * el.rows))[row];
            for (int l = 0; l < L; ++l) {
                const double w = eps[l];
                cplx c[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) c[q] = (vals + (el.vals_at + (1 + l) * plane + (long long)(e0 + q) * el.rows))[row];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q].x = fma(w, c[q].x, v[q].x);
                    v[q].y = fma(w, c[q].y, v[q].y);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) scr[(long long)(e0 + q) * el.rows + row] = v[q];
        }
    }
}

// one row of A x: offsets from the pool, values from the scratch plane (slots the controls touch) or the drift's plane
__device__ __forceinline__ cplx kh_ells_row(const KhEll &el, const int *__restrict__ offs, const cplx *__restrict__ vals,
                                            const cplx *__restrict__ scr, int row, const char *x) {
    cplx s = c_make(0.0, 0.0);
    for (int e0 = 0; e0 < el.E; e0 += 4) {
        const cplx *src = e0 < el.Ec ? scr + (long long)e0 * el.rows : vals + (el.vals_at + (long long)e0 * el.rows);
        const int *po = offs + (el.off_at + (long long)e0 * el.rows);
        int o[4];
        cplx a[4], v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            o[q] = po[(long long)q * el.rows + row];
            a[q] = src[(long long)q * el.rows + row];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *(const cplx *)(x + o[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q) c_fma(s, a[q], v[q]);
    }
    return s;
}

// (A_l x)_row for one control operator
__device__ __forceinline__ cplx kh_ells_control_row(const KhEll &el, const int *__restrict__ offs, const cplx *__restrict__ vals,
                                                    int l, int row, const char *x) {
    const long long base = el.vals_at + (long long)(1 + l) * el.E * el.rows;
    cplx s = c_make(0.0, 0.0);
    for (int e0 = 0; e0 < el.Ec; e0 += 4) {
        int o[4];
        cplx w[4], v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            o[q] = (offs + (el.off_at + (long long)(e0 + q) * el.rows))[row];
            w[q] = (vals + (base + (long long)(e0 + q) * el.rows))[row];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *(const cplx *)(x + o[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q) c_fma(s, w[q], v[q]);
    }
    return s;
}

// kh_ell_expm_action with the matrix streamed
template <int T, int RPL>
__device__ __forceinline__ int kh_ells_expm_action(const KhEll &el, const int *__restrict__ offs, const cplx *__restrict__ vals,
                                                   const cplx *__restrict__ scr, cplx (&state)[RPL], char *smem,
                                                   const double *ratio, double fre, double fim, double dt, int nsub, int m,
                                                   int tid, int N) {
    char *xa = smem, *xb = smem + KH_ELLS_XB_BYTES;
    const double h = dt / nsub;
    auto term = [&](int j, const char *xin, char *xout) {
        const double hj = h * ratio[j];
        const cplx coef = c_make(fre * hj, fim * hj);
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int row = tid + T * i;
            if (row < N) {
                const cplx t = c_mul(coef, kh_ells_row(el, offs, vals, scr, row, xin));
                ((cplx *)xout)[row] = t;
                state[i].x += t.x;
                state[i].y += t.y;
            }
        }
        __syncthreads();
    };
    for (int sub = 0; sub < nsub; ++sub) {
        const double c0 = ratio[0];
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int row = tid + T * i;
            if (row < N) ((cplx *)xa)[row] = state[i];
            state[i] = c_make(c0 * state[i].x, c0 * state[i].y);
        }
        __syncthreads();
        for (int j = 1; j <= m; j += 2) {
            term(j, xa, xb);
            if (j + 1 > m) break;
            term(j + 1, xb, xa);
        }
    }
    return nsub * m;
}

// the series' ratios of degree m -> LDS (workgroup-uniform m; contains barriers)
__device__ __forceinline__ void kh_ell_load_ratios(const KhSweepArgs &p, const KhEllLds &s, int m, int tid) {
    __syncthreads();
    if (tid < KH_RATIO_STRIDE) s.ratio[tid] = p.ratios[(size_t)m * KH_RATIO_STRIDE + tid];
    __syncthreads();
}

// state <- exp(f A dt) state: nsub sub-steps of the engine's degree-m series, term by term (T_j = ratio_j f h A T_{j-1}).
// `state`: this lane's rows, in registers on entry and on exit.  All threads of the workgroup call this.
template <int T, int RPL, int EMAX>
__device__ __forceinline__ int kh_ell_expm_action(const cplx (&a)[RPL][EMAX], const int (&off)[RPL][EMAX], cplx (&state)[RPL],
                                                  char *smem, const double *ratio, double fre, double fim, double dt,
                                                  int nsub, int m, int tid, int N) {
    char *xa = smem, *xb = smem + KH_ELL_XB_BYTES;
    const double h = dt / nsub;
    auto term = [&](int j, const char *xin, char *xout) {
        const double hj = h * ratio[j];
        const cplx coef = c_make(fre * hj, fim * hj);
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int row = tid + T * i;
            if (row < N) {
                const cplx t = c_mul(coef, kh_ell_row(a[i], off[i], xin));
                ((cplx *)xout)[row] = t;
                state[i].x += t.x;
                state[i].y += t.y;
            }
        }
        __syncthreads();
    };
    for (int sub = 0; sub < nsub; ++sub) {
        const double c0 = ratio[0];
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int row = tid + T * i;
            if (row < N) ((cplx *)xa)[row] = state[i];  // the chain starts from v itself, the sum from T_0 = c_0 v
            state[i] = c_make(c0 * state[i].x, c0 * state[i].y);
        }
        __syncthreads();
        for (int j = 1; j <= m; j += 2) {
            term(j, xa, xb);
            if (j + 1 > m) break;
            term(j + 1, xb, xa);
        }
    }
    return nsub * m;
}

// ---------------------------------------------------------------------------
// plain propagation with storage (backward sweep / iteration-0 forward sweep)
// ---------------------------------------------------------------------------
// STREAM: the streamed form (see KH_ELLS_NMAX); scratch: [gridDim.x][Ec_max * rows] elements, scratch_stride per workgroup
template <int T, int RPL, int EMAX, bool STREAM = false>
__global__ void __launch_bounds__(T)
kh_ell_sweep_store(KhSweepArgs p, const KhEll *__restrict__ ells, const int *__restrict__ offs, const cplx *__restrict__ vals,
                   const double *__restrict__ pulses, const cplx *__restrict__ state_in, cplx *__restrict__ store,
                   cplx *__restrict__ state_out, int direction, cplx *__restrict__ scratch, long long scratch_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KhEllLds s = kh_ell_carve<STREAM ? KH_ELLS_XB_BYTES : KH_ELL_XB_BYTES>(smem);
    cplx *scr = STREAM ? scratch + (long long)blockIdx.x * scratch_stride : nullptr;
    const int tid = threadIdx.x, N = p.N, L = p.L, nt = p.nt;
    double matvecs = 0.0;
    int m_cur = -1;
    if (tid <= KH_MAX_DEGREE) s.deg[tid] = p.q2_theta[tid];
    __syncthreads();
    for (int k = blockIdx.x; k < p.K; k += gridDim.x) {
        const KhEll el = ells[k];
        const double *norms_k = p.op_norms + (size_t)k * (1 + L);
        cplx a[STREAM ? 1 : RPL][EMAX];
        int off[STREAM ? 1 : RPL][EMAX];
        if constexpr (!STREAM) kh_ell_load<T, RPL, EMAX>(el, offs, vals, tid, a, off);
        cplx state[RPL];
        auto put = [&](cplx *dst) {
#pragma unroll
            for (int i = 0; i < RPL; ++i)
                if (tid + T * i < N) dst[tid + T * i] = state[i];
        };
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int row = tid + T * i;
            state[i] = row < N ? state_in[(size_t)k * N + row] : c_make(0.0, 0.0);
        }
        if (store != nullptr) put(store + ((size_t)k * nt + (direction > 0 ? 0 : nt - 1)) * N);
        KhDegreeCache dc = {12, 1.0, 0.0};
        for (int step = 0; step < nt - 1; ++step) {
            const int n = direction > 0 ? step : nt - 2 - step;
            double theta = norms_k[0];
            for (int l = 0; l < L; ++l) {
                const double v = pulses[(size_t)l * (nt - 1) + n];
                if (tid == l) s.eps[l] = v;
                theta += fabs(v) * norms_k[1 + l];
            }
            const double dt = p.dt[n];
            __syncthreads();  // (s.eps; also: the previous interval's last term has been read by everybody)
            if constexpr (STREAM)
                kh_ells_rebuild<T, RPL>(el, vals, scr, tid, L, s.eps, N);
            else
                kh_ell_rebuild<T, RPL, EMAX>(el, vals, tid, L, s.eps, a);
            int nsub, m;
            kh_degree_cached(theta * dt, s.deg, p.theta_max, p.inv_theta_max, dc, &nsub, &m);
            if (m != m_cur) {
                kh_ell_load_ratios(p, s, m, tid);
                m_cur = m;
            }
            if constexpr (STREAM)
                matvecs += kh_ells_expm_action<T, RPL>(el, offs, vals, scr, state, smem, s.ratio, p.fre, p.fim, dt, nsub, m, tid, N);
            else
                matvecs += kh_ell_expm_action<T, RPL, EMAX>(a, off, state, smem, s.ratio, p.fre, p.fim, dt, nsub, m, tid, N);
            if (store != nullptr) put(store + ((size_t)k * nt + (direction > 0 ? n + 1 : n)) * N);
        }
        if (state_out != nullptr) put(state_out + (size_t)k * N);
    }
    if (tid == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}

// ---------------------------------------------------------------------------
// forward sweep with sequential pulse update (optimize.py:444-508): ONE launch, grid == K, sums exchanged in-kernel
// ---------------------------------------------------------------------------
template <int T, int RPL, int EMAX, bool SO, bool STREAM = false>
__global__ void __launch_bounds__(T)
kh_ell_forward_update(KhSweepArgs p, const KhEll *__restrict__ ells, const int *__restrict__ offs,
                      const cplx *__restrict__ vals, KhUpdateArgs u, KhExchange ex, cplx *__restrict__ scratch,
                      long long scratch_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KhEllLds s = kh_ell_carve<STREAM ? KH_ELLS_XB_BYTES : KH_ELL_XB_BYTES>(smem);
    cplx *scr = STREAM ? scratch + (long long)blockIdx.x * scratch_stride : nullptr;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, N = p.N, L = p.L, nt = p.nt;
    const int k = blockIdx.x;
    const KhEll el = ells[k];
    const double *norms_k = p.op_norms + (size_t)k * (1 + L);
    const double chi_norm = u.chi_norms[k];
    cplx a[STREAM ? 1 : RPL][EMAX];
    int off[STREAM ? 1 : RPL][EMAX];
    if constexpr (!STREAM) kh_ell_load<T, RPL, EMAX>(el, offs, vals, tid, a, off);
    cplx state[RPL];
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int row = tid + T * i;
        state[i] = row < N ? u.phi[(size_t)k * N + row] : c_make(0.0, 0.0);
        if (SO && row < N) u.fw_store[((size_t)k * nt) * N + row] = state[i];
    }
    double matvecs = 0.0;
    if (tid < KH_MAX_L) s.g_a[tid] = 0.0;
    int m_cur = -1;
    KhDegreeCache dc = {12, 1.0, 0.0};
    if (tid <= KH_MAX_DEGREE) s.deg[tid] = p.q2_theta[tid];  // (visible after the first barrier below)

    // the workgroup's pieces of Im(mu <chi(t_n) + 0.5 sigma/||chi|| (phi - phi_prev) | A_l phi(t_n)>) -> red[wave][l];
    // phi(t_n) = `state`, which goes to the first LDS buffer for the gathers
    auto partial_pieces = [&](int n) {
        cplx bra[RPL];
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int row = tid + T * i;
            bra[i] = c_make(0.0, 0.0);
            if (row < N) {
                bra[i] = u.chi_store[((size_t)k * nt + n) * N + row];
                if constexpr (SO) {
                    const cplx prev = u.fw_prev[((size_t)k * nt + n) * N + row];
                    const double hs = 0.5 * u.sigma[n] / chi_norm;
                    bra[i].x = fma(hs, state[i].x - prev.x, bra[i].x);
                    bra[i].y = fma(hs, state[i].y - prev.y, bra[i].y);
                }
                ((cplx *)smem)[row] = state[i];
            }
        }
        __syncthreads();
        for (int l = 0; l < L; ++l) {
            double v = 0.0;
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const int row = tid + T * i;
                if (row < N) {
                    cplx z;
                    if constexpr (STREAM)
                        z = kh_ells_control_row(el, offs, vals, l, row, smem);
                    else
                        z = kh_ell_control_row<EMAX>(el, vals, l, (unsigned)row, off[i], smem);
                    cplx ov = c_make(0.0, 0.0);
                    c_fma_conj(ov, bra[i], z);
                    v += u.mu_re * ov.y + u.mu_im * ov.x;  // Im(mu <bra|A_l phi>): one real combination
                }
            }
            v = sum64(v);
            if (lane == 0) s.red[wave * KH_MAX_L + l] = v;
        }
        matvecs += (double)L;
        __syncthreads();
    };

    partial_pieces(0);

    for (int n = 0; n < nt - 1; ++n) {
        // ---- cross-objective sum (optimize.py:470): wave 0 publishes, wave l gathers control l ----
        if (wave == 0) {
            double part[KH_MAX_L];
            for (int l = 0; l < KH_MAX_L; ++l) {
                double acc = 0.0;
                if (l < L)
                    for (int w = 0; w < T / 64; ++w) acc += s.red[w * KH_MAX_L + l];
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
        if constexpr (STREAM)
            kh_ells_rebuild<T, RPL>(el, vals, scr, tid, L, s.eps, N);
        else
            kh_ell_rebuild<T, RPL, EMAX>(el, vals, tid, L, s.eps, a);
        int nsub, m;
        kh_degree_cached(theta * dt, s.deg, p.theta_max, p.inv_theta_max, dc, &nsub, &m);
        if (m != m_cur) {
            kh_ell_load_ratios(p, s, m, tid);
            m_cur = m;
        }
        if constexpr (STREAM)
            matvecs += kh_ells_expm_action<T, RPL>(el, offs, vals, scr, state, smem, s.ratio, p.fre, p.fim, dt, nsub, m, tid, N);
        else
            matvecs += kh_ell_expm_action<T, RPL, EMAX>(a, off, state, smem, s.ratio, p.fre, p.fim, dt, nsub, m, tid, N);
        if constexpr (SO) {
#pragma unroll
            for (int i = 0; i < RPL; ++i)
                if (tid + T * i < N) u.fw_store[((size_t)k * nt + n + 1) * N + tid + T * i] = state[i];
        }
        // ---- partial sums of the next interval ----
        if (n + 1 < nt - 1) partial_pieces(n + 1);
    }
#pragma unroll
    for (int i = 0; i < RPL; ++i)
        if (tid + T * i < N) u.phi[(size_t)k * N + tid + T * i] = state[i];
    if (k == 0 && tid < L) u.g_a[tid] = s.g_a[tid];
    if (tid == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}
