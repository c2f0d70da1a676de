// Generic sweep kernels: any state dimension N, any number of controls L.
//
// One 256-thread workgroup works on one objective at a time.  Vectors live in
// LDS; operators are streamed from HBM/L2 every matrix-vector product, 16 lanes
// per matrix row (256 contiguous bytes per row segment), four rows per wave.
// This is the correctness path for every shape; N <= 64 problems take the
// register-resident tile kernels of kh_tile64.h instead.
#pragma once

#include "kh_common.h"

#define KH_GEN_THREADS 256

// An operator in compressed-sparse-row form (device pointers); column indices need not be sorted.
struct KhCsr {
    long long nnz;
    const int *indptr;   // [N+1]
    const int *indices;  // [nnz]
    const cplx *data;    // [nnz]
};

struct KhSweepArgs {
    int K, N, L, nt;
    const cplx *const *ops;   // [K*(1+L)] operator pointers for this direction (CSR engines: the data arrays)
    const KhCsr *csr;         // [K*(1+L)] sparse operators of this direction, or NULL: dense row-major ops
    const double *op_norms;   // [K*(1+L)]
    const double *dt;         // [nt-1]
    double fre, fim;          // equation-of-motion factor f (propagators.py:94-99)
    double tol, theta_max, inv_theta_max;
    const double *deg_theta;  // [KH_MAX_DEGREE+1] largest theta per Taylor degree (kh_build_degree_table)
    // register-tile kernels only: their own degree thresholds and series coefficients (kh_common.h, "Series
    // coefficients"): Taylor, or the shorter real-spectrum series when every operator is Hermitian
    const double *q2_theta;   // [KH_MAX_DEGREE+1]
    const double *q2_c0;      // [KH_MAX_DEGREE+1]
    const double *q2_rows;    // [KH_MAX_DEGREE+1][KH_Q2_ROWS][2]   (two-terms-per-phase kernels)
    const double *ratios;     // [KH_MAX_DEGREE+1][KH_RATIO_STRIDE]  (one-term-per-phase kernels)
    double *stats;            // [0] += matvecs issued (per objective, summed)
    // generic kernels, dense operators too large for an LDS-resident generator: one N x N scratch matrix per workgroup
    // (gen_scratch_wgs of them) where A(eps) of the interval is formed once, or NULL: operators re-assembled per term
    cplx *gen_scratch;
    int gen_scratch_wgs;
};

// LDS layout (dynamic): xa[N] xb[N] acc[N] chi[N] + scratch [+ the interval's generator A(eps), N x N, where it fits]
// Everything per control is an LDS array of KH_GEN_MAX_L entries (thread l works on control l): the generic kernels take
// up to KH_GEN_MAX_L = 32 controls (the reference loops over any number of pulses, optimize.py:454-477; the
// register-resident families stop at KH_MAX_L = 8, where per-control values still fit registers).
struct KhGenLds {
    cplx *xa, *xb, *acc, *chi;
    double *red;  // [KH_GEN_THREADS/64 * 2 * KH_GEN_MAX_L] reduction scratch
    double *D;    // [KH_GEN_MAX_L] cross-objective sums of the current interval
    double *eps;  // [KH_GEN_MAX_L] the interval's pulse values
    double *part; // [KH_GEN_MAX_L] this workgroup's partial sums of the coming interval
    double *ga;   // [KH_GEN_MAX_L] running g_a integrals
    int *ok;      // exchange status broadcast
    double *ratio;  // [KH_RATIO_STRIDE] the series' term ratios of the current degree (a global load per term otherwise)
    cplx *A;      // [N][N] A(eps) = H0 + sum_l eps_l H_l of the objective and interval at hand, or NULL (does not fit / CSR)
};

// Dense operators with N <= 96: the generator of an interval is formed ONCE, in LDS (N = 96: 144 KiB), instead of being
// re-assembled from its 1 + L operators in every term of the series -- which re-read (1 + L) N^2 16 bytes from the
// memory side per term and objective (L = 8, N = 64, K = 256: 369 us per interval, all of it that stream).
#define KH_GEN_LDS_A_NMAX 96
__host__ __device__ inline bool kh_gen_lds_A(int N, bool dense) { return dense && N <= KH_GEN_LDS_A_NMAX; }

__host__ __device__ inline size_t kh_gen_lds_base_bytes(int N) {
    return ((size_t)4 * N * sizeof(cplx) + ((KH_GEN_THREADS / 64) * 2 * KH_GEN_MAX_L + 4 * KH_GEN_MAX_L + KH_RATIO_STRIDE) * sizeof(double) + 64 + 15) / 16 * 16;
}

__device__ __forceinline__ KhGenLds kh_gen_carve(char *smem, int N, bool dense) {
    KhGenLds s;
    s.xa = (cplx *)smem;
    s.xb = s.xa + N;
    s.acc = s.xb + N;
    s.chi = s.acc + N;
    s.red = (double *)(s.chi + N);
    s.D = s.red + (KH_GEN_THREADS / 64) * 2 * KH_GEN_MAX_L;
    s.eps = s.D + KH_GEN_MAX_L;
    s.part = s.eps + KH_GEN_MAX_L;
    s.ga = s.part + KH_GEN_MAX_L;
    s.ok = (int *)(s.ga + KH_GEN_MAX_L);
    s.ratio = (double *)(s.ok + 8);
    s.A = kh_gen_lds_A(N, dense) ? (cplx *)(smem + kh_gen_lds_base_bytes(N)) : nullptr;
    return s;
}

__host__ inline size_t kh_gen_lds_bytes(int N, bool dense) {
    return kh_gen_lds_base_bytes(N) + (kh_gen_lds_A(N, dense) ? (size_t)N * N * sizeof(cplx) : 0);
}

// A(eps) = H0 + sum_l eps_l H_l of one objective -> dst (LDS), element by element in the order kh_gen_row_dot assembles
// it (the same fused multiply-adds: bit-identical to the streamed form).  All threads; ends with a barrier.
__device__ __forceinline__ void kh_gen_build_generator(const cplx *const *ops_k, const double *eps, int L, int N, cplx *dst) {
    const int nn = N * N;
    for (int i0 = threadIdx.x; i0 < nn; i0 += 4 * KH_GEN_THREADS) {
        cplx a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q * KH_GEN_THREADS;
            a[q] = i < nn ? ops_k[0][i] : c_make(0.0, 0.0);
        }
        for (int l = 0; l < L; ++l) {
            const cplx *h = ops_k[1 + l];
            if (h == nullptr) continue;
            cplx v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q * KH_GEN_THREADS;
                v[q] = i < nn ? h[i] : c_make(0.0, 0.0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q].x = fma(eps[l], v[q].x, a[q].x);
                a[q].y = fma(eps[l], v[q].y, a[q].y);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q * KH_GEN_THREADS;
            if (i < nn) dst[i] = a[q];
        }
    }
    __syncthreads();
}

// y[row] = sum_c (h0[row][c] + sum_l eps_l h_l[row][c]) * x[c] for the rows this
// 16-lane group owns in this pass.  Returns the row sum in every lane of the
// group (undefined for row >= N).
__device__ __forceinline__ cplx kh_gen_row_dot(const cplx *const *ops_k, const KhCsr *csr_k, const double *eps, int L,
                                               int N, int row, int c16, const cplx *x) {
    cplx sum = c_make(0.0, 0.0);
    if (csr_k != nullptr) {
        // sparse rows (the reference's DensityMatrixODEPropagator regime, propagators.py:162-327: large
        // Liouvillians with a few entries per row): the 16 lanes stride over the row's non-zeros
        if (row < N) {
            for (int o = 0; o <= L; ++o) {
                const KhCsr &a = csr_k[o];
                if (a.data == nullptr) continue;
                const double w = o == 0 ? 1.0 : eps[o - 1];
                cplx part = c_make(0.0, 0.0);
                for (int j = a.indptr[row] + c16; j < a.indptr[row + 1]; j += 16) c_fma(part, a.data[j], x[a.indices[j]]);
                sum.x = fma(w, part.x, sum.x);
                sum.y = fma(w, part.y, sum.y);
            }
        }
    } else if (row < N) {
        const size_t off = (size_t)row * N;
        // four column chunks per trip, all operator loads issued before the FMAs:
        // with one load in flight per lane the row stream is latency-bound (~20 GB/s per CU)
        int c = c16;
        for (; c + 48 < N; c += 64) {
            cplx a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = ops_k[0][off + c + 16 * q];
            for (int l = 0; l < L; ++l) {
                const cplx *h = ops_k[1 + l];
                if (h != nullptr) {
                    cplx v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = h[off + c + 16 * q];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        a[q].x = fma(eps[l], v[q].x, a[q].x);
                        a[q].y = fma(eps[l], v[q].y, a[q].y);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) c_fma(sum, a[q], x[c + 16 * q]);
        }
        for (; c < N; c += 16) {
            cplx a = ops_k[0][off + c];
            for (int l = 0; l < L; ++l) {
                const cplx *h = ops_k[1 + l];
                if (h != nullptr) {
                    const cplx v = h[off + c];
                    a.x = fma(eps[l], v.x, a.x);
                    a.y = fma(eps[l], v.y, a.y);
                }
            }
            c_fma(sum, a, x[c]);
        }
    }
    sum.x = sum16(sum.x);
    sum.y = sum16(sum.y);
    return sum;
}

// Four rows of A x at once for a staged generator (row-major N x N in LDS, N <= KH_GEN_LDS_A_NMAX): rows row0 + grp +
// 16 i, i < 4, of this 16-lane group.  One read of every vector chunk serves the four rows, and the sixteen-odd loads of a
// trip are independent: their latencies overlap instead of adding up pass by pass (2 us -> see docs/HISTORY.md R5.10 per term).
__device__ __forceinline__ void kh_gen_rows4_staged(const cplx *A, int N, int row_first, int c16, const cplx *x, cplx (&out)[4]) {
    cplx sum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sum[i] = c_make(0.0, 0.0);
    // four column chunks per trip: the sixteen matrix loads (global memory when the generator sits in the scratch matrix)
    // and four vector reads of a trip are issued before the first multiply-add -- one round trip per 64 columns instead
    // of one per 16 (N = 160: 18 us per term were ten dependent round trips per four rows)
    for (int c = c16; c < N; c += 64) {
        cplx av[4][4], xv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cq = c + 16 * q;
            const bool in = cq < N;
            xv[q] = in ? x[cq] : c_make(0.0, 0.0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row_first + 16 * i;
                av[i][q] = (in && row < N) ? A[(size_t)row * N + cq] : c_make(0.0, 0.0);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) c_fma(sum[i], av[i][q], xv[q]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = c_make(sum16(sum[i].x), sum16(sum[i].y));
}

// acc <- exp(f * A(eps) * dt) acc, A = H0 + sum eps_l H_l, by s Taylor
// sub-steps of degree m.  All threads of the workgroup call this.
__device__ __forceinline__ int kh_gen_expm_action(const KhSweepArgs &p, const cplx *const *ops_k,
                                                  const KhCsr *csr_k, const double *norms_k, const double *eps,
                                                  double dt, const KhGenLds &s) {
    const int tid = threadIdx.x, N = p.N, L = p.L;
    const int grp = tid >> 4, c16 = tid & 15;  // 16 groups of 16 lanes
    double theta = norms_k[0];
    for (int l = 0; l < L; ++l) theta += fabs(eps[l]) * norms_k[1 + l];
    theta *= dt;
    int nsub, m;
    // thresholds and term ratios of the engine's series (kh_common.h, "Series coefficients": Taylor, or the
    // shorter real-spectrum series when every operator is Hermitian)
    kh_degree_lookup(theta, p.q2_theta, p.theta_max, p.inv_theta_max, 12, &nsub, &m);
    // (every thread of the workgroup passes here with the same m: the previous call's readers of s.ratio are behind a barrier)
    if (tid <= m) s.ratio[tid] = p.ratios[(size_t)m * KH_RATIO_STRIDE + tid];
    __syncthreads();
    const double *ratio = s.ratio;
    const double h = dt / nsub, c0 = ratio[0];
    // the generator of this interval, formed once (kh_gen_build_generator), or the operators streamed per term
    // ... in LDS where it fits, else in this workgroup's scratch matrix (global memory: written and read by this
    // workgroup only, its barrier orders both; the reads then stream ONE matrix per term instead of 1 + L)
    cplx *gen = s.A;
    if (gen == nullptr && csr_k == nullptr && p.gen_scratch != nullptr && (int)blockIdx.x < p.gen_scratch_wgs)
        gen = p.gen_scratch + (size_t)blockIdx.x * N * N;
    const bool staged = gen != nullptr && csr_k == nullptr;
    if (staged) kh_gen_build_generator(ops_k, eps, L, N, gen);
    for (int sub = 0; sub < nsub; ++sub) {
        for (int i = tid; i < N; i += KH_GEN_THREADS) {
            const cplx v = s.acc[i];
            s.xa[i] = v;  // the chain starts from v itself, the sum from T_0 = c_0 v
            s.acc[i] = c_make(c0 * v.x, c0 * v.y);
        }
        __syncthreads();
        cplx *xin = s.xa, *xout = s.xb;
        for (int j = 1; j <= m; ++j) {
            const double hj = h * ratio[j];
            const cplx coef = c_make(p.fre * hj, p.fim * hj);
            if (staged) {
                for (int row0 = 0; row0 < N; row0 += 64) {
                    cplx d[4];
                    kh_gen_rows4_staged(gen, N, row0 + grp, c16, xin, d);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = row0 + grp + 16 * i;
                        if (c16 == 0 && row < N) {
                            const cplx t = c_mul(coef, d[i]);
                            xout[row] = t;
                            s.acc[row].x += t.x;
                            s.acc[row].y += t.y;
                        }
                    }
                }
            } else
            for (int row0 = 0; row0 < N; row0 += 16) {
                const int row = row0 + grp;
                const cplx d = kh_gen_row_dot(ops_k, csr_k, eps, L, N, row, c16, xin);
                if (c16 == 0 && row < N) {
                    const cplx t = c_mul(coef, d);
                    xout[row] = t;
                    s.acc[row].x += t.x;
                    s.acc[row].y += t.y;
                }
            }
            __syncthreads();
            cplx *tmp = xin;
            xin = xout;
            xout = tmp;
        }
    }
    return nsub * m;
}

// ---------------------------------------------------------------------------
// plain propagation with storage (backward sweep, iteration-0 forward sweep)
// ---------------------------------------------------------------------------
// direction +1: n = 0..nt-2, state index n -> n+1 (optimize.py:806-846)
// direction -1: n = nt-2..0, state index n+1 -> n (optimize.py:849-886)
__global__ void __launch_bounds__(KH_GEN_THREADS)
kh_gen_sweep_store(KhSweepArgs p, const double *__restrict__ pulses, const cplx *__restrict__ state_in,
                   cplx *__restrict__ store, cplx *__restrict__ state_out, int direction)
#if KH_DEFINES(KH_TU_GENERIC)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KhGenLds s = kh_gen_carve(smem, p.N, p.csr == nullptr);
    const int tid = threadIdx.x, N = p.N, L = p.L, nt = p.nt;
    double matvecs = 0.0;
    for (int k = blockIdx.x; k < p.K; k += gridDim.x) {
        const cplx *const *ops_k = p.ops + (size_t)k * (1 + L);
        const double *norms_k = p.op_norms + (size_t)k * (1 + L);
        for (int i = tid; i < N; i += KH_GEN_THREADS) s.acc[i] = state_in[(size_t)k * N + i];
        __syncthreads();
        if (store != nullptr) {
            const int idx0 = direction > 0 ? 0 : nt - 1;
            for (int i = tid; i < N; i += KH_GEN_THREADS) store[((size_t)k * nt + idx0) * N + i] = s.acc[i];
        }
        for (int step = 0; step < nt - 1; ++step) {
            const int n = direction > 0 ? step : nt - 2 - step;
            // (the interval's pulse values in LDS, thread l fetching control l: up to KH_GEN_MAX_L of them; the previous
            // step's readers are behind the barrier that ends its last term)
            if (tid < L) s.eps[tid] = pulses[(size_t)tid * (nt - 1) + n];
            __syncthreads();
            matvecs += kh_gen_expm_action(p, ops_k, p.csr ? p.csr + (size_t)k * (1 + L) : nullptr, norms_k, s.eps,
                                          p.dt[n], s);
            if (store != nullptr) {
                const int idx = direction > 0 ? n + 1 : n;
                for (int i = tid; i < N; i += KH_GEN_THREADS) store[((size_t)k * nt + idx) * N + i] = s.acc[i];
            }
        }
        if (state_out != nullptr)
            for (int i = tid; i < N; i += KH_GEN_THREADS) state_out[(size_t)k * N + i] = s.acc[i];
        __syncthreads();
    }
    if (tid == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif

// ---------------------------------------------------------------------------
// forward sweep with sequential pulse update (optimize.py:444-508)
// ---------------------------------------------------------------------------
struct KhUpdateArgs {
    // dH/d eps_l of objective k is the forward control operator ops[k*(1+L)+1+l]
    // itself (mu.py:123-134: linear controls), times:
    double mu_re, mu_im;        // 1 (Hilbert) or i (Liouville), mu.py:130-134
    const cplx *chi_store;      // [K][nt][N]
    const double *chi_norms;    // [K]
    cplx *phi;                  // [K][N] running forward states (engine workspace)
    const double *guess;        // [L][nt-1]
    const double *shape;        // [L][nt-1]
    const double *lambda;       // [L]
    double *opt;                // [L][nt-1]
    double *g_a;                // [L]
    double *wg_partial;         // [G][L] per-workgroup partial sums (stepwise mode)
    const double *D_in;         // [L] all-reduced sums (stepwise mode)
    int n_begin, n_end;         // intervals [n_begin, n_end) are applied; partials of n_end are emitted
    int internal_exchange;      // 1: gather in-kernel (single launch over the grid)
    // second-order update (optimize.py:434-443, 468-469, 492-500); all three NULL for first order
    const cplx *fw_prev;        // [K][nt][N] states propagated under the guess pulses (forward_states0)
    cplx *fw_store;             // [K][nt][N] OUT states propagated under the optimized pulses
    const double *sigma;        // [nt-1] sigma at the interval mid-points
    const int *n_dev;           // stepwise mode under graph replay: the interval index lives in device memory
                                // (read here, incremented by kh_reduce_partials), so every replay is identical
    double adj_sign;            // +1 / -1 if every control operator equals +/- its own adjoint (exactly), else 0:
                                // <chi|H phi> may then be taken as <(sign H) chi|phi> (kernels with ADJ = true)
    const cplx *adj_store;      // [K][nt][N] H_1^+ chi_k(t_n) (cooperative kernels, one control, first order:
                                // kh_coop_adjoint_side); generic kernels, first order, dense operators:
                                // [L][K][nt][N] H_lk^+ chi_k(t_n) (kh_gen_adjoint_side); or NULL
};

// ---------------------------------------------------------------------------
// Update sums on the adjoint side (first order, dense operators; round 6)
// ---------------------------------------------------------------------------
// <chi_k(t_n) | H_lk phi_k(t_n)> = <H_lk^+ chi_k(t_n) | phi_k(t_n)> (optimize.py:454-470), and the left factor does not
// depend on the running forward state: V_lk = H_lk^+ [chi_k(t_0) ... chi_k(t_{nt-1})] for every objective and control
// is a batch of dense (N x N)(N x nt) products IN FRONT of the update sweep, off its serial chain, on the fp64 matrix
// cores -- instead of L streamed matrix-vector products per objective inside every interval of it (L = 8, N = 64,
// K = 256: 41 of the 81 us per interval).  In the sweep a control's partial sum is then one dot product of two vectors.
// Operators: the engine's staged adjoints (what the backward sweep propagates with), any N the generic kernels take.
// v_mfma_f64_16x16x4: A = operator block [row lane & 15][k lane >> 4], B = sixteen co-states [k lane >> 4][vector
// lane & 15], D = [row 4 reg + (lane >> 4)][vector lane & 15]; blockIdx.x: (control l, objective k) as l K + k,
// blockIdx.y: KH_GEN_ADJ_POINTS time points (every wave takes KH_GEN_ADJ_VG groups of sixteen with one fetch of an
// operator block).
typedef double kh_gen_d4 __attribute__((ext_vector_type(4)));
#define KH_GEN_ADJ_THREADS 256  // 4 waves
#define KH_GEN_ADJ_VG 4         // groups of 16 time points per wave: an operator block fetched once serves 64 co-states
#define KH_GEN_ADJ_POINTS (4 * 16 * KH_GEN_ADJ_VG)  // time points per workgroup
__global__ void __launch_bounds__(KH_GEN_ADJ_THREADS)
kh_gen_adjoint_side(const cplx *const *__restrict__ ops_adj /*[K (1 + L)] adjoint operators, row-major N x N*/,
                    const cplx *__restrict__ chi_store /*[K][nt][N]*/, cplx *__restrict__ V /*[L][K][nt][N]*/, int K,
                    int N, int L, int nt)
#if KH_DEFINES(KH_TU_GENERIC)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = lane & 15, kq = lane >> 4;
    const int l = blockIdx.x / K, k = blockIdx.x % K;
    const cplx *op = ops_adj[(size_t)k * (1 + L) + 1 + l];
    if (op == nullptr) return;  // (the control does not occur in this objective: its sums are skipped in the sweep too)
    // this lane's time points (B operand columns): one per vector group
    const int n0 = (blockIdx.y * 4 + wave) * 16 * KH_GEN_ADJ_VG + j;
    const cplx *xk = chi_store + (size_t)k * nt * N;
    cplx *vk = V + ((size_t)l * K + k) * nt * N;
    const int G = (N + 15) / 16;
    for (int g = 0; g < G; ++g) {
        kh_gen_d4 dr[KH_GEN_ADJ_VG], di[KH_GEN_ADJ_VG];
#pragma unroll
        for (int vg = 0; vg < KH_GEN_ADJ_VG; ++vg) dr[vg] = di[vg] = kh_gen_d4{0.0, 0.0, 0.0, 0.0};
        const int arow = 16 * g + j;  // A operand: row lane & 15
        for (int kb = 0; kb < G; ++kb) {
            cplx a[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int kk = 16 * kb + 4 * ks + kq;
                a[ks] = (kk < N && arow < N) ? op[(size_t)arow * N + kk] : c_make(0.0, 0.0);
            }
#pragma unroll
            for (int vg = 0; vg < KH_GEN_ADJ_VG; ++vg) {
                const int n = n0 + 16 * vg;
                cplx b[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int kk = 16 * kb + 4 * ks + kq;
                    b[ks] = (kk < N && n < nt) ? xk[(size_t)n * N + kk] : c_make(0.0, 0.0);
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    dr[vg] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks].x, b[ks].x, dr[vg], 0, 0, 0);
                    dr[vg] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks].y, -b[ks].y, dr[vg], 0, 0, 0);
                    di[vg] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks].x, b[ks].y, di[vg], 0, 0, 0);
                    di[vg] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks].y, b[ks].x, di[vg], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int vg = 0; vg < KH_GEN_ADJ_VG; ++vg) {
            const int n = n0 + 16 * vg;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 16 * g + 4 * reg + kq;
                if (n < nt && row < N) vk[(size_t)n * N + row] = c_make(dr[vg][reg], di[vg][reg]);
            }
        }
    }
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif

// ... first order with the adjoint-side store (u.adj_store, kh_gen_adjoint_side): <H_l^+ chi_k(t_n) | phi_k> is an
// element-wise product of two vectors -- no LDS staging, no matrix; the element loads of a group of eight controls (and
// of phi, which this thread itself wrote) are in flight together, one reduction tree per control.
// `resident`: the workgroup owns ONE objective whose running state sits in s.acc (kh_gen_forward_update keeps it there
// from interval to interval): read from LDS instead of the copy in global memory.
// Both forms ADD this workgroup's pieces to s.part[l] (thread l; zeroed by kh_gen_partials) and end behind a barrier.
__device__ __forceinline__ void kh_gen_partials_adj(const KhSweepArgs &p, const KhUpdateArgs &u, int n, const KhGenLds &s,
                                                    bool resident) {
    const int tid = threadIdx.x, N = p.N, L = p.L, nt = p.nt;
    const int wave = tid >> 6, lane = tid & 63;
    for (int k = blockIdx.x; k < p.K; k += gridDim.x) {
        for (int l0 = 0; l0 < L; l0 += KH_MAX_L) {
            cplx ov[KH_MAX_L];
#pragma unroll
            for (int l = 0; l < KH_MAX_L; ++l) ov[l] = c_make(0.0, 0.0);
            for (int i = tid; i < N; i += KH_GEN_THREADS) {
                const cplx x = resident ? s.acc[i] : u.phi[(size_t)k * N + i];
                cplx v[KH_MAX_L];
#pragma unroll
                for (int l = 0; l < KH_MAX_L; ++l)
                    v[l] = (l0 + l < L && p.ops[(size_t)k * (1 + L) + 1 + l0 + l] != nullptr)
                               ? u.adj_store[(((size_t)(l0 + l) * p.K + k) * nt + n) * N + i]
                               : c_make(0.0, 0.0);
#pragma unroll
                for (int l = 0; l < KH_MAX_L; ++l) c_fma_conj(ov[l], v[l], x);
            }
#pragma unroll
            for (int l = 0; l < KH_MAX_L; ++l) {
                if (l0 + l >= L) break;
                const double re = sum64(ov[l].x), im = sum64(ov[l].y);
                if (lane == 0) {
                    s.red[(wave * KH_GEN_MAX_L + l0 + l) * 2 + 0] = re;
                    s.red[(wave * KH_GEN_MAX_L + l0 + l) * 2 + 1] = im;
                }
            }
        }
        __syncthreads();
        if (tid < L) {
            double re = 0.0, im = 0.0;
            for (int w = 0; w < KH_GEN_THREADS / 64; ++w) {
                re += s.red[(w * KH_GEN_MAX_L + tid) * 2 + 0];
                im += s.red[(w * KH_GEN_MAX_L + tid) * 2 + 1];
            }
            s.part[tid] += u.chi_norms[k] * (u.mu_re * im + u.mu_im * re);  // Im(mu * ov) * norm  (optimize.py:466-467, 473)
        }
        __syncthreads();
    }
}

// Im( mu * norm_k * <chi_k(t_n) | H_l phi_k> ) summed over this workgroup's objectives, for every control l -> s.part[l]
// (LDS; complete behind the function's last barrier).
__device__ __forceinline__ void kh_gen_partials(const KhSweepArgs &p, const KhUpdateArgs &u, int n,
                                                const KhGenLds &s, bool resident = false) {
    const int tid = threadIdx.x, N = p.N, L = p.L, nt = p.nt;
    const int grp = tid >> 4, c16 = tid & 15, wave = tid >> 6, lane = tid & 63;
    if (tid < L) s.part[tid] = 0.0;
    __syncthreads();
    if (u.adj_store != nullptr) {
        kh_gen_partials_adj(p, u, n, s, resident);
        return;
    }
    const double zero_eps[1] = {0.0};
    for (int k = blockIdx.x; k < p.K; k += gridDim.x) {
        for (int i = tid; i < N; i += KH_GEN_THREADS) {
            s.xa[i] = u.phi[(size_t)k * N + i];
            s.chi[i] = u.chi_store[((size_t)k * nt + n) * N + i];
        }
        __syncthreads();
        const double nrm = u.chi_norms[k];
        const double hs = u.sigma != nullptr ? 0.5 * u.sigma[n] / nrm : 0.0;  // folded into the chi term below
        for (int l = 0; l < L; ++l) {
            const cplx *h = p.ops[(size_t)k * (1 + L) + 1 + l];
            cplx ov = c_make(0.0, 0.0);  // <chi + hs * dphi | H_l phi>, partial over this thread's rows
            if (h != nullptr) {
                const cplx *one_op[1] = {h};
                const KhCsr *one_csr = p.csr ? p.csr + (size_t)k * (1 + L) + 1 + l : nullptr;
                for (int row0 = 0; row0 < N; row0 += 16) {
                    const int row = row0 + grp;
                    const cplx d = kh_gen_row_dot(one_op, one_csr, zero_eps, 0, N, row, c16, s.xa);
                    if (c16 == 0 && row < N) {
                        cplx bra = s.chi[row];
                        if (u.sigma != nullptr) {  // second order: + 0.5 sigma <phi - phi_prev | (optimize.py:469)
                            const cplx prev = u.fw_prev[((size_t)k * nt + n) * N + row];
                            bra.x = fma(hs, s.xa[row].x - prev.x, bra.x);
                            bra.y = fma(hs, s.xa[row].y - prev.y, bra.y);
                        }
                        c_fma_conj(ov, bra, d);
                    }
                }
            }
            // workgroup reduction in a fixed order: lanes (sum64) then waves
            const double re = sum64(ov.x), im = sum64(ov.y);
            if (lane == 0) {
                s.red[(wave * KH_GEN_MAX_L + l) * 2 + 0] = re;
                s.red[(wave * KH_GEN_MAX_L + l) * 2 + 1] = im;
            }
        }
        __syncthreads();
        if (tid < L) {
            double re = 0.0, im = 0.0;
            for (int w = 0; w < KH_GEN_THREADS / 64; ++w) {
                re += s.red[(w * KH_GEN_MAX_L + tid) * 2 + 0];
                im += s.red[(w * KH_GEN_MAX_L + tid) * 2 + 1];
            }
            // Im(mu * ov) * norm  (optimize.py:466-467, 473)
            s.part[tid] += nrm * (u.mu_re * im + u.mu_im * re);
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(KH_GEN_THREADS)
kh_gen_forward_update(KhSweepArgs p, KhUpdateArgs u, KhExchange ex)
#if KH_DEFINES(KH_TU_GENERIC)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (u.n_dev != nullptr) {
        u.n_begin = *u.n_dev;
        u.n_end = u.n_begin + 1;
        if (u.n_begin >= p.nt - 1) return;  // replay past the last interval: nothing to do
    }
    const KhGenLds s = kh_gen_carve(smem, p.N, p.csr == nullptr);
    double *D_sh = s.D;  // all LDS in the dynamic region (keeps its base 16-byte aligned)
    int *ok_sh_p = s.ok;
    const int tid = threadIdx.x, N = p.N, L = p.L, nt = p.nt;
    const int wave = tid >> 6, lane = tid & 63;
    double matvecs = 0.0;
    if (tid < L) s.ga[tid] = 0.0;  // (thread l keeps control l's scalars: eps, g_a, partial sum -- all in LDS)

    // One objective per workgroup and the sums on the adjoint side (no stage of the sweep then needs the state in global
    // memory): the running state stays in s.acc from interval to interval -- read once, written back once -- instead of
    // a global round trip in front of and behind every step.
    const bool resident = (int)gridDim.x >= p.K && u.adj_store != nullptr;
    if (resident && (int)blockIdx.x < p.K) {
        for (int i = tid; i < N; i += KH_GEN_THREADS) s.acc[i] = u.phi[(size_t)blockIdx.x * N + i];
        __syncthreads();
    }
    // partial sums of the first interval handled by this launch
    if (u.internal_exchange || u.n_begin == u.n_end) {
        // (stepwise mode enters with the partials of n_begin already reduced in D_in,
        //  except for the begin call n_begin == n_end == 0 which only emits them)
        if (u.n_begin < nt - 1) kh_gen_partials(p, u, u.n_begin, s, resident);
    }
    if (!u.internal_exchange && u.n_begin == u.n_end) {
        if (tid < L) u.wg_partial[(size_t)blockIdx.x * L + tid] = u.n_begin < nt - 1 ? s.part[tid] : 0.0;
        return;
    }

    for (int n = u.n_begin; n < u.n_end; ++n) {
        // ---- cross-objective sum D_l (optimize.py:470) ----
        if (u.internal_exchange) {
            if (wave == 0) {
                bool ok;
                if (L <= KH_MAX_L) {
                    double part[KH_MAX_L], D[KH_MAX_L];
#pragma unroll
                    for (int l = 0; l < KH_MAX_L; ++l) part[l] = l < L ? s.part[l] : 0.0;
                    ok = kh_exchange<KH_MAX_L>(ex, n, blockIdx.x, L, lane, part, D);
                    if (lane == 0)
                        for (int l = 0; l < L; ++l) D_sh[l] = D[l];
                } else {
                    // more controls than a polling round keeps in registers: published at once (two granules per control
                    // and lane: 2 L <= 64), gathered in groups of KH_MAX_L.  One GPU only (the host sees to it: with more
                    // than KH_MAX_L controls a sharded run takes the all-reduce per interval, kh_p2p_create_window).
                    kh_exchange_publish(ex, n, blockIdx.x, L, lane, s.part);
                    ok = true;
                    for (int l0 = 0; l0 < L && ok; l0 += KH_MAX_L) {
                        double D[KH_MAX_L];
                        if (ex.G == 1) {
#pragma unroll
                            for (int l = 0; l < KH_MAX_L; ++l) D[l] = l0 + l < L ? s.part[l0 + l] : 0.0;
                        } else {
                            ok = kh_gather_range<KH_MAX_L, KH_GATHER_CHUNKS>(ex, n & 1, L, l0, (unsigned)(n + 1), lane, D);
                        }
                        if (lane == 0)
                            for (int l = 0; l < KH_MAX_L && l0 + l < L; ++l) D_sh[l0 + l] = D[l];
                    }
                }
                if (lane == 0) *ok_sh_p = ok ? 1 : 0;
            }
            __syncthreads();
            if (!*ok_sh_p) return;
        } else {
            if (tid < L) D_sh[tid] = u.D_in[tid];
            __syncthreads();
        }
        // ---- pulse update (optimize.py:471-477): thread l takes control l ----
        const double dt = p.dt[n];
        if (tid < L) {
            const int l = tid;
            const double S = u.shape[(size_t)l * (nt - 1) + n];
            const double lam = u.lambda[l];
            const double d1 = D_sh[l];
            const double e = u.guess[(size_t)l * (nt - 1) + n] + (S / lam) * d1;
            s.eps[l] = e;
            s.ga[l] += (S / lam) * (d1 * d1) * dt;
            if (blockIdx.x == 0) u.opt[(size_t)l * (nt - 1) + n] = e;
        }
        __syncthreads();
        // ---- propagate every local objective over interval n (optimize.py:479-491) ----
        for (int k = blockIdx.x; k < p.K; k += gridDim.x) {
            const cplx *const *ops_k = p.ops + (size_t)k * (1 + L);
            const double *norms_k = p.op_norms + (size_t)k * (1 + L);
            if (!resident) {
                for (int i = tid; i < N; i += KH_GEN_THREADS) s.acc[i] = u.phi[(size_t)k * N + i];
                __syncthreads();
            }
            if (u.fw_store != nullptr && n == 0)
                for (int i = tid; i < N; i += KH_GEN_THREADS) u.fw_store[((size_t)k * nt) * N + i] = s.acc[i];
            matvecs += kh_gen_expm_action(p, ops_k, p.csr ? p.csr + (size_t)k * (1 + L) : nullptr, norms_k, s.eps, dt,
                                          s);
            if (!resident || n + 1 == u.n_end)
                for (int i = tid; i < N; i += KH_GEN_THREADS) u.phi[(size_t)k * N + i] = s.acc[i];
            if (u.fw_store != nullptr)
                for (int i = tid; i < N; i += KH_GEN_THREADS) u.fw_store[((size_t)k * nt + n + 1) * N + i] = s.acc[i];
            __syncthreads();
        }
        // ---- partial sums of the next interval ----
        if (n + 1 < nt - 1) {
            // phi written above by this same workgroup: visible after the barrier
            kh_gen_partials(p, u, n + 1, s, resident);
            matvecs += (double)L;
        }
    }
    if (!u.internal_exchange && u.n_end < nt - 1 && tid < L) u.wg_partial[(size_t)blockIdx.x * L + tid] = s.part[tid];
    if (blockIdx.x == 0 && tid < L) u.g_a[tid] = (u.internal_exchange ? 0.0 : u.g_a[tid]) + s.ga[tid];
    if (tid == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif

// sum of per-workgroup partials in workgroup order -> out[L]  (stepwise mode)
__global__ void kh_reduce_partials(const double *__restrict__ wg_partial, int G, int L, double *__restrict__ out,
                                   int *n_dev)
#if KH_DEFINES(KH_TU_MAIN)
{
    // one wave; fixed order: lane-strided partial sums in workgroup order, then the sum64 tree
    const int lane = threadIdx.x;
    for (int l = 0; l < L; ++l) {
        double acc = 0.0;
        for (int g = lane; g < G; g += 64) acc += wg_partial[(size_t)g * L + l];
        acc = sum64(acc);
        if (lane == 0) out[l] = acc;
    }
    if (n_dev != nullptr && lane == 0) *n_dev += 1;
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif

// tau_k = <target_k | psi_k>  (second_order.py:69-83); one wave per objective
__global__ void kh_tau_kernel(const cplx *__restrict__ targets, const cplx *__restrict__ psi, cplx *__restrict__ tau,
                              int K, int N)
#if KH_DEFINES(KH_TU_MAIN)
{
    const int k = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (k >= K) return;
    cplx acc = c_make(0.0, 0.0);
    for (int i = lane; i < N; i += 64) c_fma_conj(acc, targets[(size_t)k * N + i], psi[(size_t)k * N + i]);
    const double re = sum64(acc.x), im = sum64(acc.y);
    if (lane == 0) tau[k] = c_make(re, im);
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif

// chi_k(T) of the built-in functionals (functionals.py:177-197, 225-253, 293-317, 389-437) is a
// linear combination of the target and the propagated state with per-objective scalars:
//   v_k = c_k target_k + d_k psi_k(T);   out_k = v_k / ||v_k||_2,  norms[k] = ||v_k||_2
// (optimize.py:407-410; a zero v_k gives NaN like the reference's unguarded division).
// One wave per objective, fixed summation order.
__global__ void kh_chi_kernel(const cplx *__restrict__ targets, const cplx *__restrict__ psi,
                              const cplx *__restrict__ c, const cplx *__restrict__ d, cplx *__restrict__ out,
                              double *__restrict__ norms, int K, int N)
#if KH_DEFINES(KH_TU_MAIN)
{
    const int k = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (k >= K) return;
    const cplx ck = c[k], dk = d[k];
    double acc = 0.0;
    for (int i = lane; i < N; i += 64) {
        cplx v = c_mul(ck, targets[(size_t)k * N + i]);
        c_fma(v, dk, psi[(size_t)k * N + i]);
        acc = fma(v.x, v.x, fma(v.y, v.y, acc));
    }
    const double nrm = sqrt(sum64(acc));
    for (int i = lane; i < N; i += 64) {
        cplx v = c_mul(ck, targets[(size_t)k * N + i]);
        c_fma(v, dk, psi[(size_t)k * N + i]);
        out[(size_t)k * N + i] = c_make(v.x / nrm, v.y / nrm);
    }
    if (lane == 0) norms[k] = nrm;
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif

// Frobenius norms of the operators (fallback when the caller gives no bounds)
__global__ void kh_fro_norms(const cplx *const *ops, int count, int N, double *norms)
#if KH_DEFINES(KH_TU_MAIN)
{
    const int idx = blockIdx.x;
    if (idx >= count) return;
    const cplx *a = ops[idx];
    double acc = 0.0;
    if (a != nullptr)
        for (size_t i = threadIdx.x; i < (size_t)N * N; i += blockDim.x) acc += a[i].x * a[i].x + a[i].y * a[i].y;
    acc = sum64(acc);
    __shared__ double red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (unsigned w = 0; w < blockDim.x / 64; ++w) t += red[w];
        norms[idx] = sqrt(t);
    }
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif

// flags[0] (flags[1]) is raised when some control operator (index i with i % Lp1 != 0) differs from
// plus (minus) its staged adjoint in any bit: decides KhUpdateArgs::adj_sign at engine creation;
// flags[2] when some drift operator (i % Lp1 == 0) differs from its adjoint (flags[0] == flags[2] == 0: every
// generator H0 + sum eps_l H_l is Hermitian, its spectrum real)
__global__ void kh_adjoint_sign_kernel(const cplx *const *__restrict__ ops, const cplx *const *__restrict__ ops_adj,
                                       int nops, int Lp1, int N, int *__restrict__ flags)
#if KH_DEFINES(KH_TU_MAIN)
{
    for (int i = blockIdx.x; i < nops; i += gridDim.x) {
        if (ops[i] == nullptr) continue;
        const bool drift = i % Lp1 == 0;
        const cplx *a = ops[i], *b = ops_adj[i];
        bool plus_bad = false, minus_bad = false;
        for (int idx = threadIdx.x; idx < N * N; idx += blockDim.x) {
            const cplx x = a[idx], y = b[idx];
            plus_bad = plus_bad || x.x != y.x || x.y != y.y;
            minus_bad = minus_bad || x.x != -y.x || x.y != -y.y;
        }
        if (drift) {
            if (plus_bad) flags[2] = 1;
        } else {
            if (plus_bad) flags[0] = 1;
            if (minus_bad) flags[1] = 1;
        }
    }
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif

// max over the drift operators of || (A + sign A^dagger) / 2 ||_F^2  (sign = +1: Hermitian part, -1: anti-Hermitian
// part) -> *out (bits of a non-negative double, atomicMax); one workgroup per operator
__global__ void kh_herm_defect_kernel(const cplx *const *__restrict__ ops, const cplx *const *__restrict__ ops_adj,
                                      int nops, int Lp1, int N, double sign, unsigned long long *__restrict__ out)
#if KH_DEFINES(KH_TU_MAIN)
{
    __shared__ double red[256];
    for (int i = blockIdx.x * Lp1; i < nops; i += gridDim.x * Lp1) {
        double acc = 0.0;
        if (ops[i] != nullptr) {
            const cplx *a = ops[i], *b = ops_adj[i];
            for (int idx = threadIdx.x; idx < N * N; idx += blockDim.x) {
                const double re = 0.5 * (a[idx].x + sign * b[idx].x), im = 0.5 * (a[idx].y + sign * b[idx].y);
                acc += re * re + im * im;
            }
        }
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int st = blockDim.x / 2; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) atomicMax(out, (unsigned long long)__double_as_longlong(red[0]));
        __syncthreads();
    }
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif

// out[c][r] = conj(in[r][c]); 32x32 tiles through LDS
__global__ void kh_adjoint_kernel(const cplx *__restrict__ in, cplx *__restrict__ out, int N)
#if KH_DEFINES(KH_TU_MAIN)
{
    __shared__ cplx tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int r = by + j, c = bx + tx;
        if (r < N && c < N) tile[j][tx] = in[(size_t)r * N + c];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int r = bx + j, c = by + tx;  // out row = in col
        if (r < N && c < N) {
            const cplx v = tile[tx][j];
            out[(size_t)r * N + c] = c_make(v.x, -v.y);
        }
    }
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif
