// Matrix-core update sweep with the cross-objective exchange hidden behind half of the series:
// N <= 64, one control, first order, every operator Hermitian (Hilbert space, f = -+i).
//
// Per interval the chain  eps[n] -> phi(t_{n+1}) -> partial sum of interval n+1 -> exchange -> eps[n+1]
// is serial (optimize.py:449-501).  kh_tile64q2.h runs all P = ceil(m/2) dependent products of the series
// before it can publish its partial sum  d = Im <w | phi(t_{n+1})>,  w = (+-H1) chi(t_{n+1}),  and then idles
// through the exchange.  Here the sum is taken "in the middle" of the polynomial instead.  With B = A^2
// Hermitian and the even terms  T_2p = gamma_2p (f h)^2p B^p phi:
//     <w | T_2(Pf+q)> = gamma_2(Pf+q) / (gamma_2Pf gamma_2q) <V_q | T_2Pf>,   V_q = gamma_2q (f h)^2q B^q w,
// so after only Pf = ceil(P/2) products of TWO chains (T on phi, V on w) every inner product the sum needs is
// available:
//     <w | even part> = <w | sum_{p<=Pf} T_2p> + <w_e | T_2Pf>,     w_e  = sum_q ge_q V_q
//     <w | odd part>  = f [ <A w | s_lo> + <A w_hi | T_2Pf> ],      w_hi = sum_q h go_q V_q,
// (s_lo: the part of the odd-term source s = sum_p h r1_p T_2p known by then; A Hermitian).  The partial sum is
// published after Pf + 1 products; the remaining P - Pf products of the phi chain, the odd-term product A s
// and the next w run in the shadow of the exchange.  Same polynomial, same result to rounding.
//
// Two chains sharing one operator are four real right-hand sides [T_re, T_im, V_re, V_im] -- exactly the four
// columns of v_mfma_f64_4x4x4_4b, so the products run on the fp64 matrix cores at full column use, read the
// vector from LDS once per wave (2 ds_read_b128 instead of 8 per chain) and need no DPP row sums:
//   wave w owns rows 8w..8w+7 as two 4-row blocks I = 2w + io;  lane = 16 hi + 4 b + lo
//   operator tile (A operand):  t[io][j] = M[4 I + lo][16 j + 4 b + hi]            (j = 0..3, re and im)
//   vector (B operand), reg j:  X[16 j + 4 b + hi][lo]       columns lo = 0,1: first vector, 2,3: second
//   result (D):                 Y[4 I + hi][lo], partial over the column blocks b -> two row rotations
// Complex arithmetic: D1 = M_re X, D2 = M_im X,  Y[:, n] = D1[:, n] -+ D2[:, n ^ 1].
//
// Registers: H1 (control operator: w = +-H1 chi), A = H0 + eps H1 and B = A^2 are resident in operand order
// (3 x 32 VGPRs).  P1 = H0 H1 + H1 H0 and P2 = H1 H1 live in LDS (lane-linear, 2 x 64 KiB) and are streamed once
// per interval for the INCREMENTAL update  A += (eps - eps') H1,  B += (eps - eps') P1 + (eps^2 - eps'^2) P2;
// every KH_MM_REFRESH intervals A and B are rebuilt from H0 and P0 = H0 H0 in global memory (L2), so rounding
// cannot drift (64 updates: a few ulp).
#pragma once

#include "kh_common.h"
#include "kh_generic.h"
#include "kh_tile64.h"
#include "kh_tile64q2.h"

#include <type_traits>
#include <utility>

#define KH_MM_XLEN 256                      // doubles of one vector pair in operand order
#define KH_MM_TAB_ROWS (KH_Q2_ROWS + 1)     // per product t: {scale, c1, c2, -} per lane kind; last row: start values
#define KH_MM_TAB_STRIDE (KH_MM_TAB_ROWS * 2 * 4)  // doubles per degree
#define KH_MM_REFRESH 64                    // intervals between two restarts of the incremental A, B from H0, P0

#ifdef KH_TIMING
#define KH_MM_TRACE(i) do { if (trace_on) p.stats[4 + (i)] = (double)clock64(); } while (0)
#else
#define KH_MM_TRACE(i) do { } while (0)
#endif
#define KH_DPP_REV4 0x1B    // quad_perm [3,2,1,0]

// Host: the per-degree coefficient rows of the two-chain form, from the series' c0 / rows tables
// (gamma_2p = ge[p], gamma_2p+1 = go[p]; P = ceil(m/2) products, the first Pf = ceil(P/2) before the sum):
//   kind 0 (lanes of the phi chain), product t:  {r2_t, 1, r1_{t+1} if t+1 <= P-1}
//       T_{2t+2} = r2_t (fh)^2 B T_2t;  E += 1 T;  s += h r1 T
//   kind 1 (lanes of the w chain), product t:    {kappa_t, a_q, b_q},  q = Pf-1-t,  kappa_0 = a_Pf, else 1
//       Y <- kappa_t (fh)^2 B Y + a_q w + h b_q conj(f) w',
//       a_q = gamma_2(Pf+q) / gamma_2Pf (1 <= q <= P-Pf),  b_q = gamma_2(Pf+q)+1 / gamma_2Pf (1 <= q <= P-1-Pf)
//   row KH_Q2_ROWS: start values {c0, r1_0} / {0, 0}
__host__ inline void kh_build_mm_tab(const double *c0, const double *rows, double *tab /*[KH_MAX_DEGREE+1][KH_MM_TAB_STRIDE]*/) {
    for (int m = 0; m <= KH_MAX_DEGREE; ++m) {
        double *T = tab + (size_t)m * KH_MM_TAB_STRIDE;
        for (int i = 0; i < KH_MM_TAB_STRIDE; ++i) T[i] = 0.0;
        if (m < 1) continue;
        const int P = (m + 1) >> 1, Pf = (P + 1) >> 1;
        const double *r = rows + (size_t)m * KH_Q2_ROWS * 2;  // r[2p] = r1_p, r[2p+1] = r2_p
        long double ge[KH_Q2_ROWS + 2], go[KH_Q2_ROWS + 2];
        ge[0] = c0[m];
        go[0] = r[0];
        for (int p = 0; p < P && p < KH_Q2_ROWS; ++p) {
            ge[p + 1] = p == 0 ? (long double)r[1] : (long double)r[2 * p + 1] * ge[p];
            if (p + 1 < KH_Q2_ROWS) go[p + 1] = (long double)r[2 * (p + 1)] * ge[p + 1];
        }
        auto a_ = [&](int q) { return (q >= 1 && q <= P - Pf) ? (double)(ge[Pf + q] / ge[Pf]) : 0.0; };
        auto b_ = [&](int q) { return (q >= 1 && q <= P - 1 - Pf) ? (double)(go[Pf + q] / ge[Pf]) : 0.0; };
        for (int t = 0; t < P && t < KH_Q2_ROWS; ++t) {
            double *F = T + (size_t)(t * 2 + 0) * 4, *W = T + (size_t)(t * 2 + 1) * 4;
            F[0] = r[2 * t + 1];
            F[1] = 1.0;
            F[2] = (t + 1 <= P - 1 && t + 1 < KH_Q2_ROWS) ? r[2 * (t + 1)] : 0.0;
            const int q = Pf - 1 - t;
            W[0] = t == 0 ? a_(Pf) : 1.0;
            W[1] = a_(q);
            W[2] = b_(q);
        }
        double *S = T + (size_t)(KH_Q2_ROWS * 2) * 4;
        S[0] = c0[m];
        S[1] = r[0];
    }
}

struct KhMmLds {
    cplx *p1, *p2;  // [8][512] lane-linear operator tiles
    // seven vector pairs in operand order, KH_MM_XLEN doubles each, addressed as x0 + slot * KH_MM_XLEN (no
    // pointer table: a dynamically indexed one lands in scratch and turns the LDS accesses into FLAT ones):
    //   slot 0: [phi(t_n) | w(t_{n+1})], input of the interval's first product;  1, 2: chain ping-pong;
    //   3 + par (xs): [s | w(t_{n+2})], input of the A product;  5 + par (xh): [chi(t_{n+3}) | w(t_{n+2})], of the H1 product
    double *x0;
    double *red;    // [2][8]
    double *D;      // [2][2]
    double *gsum;   // [8][2]: the waves' partial sums of the exchange and their success flags
    double *tab;    // [KH_MM_TAB_STRIDE] rows of the current degree
    double *deg;    // [KH_MAX_DEGREE+1]
};
#define KH_MM_SLOT_XS 3
#define KH_MM_SLOT_XH 5

__host__ __device__ inline size_t kh_mm_lds_bytes() {
    return (size_t)2 * KH_Q2_TILE_ELEMS * sizeof(cplx) +
           (7 * KH_MM_XLEN + 16 + 4 + 16 + KH_MM_TAB_STRIDE + KH_MAX_DEGREE + 2) * sizeof(double);
}

__device__ __forceinline__ KhMmLds kh_mm_carve(char *smem) {
    KhMmLds s;
    s.p1 = (cplx *)smem;
    s.p2 = s.p1 + KH_Q2_TILE_ELEMS;
    s.x0 = (double *)(s.p2 + KH_Q2_TILE_ELEMS);
    s.red = s.x0 + 7 * KH_MM_XLEN;
    s.D = s.red + 16;
    s.gsum = s.D + 4;
    s.tab = s.gsum + 16;
    s.deg = s.tab + KH_MM_TAB_STRIDE;
    return s;
}

// f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N-1>): a loop whose index is a constant
// expression inside the body (the unrolled interval below selects its work with `if constexpr`)
template <class F, int... I>
__device__ __forceinline__ void kh_static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void kh_static_for(F &&f) {
    kh_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// NIO = 4-row blocks per wave: 2 -> 8 waves (two per SIMD, 256 registers each), 4 -> 4 waves (one per SIMD, 512)
template <int NIO>
struct KhMmTile {
    double re[NIO][4], im[NIO][4];
};
template <int NIO>
struct KhMm {
    static constexpr int WAVES = 16 / NIO;
    static constexpr int THREADS = 64 * WAVES;
};

// element (io, j) of this lane in A-operand order: row 4 (NIO wave + io) + lo, column 16 j + 4 b + hi
template <int NIO>
__device__ __forceinline__ cplx kh_mm_elem(const cplx *op, int N, int wave, int lane, int io, int j) {
    const int hi = lane >> 4, b = (lane >> 2) & 3, lo = lane & 3;
    const int row = 4 * NIO * wave + 4 * io + lo, col = 16 * j + 4 * b + hi;
    return (op != nullptr && row < N && col < N) ? op[(size_t)row * N + col] : c_make(0.0, 0.0);
}

template <int NIO>
__device__ __forceinline__ void kh_mm_load_tile(const cplx *op, int N, int wave, int lane, KhMmTile<NIO> &t) {
#pragma unroll
    for (int io = 0; io < NIO; ++io)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const cplx v = kh_mm_elem<NIO>(op, N, wave, lane, io, j);
            t.re[io][j] = v.x;
            t.im[io][j] = v.y;
            __builtin_amdgcn_sched_barrier(0);  // (set-up code: keep its register footprint small)
        }
}

template <int NIO>
__device__ __forceinline__ void kh_mm_stage_tile(const cplx *op, int N, int wave, int lane, int tid, cplx *dst) {
#pragma unroll
    for (int io = 0; io < NIO; ++io)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dst[(io * 4 + j) * KhMm<NIO>::THREADS + tid] = kh_mm_elem<NIO>(op, N, wave, lane, io, j);
            __builtin_amdgcn_sched_barrier(0);  // (set-up code: keep its register footprint small)
        }
}

// the vector pair in operand order: this lane's four k-blocks
__device__ __forceinline__ void kh_mm_readx(const double *xb, int lane, double (&x)[4]) {
    const double2 a = *(const double2 *)(xb + lane * 2);
    const double2 c = *(const double2 *)(xb + 128 + lane * 2);
    x[0] = a.x;
    x[1] = a.y;
    x[2] = c.x;
    x[3] = c.y;
}

// Y = M X for the wave's rows: y[io] = Y[4 (NIO wave + io) + hi][lo], the same in all four lanes b
template <int NIO>
__device__ __forceinline__ void kh_mm_pass(const KhMmTile<NIO> &t, const double (&x)[4], double sgn, double (&y)[NIO]) {
    double d1[NIO], d2[NIO];
#pragma unroll
    for (int io = 0; io < NIO; ++io) d1[io] = d2[io] = 0.0;
#ifdef KH_MM_X_NOMFMA  // (timing experiment: results are wrong)
#pragma unroll
    for (int io = 0; io < NIO; ++io) {
        d1[io] = t.re[io][0] * x[io];
        d2[io] = t.im[io][1] * x[io + 2];
    }
#else
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int io = 0; io < NIO; ++io) {
            d1[io] = __builtin_amdgcn_mfma_f64_4x4x4f64(t.re[io][j], x[j], d1[io], 0, 0, 0);
            d2[io] = __builtin_amdgcn_mfma_f64_4x4x4f64(t.im[io][j], x[j], d2[io], 0, 0, 0);
        }
    }
#endif
#pragma unroll
    for (int io = 0; io < NIO; ++io) {
#ifdef KH_MM_X_NOTAIL  // (timing experiment: results are wrong)
        y[io] = fma(sgn, d2[io], d1[io]);
#else
        double a = fma(sgn, dpp_move<KH_DPP_XOR1>(d2[io]), d1[io]);
        a += dpp_move<KH_DPP_ROR8>(a);
        a += dpp_move<KH_DPP_ROR4>(a);
        y[io] = a;
#endif
    }
}

// sum over the wave of a value that is zero except on lanes with b == 0 and lo < 2 (valid in every lane)
__device__ __forceinline__ double kh_mm_wave_sum(double v) {
    const double r = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);  // lanes 16 lo' + ...: sum over hi
    return readlane_f64(r, 0) + readlane_f64(r, 16);
}

template <int NIO>
__global__ void __launch_bounds__(KhMm<NIO>::THREADS)
kh_mm_forward_update(KhSweepArgs p, const cplx *const *__restrict__ sq, KhUpdateArgs u, KhExchange ex,
                     const double *__restrict__ mm_tab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KhMmLds s = kh_mm_carve(smem);
    double(*red)[8] = (double(*)[8])s.red;  // [parity][wave]
    double(*D_sh)[2] = (double(*)[2])s.D;   // [parity][value, ok]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int hi = lane >> 4, b = (lane >> 2) & 3, lo = lane & 3;
    if (tid <= KH_MAX_DEGREE) s.deg[tid] = p.q2_theta[tid];
    const int N = p.N, nt = p.nt, k = blockIdx.x;
    constexpr int THREADS = KhMm<NIO>::THREADS;
    constexpr int WAVES = KhMm<NIO>::WAVES;
    constexpr double fim = -1.0;        // f = -i: forward propagation in Hilbert space; control operator Hermitian
                                        // (adj_sign = +1): both checked by the host
    const int kind = lo >> 1;           // 0: lanes of the phi chain (columns 0, 1), 1: of the w chain (2, 3)
    const bool isF = kind == 0;
    const double sgn = (lo & 1) ? 1.0 : -1.0;
    // own rows: R(io) = 4 (NIO wave + io) + hi, component lo & 1.  Writer of block io: the lane with
    // b == (NIO wave + io) & 3, into its own lane's slot of register j' = (NIO wave) >> 2
    const int iw = (b - NIO * wave) & 3;  // < NIO: writes v[iw]
    const bool wr = iw < NIO;
    const int wj = (NIO * wave) >> 2;
    const int waddr = (wj >> 1) * 128 + lane * 2 + (wj & 1);
    const double maskd = (b == 0 && isF) ? 1.0 : 0.0;
    int nmv = 0;  // products issued (workgroup-uniform)
    // one wave of each SIMD's pair runs its matrix-core block first; its partner's block then covers the first
    // one's vector tail (both tails at once would queue on the SIMD's VALU)
    if (WAVES == 8 && wave < 4) __builtin_amdgcn_s_setprio(1);

    const cplx *const *ops_k = p.ops + (size_t)k * 2;
    const cplx *const *sq_k = sq + (size_t)k * 3;
    kh_mm_stage_tile<NIO>(sq_k[1], N, wave, lane, tid, s.p1);
    kh_mm_stage_tile<NIO>(sq_k[2], N, wave, lane, tid, s.p2);
    KhMmTile<NIO> h1, A, B;
    kh_mm_load_tile<NIO>(ops_k[1], N, wave, lane, h1);
    const double nrm0 = kh_uniform(p.op_norms[(size_t)k * 2]), nrm1 = kh_uniform(p.op_norms[(size_t)k * 2 + 1]);
    const double chi_norm = kh_uniform(u.chi_norms[k]);

    // own rows of chi(t_n), component lo & 1 (index clamped: rows past the end are never used)
    auto load_chi = [&](int n, double (&c)[NIO]) {
        const int nn = n < nt ? n : nt - 1;
        const double *base = (const double *)(u.chi_store + ((size_t)k * nt + nn) * N);
#pragma unroll
        for (int io = 0; io < NIO; ++io) {
            const int R = 4 * (NIO * wave + io) + hi;
            c[io] = R < N ? base[2 * R + (lo & 1)] : 0.0;
        }
    };
    // v[io] of the writer lanes -> their own slots (both column pairs have writers: `on` selects)
    auto put = [&](int slot, bool on, const double (&v)[NIO]) {
        double sel = v[0];
#pragma unroll
        for (int io = 1; io < NIO; ++io) sel = iw == io ? v[io] : sel;
        if (wr && on) s.x0[slot * KH_MM_XLEN + waddr] = sel;
    };
    // the phi-chain lanes' values into the SECOND column pair (slot of lane + 2)
    auto put_f2w = [&](int slot, const double (&v)[NIO]) {
        double sel = v[0];
#pragma unroll
        for (int io = 1; io < NIO; ++io) sel = iw == io ? v[io] : sel;
        if (wr && isF) s.x0[slot * KH_MM_XLEN + waddr + 4] = sel;
    };

    // ---- per-lane state carried across intervals --------------------------------------------------------
    //   own : phi(t_n) (phi lanes) | w(t_{n+1}) = +-H1 chi(t_{n+1}) (w lanes)
    //   (w(t_{n+2}) waits in LDS: second column pair of the A product's operand)
    //   u1  : -                    | A(eps_{n-1}) w(t_{n+1})         u2 : - | H1 w(t_{n+1})
    double own[NIO], u1[NIO], u2[NIO], chin[NIO];
    double eps_last = 0.0;  // the pulse value A had when u1 was formed

    // ---- prologue --------------------------------------------------------------------------------------
    const int n0 = u.n_begin;
#pragma unroll
    for (int io = 0; io < NIO; ++io) {
        const int R = 4 * (NIO * wave + io) + hi;
        own[io] = (isF && R < N) ? ((const double *)(u.phi + (size_t)k * N))[2 * R + lo] : 0.0;
    }
    {   // A <- H0 (B is set at the first refresh below)
#pragma unroll
        for (int io = 0; io < NIO; ++io)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const cplx h0 = kh_mm_elem<NIO>(ops_k[0], N, wave, lane, io, j);
                A.re[io][j] = h0.x;
                A.im[io][j] = h0.y;
                B.re[io][j] = 0.0;
                B.im[io][j] = 0.0;
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    {   // H1 [chi(t_n0) | chi(t_n0+1)]: the first partial sum and w(t_n0+1)
        double c0v[NIO], c1v[NIO], x[4], y[NIO];
        load_chi(n0, c0v);
        load_chi(n0 + 1, c1v);
        put(1, isF, c0v);
        put(1, !isF, c1v);
        put(0, isF, own);
        __syncthreads();
        kh_mm_readx(s.x0 + 1 * KH_MM_XLEN, lane, x);
        kh_mm_pass<NIO>(h1, x, sgn, y);
        nmv += 1;
        double v = 0.0;
#pragma unroll
        for (int io = 0; io < NIO; ++io) {
            v = fma(own[io], dpp_move<KH_DPP_XOR1>(y[io]), v);  // Im <w(t_n0) | phi>: phi lanes
            if (!isF) own[io] = y[io];
        }
        const double tot = kh_mm_wave_sum(maskd * sgn * v);
        if (lane == 0) red[n0 & 1][wave] = tot;
        put(0, !isF, own);
        put(2, !isF, own);
        load_chi(n0 + 2, c0v);
        put(2, isF, c0v);
        __syncthreads();
        // H1 [chi(t_n0+2) | w(t_n0+1)] and H0 [- | w(t_n0+1)]
        kh_mm_readx(s.x0 + 2 * KH_MM_XLEN, lane, x);
        kh_mm_pass<NIO>(h1, x, sgn, y);
        double ya[NIO];
        kh_mm_pass<NIO>(A, x, sgn, ya);
        nmv += 2;
        double wF[NIO];
#pragma unroll
        for (int io = 0; io < NIO; ++io) {
            u2[io] = y[io];
            u1[io] = ya[io];
            wF[io] = y[io];                          // phi lanes: w(t_n0+2)
        }
        put_f2w(KH_MM_SLOT_XS + (n0 & 1), wF);
        put_f2w(KH_MM_SLOT_XH + (n0 & 1), wF);
        load_chi(n0 + 3, chin);  // (goes to the H1 product's operand behind the first product of the interval)
    }
    __syncthreads();
    // the exchange of one interval: wave 0 publishes the workgroup's partial sum; later every wave gathers its
    // share of the slots (loads issued ahead by exch_begin) and the total is formed after a barrier
    double part[1] = {0.0};
    KhGatherPart gp;
    gp.a = gp.b = 0;
#ifdef KH_MM_X_NOEXCH  // (timing experiment: results are wrong)
    const bool solo = true;
#else
    const bool solo = ex.G == 1 && ex.world == 1;  // a single workgroup on a single GPU: nothing to exchange
#endif
    auto exch_publish = [&](int n) {  // (after the barrier that completes red[n & 1])
        if (wave == 0) {
            double acc = 0.0;
#pragma unroll
            for (int w8 = 0; w8 < WAVES; ++w8) acc += red[n & 1][w8];
            part[0] = chi_norm * acc;
            kh_exchange_publish(ex, n, k, 1, lane, part);
        }
    };
    auto exch_begin = [&](int n) {
        if (!solo) kh_gather_part_begin<WAVES>(ex, n & 1, (unsigned)(n + 1), wave, lane, gp);
    };
    auto exch_end = [&](int n) {  // contains barriers: called by all threads
        if (solo) {
            if (tid == 0) {
                D_sh[n & 1][0] = part[0];
                D_sh[n & 1][1] = 1.0;
            }
            __syncthreads();
            return;
        }
        double ps = 0.0;
        const bool ok = kh_gather_part_end<WAVES>(ex, n & 1, (unsigned)(n + 1), wave, lane, gp, &ps);
        if (lane == 0) {
            s.gsum[wave * 2] = ps;
            s.gsum[wave * 2 + 1] = ok ? 1.0 : 0.0;
        }
        __syncthreads();
        if (ex.world > 1) {  // second stage across the GPUs (kh_common.h): by wave 0, broadcast through LDS
            if (wave == 0) {
                double tot = 0.0, good = 1.0;
#pragma unroll
                for (int w8 = 0; w8 < WAVES; ++w8) {
                    tot += s.gsum[w8 * 2];
                    good *= s.gsum[w8 * 2 + 1];
                }
                bool ok2 = good != 0.0;
                double D[1] = {tot};
                if (ok2) {
                    const unsigned int epoch = ex.epoch_base + (unsigned)(n + 1);
                    if (k == 0) kh_p2p_publish(ex, n & 1, 1, lane, D, epoch);
                    ok2 = kh_p2p_gather<1>(ex, n & 1, 1, epoch, lane, D);
                }
                if (lane == 0) {
                    D_sh[n & 1][0] = D[0];
                    D_sh[n & 1][1] = ok2 ? 1.0 : 0.0;
                }
            }
            __syncthreads();
        }
    };
    // the interval's total and its validity: from D_sh (one workgroup, or several GPUs) or straight from the waves'
    // partial sums (same order of additions in every workgroup)
    auto exch_result = [&](int n, double &D, bool &ok) {
        if (solo || ex.world > 1) {
            D = D_sh[n & 1][0];
            ok = D_sh[n & 1][1] != 0.0;
            return;
        }
        double tot = 0.0, good = 1.0;
#pragma unroll
        for (int w8 = 0; w8 < WAVES; ++w8) {
            tot += s.gsum[w8 * 2];
            good *= s.gsum[w8 * 2 + 1];
        }
        D = tot;
        ok = good != 0.0;
    };
    exch_publish(n0);
    exch_begin(n0);
    double dt_next = kh_uniform(p.dt[n0]), guess_next = kh_uniform(u.guess[n0]), shape_next = kh_uniform(u.shape[n0]);
    const double lam = kh_uniform(u.lambda[0]);
    double stepw_next = kh_uniform(shape_next / lam);
    KhDegreeCache dc = {12, 1.0, 0.0};
    int m_rows = -1;
    double g_a_loc = 0.0, eps_prev = 0.0;
    exch_end(n0);

#ifdef KH_TIMING
    long long t_crit = 0, t_shadow = 0, t_ex = 0, t_build = 0;
#endif
    for (int nr = n0; nr < u.n_end; nr += KH_MM_REFRESH) {
    {   // restart from A = H0, B = P0 (global memory), eps' = 0: the incremental updates below cannot drift
#pragma unroll
        for (int io = 0; io < NIO; ++io)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const cplx h0 = kh_mm_elem<NIO>(ops_k[0], N, wave, lane, io, j);
                const cplx q0 = kh_mm_elem<NIO>(sq_k[0], N, wave, lane, io, j);
                A.re[io][j] = h0.x;
                A.im[io][j] = h0.y;
                B.re[io][j] = q0.x;
                B.im[io][j] = q0.y;
                __builtin_amdgcn_sched_barrier(0);  // (one element at a time: no 64 registers of loads in flight)
            }
        eps_prev = 0.0;
    }
    const int n1 = nr + KH_MM_REFRESH < u.n_end ? nr + KH_MM_REFRESH : u.n_end;
    for (int n = nr; n < n1; ++n) {
        const int par = n & 1;
#ifdef KH_TIMING
        const long long tq0 = clock64();
        const bool trace_on = k == 0 && tid == 0 && n == n0 + 2000 && p.stats != nullptr;
        KH_MM_TRACE(0);
#endif
        double d1;
        bool d_ok;
        exch_result(n, d1, d_ok);
        if (!d_ok) return;
        // ---- pulse update (optimize.py:471-477) ----
        const double dt = dt_next, guess = guess_next, stepw = stepw_next;
        const double eps = kh_uniform(guess + stepw * d1);
        g_a_loc = kh_uniform(g_a_loc + stepw * (d1 * d1) * dt);
        if (k == 0 && tid == 0) u.opt[n] = eps;
#ifndef KH_MM_X_NOREBUILD
        {  // A += (eps - eps') H1,  B += (eps - eps') P1 + (eps^2 - eps'^2) P2
            const double e1 = eps - eps_prev, e2 = e1 * (eps + eps_prev);
#pragma unroll
            for (int io = 0; io < NIO; ++io)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const cplx q1 = s.p1[(io * 4 + j) * THREADS + tid], q2 = s.p2[(io * 4 + j) * THREADS + tid];
                    B.re[io][j] = fma(e2, q2.x, fma(e1, q1.x, B.re[io][j]));
                    B.im[io][j] = fma(e2, q2.y, fma(e1, q1.y, B.im[io][j]));
                    A.re[io][j] = fma(e1, h1.re[io][j], A.re[io][j]);
                    A.im[io][j] = fma(e1, h1.im[io][j], A.im[io][j]);
                    if (j == 3) __builtin_amdgcn_sched_barrier(0);  // (four elements' reads in flight, not sixteen)
                }
        }
#endif
        KH_MM_TRACE(1);
        eps_prev = eps;
        double dt_ld = 0.0, guess_ld = 0.0, shape_ld = 0.0;  // the next interval's scalars (loaded behind the first product)
        const bool with_d = n + 1 < nt - 1;  // a partial sum of interval n+1 is due
        int nsub, m;
        kh_degree_cached((nrm0 + fabs(eps) * nrm1) * dt, s.deg, p.theta_max, p.inv_theta_max, dc, &nsub, &m);
        if (m != m_rows) {  // (rare along a smooth pulse)
            __syncthreads();
            for (int i = tid; i < KH_MM_TAB_STRIDE; i += THREADS) s.tab[i] = mm_tab[(size_t)m * KH_MM_TAB_STRIDE + i];
            __syncthreads();
            m_rows = m;
        }
        const int P = (m + 1) >> 1;
        const double h = nsub == 1 ? dt : dt / nsub;
        const double mh2 = -(h * h);  // (f h)^2
        const double2 ini = *(const double2 *)(s.tab + (size_t)(KH_Q2_ROWS * 2 + kind) * 4);
        // w' = A(eps) w = u1 + (eps - eps_last) u2 on the w lanes;  phi lanes: their copies of w and w' for the sum
        // (few registers are free next to three operator tiles: what the sum needs of w and w' on the phi lanes is
        // formed where it is used)
        const double deps = eps - eps_last;
        double g2[NIO];
#pragma unroll
        for (int io = 0; io < NIO; ++io) {
            const double wp = fma(deps, u2[io], u1[io]);
            g2[io] = isF ? 0.0 : (-fim * sgn) * dpp_move<KH_DPP_XOR1>(wp);  // conj(f) w'
        }
#ifdef KH_TIMING
        const long long tq1 = clock64();
        t_build += tq1 - tq0;
        KH_MM_TRACE(2);
#endif
        // ---- the common interval, unrolled: one sub-step, a partial sum due, P products with P known at compile
        // time -- every decision the generic loop below takes per product is a constant here
        auto fast_interval = [&](auto Pc) {
            constexpr int P = decltype(Pc)::value, Pf = (P + 1) / 2;
            double acc1[NIO], acc2[NIO], yS[NIO], wF[NIO];
#pragma unroll
            for (int io = 0; io < NIO; ++io) {
                acc1[io] = ini.x * own[io];
                acc2[io] = (h * ini.y) * own[io];
            }
            kh_static_for<P>([&](auto Tc) {
                constexpr int t = decltype(Tc)::value;
                constexpr int xin = t == 0 ? 0 : 1 + ((t - 1) & 1), xout = 1 + (t & 1);
                double x[4], y[NIO];
                kh_mm_readx(s.x0 + xin * KH_MM_XLEN, lane, x);
                const double *tb = s.tab + (size_t)(t * 2 + kind) * 4;
                const double2 tb01 = *(const double2 *)tb;
                const double c2t = tb[2];
                kh_mm_pass<NIO>(B, x, sgn, y);
                const double sc = mh2 * tb01.x, hc = h * c2t;
#pragma unroll
                for (int io = 0; io < NIO; ++io) {
                    y[io] *= sc;
                    acc1[io] = fma(tb01.y, y[io], acc1[io]);
                    acc2[io] = fma(hc, y[io], acc2[io]);
                }
                if constexpr (t + 1 < Pf) {  // the w chain takes its injection a_q w + h b_q conj(f) w'
                    double out[NIO];
#pragma unroll
                    for (int io = 0; io < NIO; ++io) out[io] = isF ? y[io] : fma(hc, g2[io], fma(tb01.y, own[io], y[io]));
                    put(xout, true, out);
                } else if constexpr (t + 1 < P) {
                    put(xout, true, y);
                }
                if constexpr (t == P - 2) put(KH_MM_SLOT_XS + par, isF, acc2);  // s is complete
                if constexpr (t == 0) {
                    put(KH_MM_SLOT_XH + par, isF, chin);  // chi(t_{n+3}), loaded an interval ago
                    if (n + 1 < nt - 1) {  // (issued here, read at the end of the interval: never waited for)
                        dt_ld = p.dt[n + 1];
                        guess_ld = u.guess[n + 1];
                        shape_ld = u.shape[n + 1];
                    }
                }
                if constexpr (t == Pf - 1) {  // Im <w | phi(t_{n+1})> from the two half chains
                    double v = 0.0;
#pragma unroll
                    for (int io = 0; io < NIO; ++io) {
                        const double wrev = dpp_move<KH_DPP_REV4>(own[io]);                   // w, other component
                        const double wpF = dpp_move<KH_DPP_XOR2>(fma(deps, u2[io], u1[io]));  // w', same component
                        const double e1 = fma(y[io], dpp_move<KH_DPP_REV4>(y[io]), acc1[io] * wrev);
                        v += fma(fim, wpF * acc2[io], sgn * e1);
                    }
                    const double tot = kh_mm_wave_sum(maskd * v);
                    if (lane == 0) red[(n + 1) & 1][wave] = tot;
                }
                if constexpr (t == Pf) {  // first product in the shadow: H1 [chi(t_{n+3}) | w(t_{n+2})]
                    kh_mm_readx(s.x0 + (KH_MM_SLOT_XH + par) * KH_MM_XLEN, lane, x);
                    kh_mm_pass<NIO>(h1, x, sgn, wF);
                }
                if constexpr (t == P - 1) {  // A [s | w(t_{n+2})]: the odd terms, and u1 of the next interval
                    kh_mm_readx(s.x0 + (KH_MM_SLOT_XS + par) * KH_MM_XLEN, lane, x);
                    kh_mm_pass<NIO>(A, x, sgn, yS);
                    exch_begin(n + 1);  // the exchange's granule loads go out here: back when the interval is done
                }
                if constexpr (t + 1 < P || t == Pf - 1) __syncthreads();
                if constexpr (t == Pf - 1) {
#ifdef KH_TIMING
                    t_crit += clock64() - tq1;
#endif
                    exch_publish(n + 1);
                }
            });
            if constexpr (Pf == P) {  // (P == 1 is not routed here; kept for completeness)
                double x[4];
                kh_mm_readx(s.x0 + (KH_MM_SLOT_XH + par) * KH_MM_XLEN, lane, x);
                kh_mm_pass<NIO>(h1, x, sgn, wF);
            }
            nmv += P + 2;
            // new state: even terms + f A s;  the w pipeline moves on (see the generic loop)
#pragma unroll
            for (int io = 0; io < NIO; ++io) {
                const double nv = fma(fim * sgn, dpp_move<KH_DPP_XOR1>(yS[io]), acc1[io]);
                const int bw = (NIO * wave + io) & 3;
                const double w2 = s.x0[(KH_MM_SLOT_XS + par) * KH_MM_XLEN + (wj >> 1) * 128 + (16 * hi + 4 * bw + lo) * 2 + (wj & 1)];
                own[io] = isF ? nv : w2;
                if (!isF) {
                    u1[io] = yS[io];
                    u2[io] = wF[io];
                }
            }
            eps_last = eps;
            put_f2w(KH_MM_SLOT_XS + (par ^ 1), wF);
            put_f2w(KH_MM_SLOT_XH + (par ^ 1), wF);
            load_chi(n + 4, chin);  // for the next interval's H1 product: in flight across the exchange
            put(0, true, own);
        };
        bool fast_done = false;
#ifndef KH_MM_NO_FAST
        if (nsub == 1 && with_d) {
            fast_done = true;
            switch (P) {
                case 3: fast_interval(std::integral_constant<int, 3>{}); break;
                case 4: fast_interval(std::integral_constant<int, 4>{}); break;
                case 5: fast_interval(std::integral_constant<int, 5>{}); break;
                case 6: fast_interval(std::integral_constant<int, 6>{}); break;
                case 7: fast_interval(std::integral_constant<int, 7>{}); break;
                case 8: fast_interval(std::integral_constant<int, 8>{}); break;
                default: fast_done = false;
            }
        }
#endif
        for (int sub = fast_done ? nsub : 0; sub < nsub; ++sub) {
            // the partial sum is taken in the middle of the LAST sub-step; earlier ones run straight through
            const bool mid = with_d && sub + 1 == nsub;
            const bool lastsub = sub + 1 == nsub;
            const int Pf = mid ? (P + 1) >> 1 : P;
            double acc1[NIO], acc2[NIO], yS[NIO], wF[NIO];
#pragma unroll
            for (int io = 0; io < NIO; ++io) {
                acc1[io] = ini.x * own[io];
                acc2[io] = (h * ini.y) * own[io];
                yS[io] = 0.0;
                wF[io] = 0.0;
            }
            if (P == 1) {  // (degree <= 2: s cannot ride on an earlier product's barrier)
                put(KH_MM_SLOT_XS + par, isF, acc2);
                __syncthreads();
            }
            bool h_done = false, g_issued = false;
            int xin = 0;  // slot of the product's input
            for (int t = 0; t < P; ++t) {
                // the exchange's granule loads go out one product ahead of their use (wave 0)
                double x[4], y[NIO];
                kh_mm_readx(s.x0 + xin * KH_MM_XLEN, lane, x);
                const double *tb = s.tab + (size_t)(t * 2 + kind) * 4;
                const double2 tb01 = *(const double2 *)tb;
                const double c2t = tb[2];
                kh_mm_pass<NIO>(B, x, sgn, y);
                nmv += 1;
                const double sc = mh2 * tb01.x;
                double out[NIO];
#pragma unroll
                for (int io = 0; io < NIO; ++io) {
                    y[io] *= sc;
                    acc1[io] = fma(tb01.y, y[io], acc1[io]);
                    acc2[io] = fma(h * c2t, y[io], acc2[io]);
                    out[io] = fma(h * c2t, g2[io], fma(tb01.y, isF ? 0.0 : own[io], y[io]));  // (phi lanes: y itself)
                }
                KH_MM_TRACE(3 + 4 * t);
                const int xout = 1 + (t & 1);
                if (t + 1 < P) put(xout, true, out);
                if (t == P - 2) put(KH_MM_SLOT_XS + par, isF, acc2);  // s is complete
                if (t == 0 && lastsub) {
                    put(KH_MM_SLOT_XH + par, isF, chin);  // chi(t_{n+3}), loaded an interval ago
                    if (n + 1 < nt - 1) {  // (issued here, read at the end of the interval: never waited for)
                        dt_ld = p.dt[n + 1];
                        guess_ld = u.guess[n + 1];
                        shape_ld = u.shape[n + 1];
                    }
                }
                if (mid && t == Pf - 1) {  // Im <w | phi(t_{n+1})> from the two half chains
                    double v = 0.0;
#pragma unroll
                    for (int io = 0; io < NIO; ++io) {
                        const double wrev = dpp_move<KH_DPP_REV4>(own[io]);                         // w, other component
                        const double wpF = dpp_move<KH_DPP_XOR2>(fma(deps, u2[io], u1[io]));        // w', same component
                        const double e1 = fma(y[io], dpp_move<KH_DPP_REV4>(y[io]), acc1[io] * wrev);
                        v += fma(fim, wpF * acc2[io], sgn * e1);
                    }
                    const double tot = kh_mm_wave_sum(maskd * v);
                    if (lane == 0) red[(n + 1) & 1][wave] = tot;
                }
                if (t == P - 1) {  // A [s | w(t_{n+2})]: the odd terms, and u1 of the next interval
                    kh_mm_readx(s.x0 + (KH_MM_SLOT_XS + par) * KH_MM_XLEN, lane, x);
                    kh_mm_pass<NIO>(A, x, sgn, yS);
                    nmv += 1;
                    if (mid && t >= Pf) {  // the exchange's granule loads go out here: back when the interval is done
                        exch_begin(n + 1);
                        g_issued = true;
                    }
                }
                if (mid && t == Pf && !h_done) {  // first product in the shadow: H1 [chi(t_{n+3}) | w(t_{n+2})]
                    kh_mm_readx(s.x0 + (KH_MM_SLOT_XH + par) * KH_MM_XLEN, lane, x);
                    kh_mm_pass<NIO>(h1, x, sgn, wF);
                    nmv += 1;
                    h_done = true;
                }
                KH_MM_TRACE(4 + 4 * t);
#ifndef KH_MM_X_NOBARRIER
                if (t + 1 < P || (mid && t == Pf - 1)) __syncthreads();
#endif
                KH_MM_TRACE(5 + 4 * t);
                if (mid && t == Pf - 1) {
#ifdef KH_TIMING
                    t_crit += clock64() - tq1;
#endif
                    exch_publish(n + 1);
                }
                KH_MM_TRACE(6 + 4 * t);
                xin = xout;
            }
            if (mid && !h_done) {  // (P == Pf: no product ran in the shadow)
                double x[4];
                kh_mm_readx(s.x0 + (KH_MM_SLOT_XH + par) * KH_MM_XLEN, lane, x);
                kh_mm_pass<NIO>(h1, x, sgn, wF);
                nmv += 1;
            }
            if (P == 1 && !mid) __syncthreads();  // (x0 was read by this sub-step's only product)
            if (mid && !g_issued) exch_begin(n + 1);
            // new state: even terms + f A s   (f = i fim:  (re, im) += fim (-y_im, y_re))
#pragma unroll
            for (int io = 0; io < NIO; ++io) {
                const double nv = fma(fim * sgn, dpp_move<KH_DPP_XOR1>(yS[io]), acc1[io]);
                if (isF) own[io] = nv;
            }
            if (lastsub) {
                // shift the w pipeline: w(t_{n+2}) becomes the next interval's w, with its u1 = A(eps_n) w, u2 = H1 w;
                // w(t_{n+3}) (phi lanes of the H1 product) moves to the w lanes and into the next operands
#pragma unroll
                for (int io = 0; io < NIO; ++io) {
                    // (own rows of w(t_{n+2}): the slot of the block's writer lane, this lane's hi and lo)
                    const int bw = (NIO * wave + io) & 3;
                    const double w2 = s.x0[(KH_MM_SLOT_XS + par) * KH_MM_XLEN + (wj >> 1) * 128 + (16 * hi + 4 * bw + lo) * 2 + (wj & 1)];
                    if (!isF) {
                        own[io] = w2;
                        u1[io] = yS[io];
                        u2[io] = wF[io];
                    }
                }
                eps_last = eps;
                if (mid) {
                    put_f2w(KH_MM_SLOT_XS + (par ^ 1), wF);
                    put_f2w(KH_MM_SLOT_XH + (par ^ 1), wF);
                    load_chi(n + 4, chin);  // for the next interval's H1 product: in flight across the exchange
                }
            }
            put(0, true, own);
            if (sub + 1 < nsub) __syncthreads();
        }
#ifdef KH_TIMING
        const long long tq2 = clock64();
        t_shadow += tq2 - tq1;
        KH_MM_TRACE(40);
#endif
        dt_next = kh_uniform(dt_ld);
        guess_next = kh_uniform(guess_ld);
        shape_next = kh_uniform(shape_ld);
        stepw_next = kh_uniform(shape_next / lam);  // (S/lambda of the coming interval: the division is off the critical path here)
        KH_MM_TRACE(41);
        if (with_d)
            exch_end(n + 1);
        else
            __syncthreads();
        KH_MM_TRACE(42);
#ifdef KH_TIMING
        t_ex += clock64() - tq2;
#endif
    }
    }
#ifdef KH_TIMING
    if (tid == 0 && k == 0 && p.stats != nullptr) {
        p.stats[1] = (double)t_crit;    // tiles rebuilt -> partial sum ready
        p.stats[2] = (double)t_shadow;  // tiles rebuilt -> end of the interval's products
        p.stats[3] = (double)t_ex;      // waiting for the exchange after that
        p.stats[0] = -(double)t_build;  // eps known -> tiles rebuilt (the timing build gives up the product count)
    }
#endif
    if (isF && b == 0) {
#pragma unroll
        for (int io = 0; io < NIO; ++io) {
            const int R = 4 * (NIO * wave + io) + hi;
            if (R < N) ((double *)(u.phi + (size_t)k * N))[2 * R + lo] = own[io];
        }
    }
    if (k == 0 && tid == 0) u.g_a[0] = g_a_loc;
#ifndef KH_TIMING
    if (tid == 0 && p.stats != nullptr) atomicAdd(p.stats, (double)nmv);
#endif
}
