// Hermitian-packed two-terms-per-phase kernels: N <= 64, one control, EVERY operator equal to its own adjoint bit for
// bit (closed-system Hamiltonians), 256 < K <= 512 objectives -- TWO workgroups per CU.
//
// kh_tile64q2.h keeps full 64 x 64 tiles of A, B = A^2 and H1 in registers (96 VGPRs at 512 threads) and P1, P2 in LDS
// (128 KiB): one workgroup fills a CU, and with more objectives than CUs the engine had to fall back to the
// one-term-per-phase kernel (13 products per step instead of 8).  A Hermitian matrix is determined by half of its
// elements.  Stored that way A, B, H1 take 60 VGPRs and P1 + P2 66 KiB of LDS: two 512-thread workgroups (128 VGPRs, 79 KiB
// each) share a CU, 512 objectives stay co-resident for the in-kernel exchange of the update sums, and the two
// workgroups fill each other's latency gaps (the single-workgroup kernels leave the FMA pipe idle 80 % of the time).
//
// Packed layout ("cyclic diagonals").  Lane i of wave w holds, of every operator M,
//     slot q = 0..3:  M[i][(i + d) mod 64],  d = 4 w + 1 + q   (d = 1..32; for d = 32 only lanes i < 32: the pair
//                                                               {i, i + 32} would otherwise be held twice)
//     slot 4       :  M[i][i]                                   (used by wave 7 only)
// so every unordered pair {i, j} is held exactly once.  A product y = M x then has two halves per held element:
//     direct      y_i       += M[i][i+d] x_{i+d}           x_{i+d}: one ROTATED, conflict-free LDS read per slot
//     transposed  y_{i+d}   += conj(M[i][i+d]) x_i         destined to another row: collected per wave by a Horner
//                                                          chain of single-lane wave rotations (wave_ror:1 DPP) over its
//                                                          four consecutive d, then ONE ds_bpermute by the wave's
//                                                          first d brings it to the destination lane
// The eight waves' partial vectors meet in LDS (8 KiB); after a barrier every row's eight partials are added on the
// matrix core (v_mfma_f64_4x4x4 with a constant B operand, as in kh_tile64q2.h) in a fixed order.  Same FMA count as
// the full-tile product, 5 LDS vector reads per lane instead of 8, two barriers per product instead of one: longer
// latency per phase, which the second workgroup of the CU hides.
//
// Series, A^2 chain, one A product by linearity, tile advance, adjoint-side partial sums, exchange: as kh_tile64q2.h.
#pragma once

#include "kh_common.h"
#include "kh_generic.h"
#include "kh_tile64.h"
#include "kh_tile64q2.h"

#define KH_Q2H_THREADS 512
#define KH_Q2H_PACKED (4 * KH_Q2H_THREADS + 64)  // complex elements of one packed operator: four slots per lane + the diagonal

struct KhQ2hLds {
    cplx *p1, *p2;            // [KH_Q2H_PACKED] each
    cplx (*part)[KH_TILE_N];  // [8 waves][64] the waves' partial result vectors
    cplx (*buf)[KH_TILE_N];   // [2][64]
    cplx *chib;               // [64] chi(t_{n+1}) for the adjoint-side product
    cplx *sbuf;               // [64] the vector s of the A^2 chain
    double *red;              // [2][8 waves]
    double *D;                // [2][2]
    double2 *inv2;            // [KH_Q2_ROWS + 1]
    double *deg;              // [KH_MAX_DEGREE + 1]
};

__host__ __device__ inline size_t kh_q2h_lds_bytes() {
    return (size_t)2 * KH_Q2H_PACKED * sizeof(cplx) + (8 + 4) * KH_TILE_N * sizeof(cplx) + (2 * 8 + 4) * sizeof(double) +
           (KH_Q2_ROWS + 1) * sizeof(double2) + (KH_MAX_DEGREE + 1) * sizeof(double) + 8;
}

__device__ __forceinline__ KhQ2hLds kh_q2h_carve(char *smem) {
    KhQ2hLds s;
    s.p1 = (cplx *)smem;
    s.p2 = s.p1 + KH_Q2H_PACKED;
    s.part = (cplx(*)[KH_TILE_N])(s.p2 + KH_Q2H_PACKED);
    s.buf = s.part + 8;
    s.chib = (cplx *)(s.buf + 2);
    s.sbuf = s.chib + KH_TILE_N;
    s.inv2 = (double2 *)(s.sbuf + KH_TILE_N);
    s.red = (double *)(s.inv2 + KH_Q2_ROWS + 1);
    s.D = s.red + 2 * 8;
    s.deg = s.D + 4;
    return s;
}

// dense row-major N x N (N <= 64) -> packed layout; one 512-thread block per operator (engine set-up)
__global__ void kh_q2h_pack(const cplx *__restrict__ op, cplx *__restrict__ out, int N) {
    const int tid = threadIdx.x, w = tid >> 6, i = tid & 63;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int d = 4 * w + 1 + q, col = (i + d) & 63;
        cplx v = c_make(0.0, 0.0);
        if (op != nullptr && i < N && col < N && !(d == 32 && i >= 32)) v = op[(size_t)i * N + col];
        out[q * KH_Q2H_THREADS + tid] = v;
    }
    if (tid < 64) out[4 * KH_Q2H_THREADS + tid] = (op != nullptr && tid < N) ? op[(size_t)tid * N + tid] : c_make(0.0, 0.0);
}

__device__ __forceinline__ void kh_q2h_load(const cplx *__restrict__ pk, int tid, int lane, cplx (&t)[5]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = pk[q * KH_Q2H_THREADS + tid];
    t[4] = pk[4 * KH_Q2H_THREADS + lane];
}

__device__ __forceinline__ void kh_q2h_stage(const cplx *__restrict__ pk, int tid, cplx *dst) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q * KH_Q2H_THREADS + tid] = pk[q * KH_Q2H_THREADS + tid];
    if (tid < 64) dst[4 * KH_Q2H_THREADS + tid] = pk[4 * KH_Q2H_THREADS + tid];
}

// A += e1 H1,  B += e1 P1 + e2 P2   (P1, P2 from LDS; e1 = eps - eps', e2 = eps^2 - eps'^2)
__device__ __forceinline__ void kh_q2h_advance(const KhQ2hLds &s, int tid, int lane, double eps, double eps_prev,
                                               const cplx (&h1)[5], cplx (&a)[5], cplx (&b)[5]) {
    const double e1 = eps - eps_prev, e2 = e1 * (eps + eps_prev);
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int at = q < 4 ? q * KH_Q2H_THREADS + tid : 4 * KH_Q2H_THREADS + lane;
        const cplx q1 = s.p1[at], q2 = s.p2[at];
        a[q].x = fma(e1, h1[q].x, a[q].x);
        a[q].y = fma(e1, h1[q].y, a[q].y);
        b[q].x = fma(e2, q2.x, fma(e1, q1.x, b[q].x));
        b[q].y = fma(e2, q2.y, fma(e1, q1.y, b[q].y));
    }
}

// value of lane (i - 1) mod 64 (a rotation over the whole wave: DPP wave_ror:1)
__device__ __forceinline__ double kh_wave_ror1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x13C, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x13C, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// value of lane `src` (any permutation of the wave: through the LDS crossbar, no memory)
__device__ __forceinline__ double kh_wave_fetch(double v, int src) {
    const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// This wave's two halves of M x for its four cyclic diagonals: `acc` belongs to row `lane`, `tr` (after the Horner
// chain of rotations) to row (lane + 4 wave + 1) mod 64.  x: the vector in LDS.
__device__ __forceinline__ void kh_q2h_halves(const cplx (&m)[5], const cplx *x, int wave, int lane, cplx &acc, cplx &tr) {
    const cplx xi = x[lane];
    acc = c_make(0.0, 0.0);
    tr = c_make(0.0, 0.0);
#pragma unroll
    for (int q = 3; q >= 0; --q) {
        const cplx xj = x[(lane + 4 * wave + 1 + q) & 63];
        c_fma(acc, m[q], xj);
        if (q < 3) {
            tr.x = kh_wave_ror1(tr.x);
            tr.y = kh_wave_ror1(tr.y);
        }
        c_fma_conj(tr, m[q], xi);
    }
    if (wave == 7) c_fma(acc, m[4], xi);  // the diagonal
}

// part[wave][row] <- acc[row] + tr[row - (4 wave + 1)]
__device__ __forceinline__ void kh_q2h_publish(const KhQ2hLds &s, int wave, int lane, cplx acc, cplx tr) {
    const int src = (lane - (4 * wave + 1)) & 63;
    acc.x += kh_wave_fetch(tr.x, src);
    acc.y += kh_wave_fetch(tr.y, src);
    s.part[wave][lane] = acc;
}

// Second stage (after a barrier): lane l of wave w' reads partial l >> 3 of row 8 w' + (l & 7); the matrix core adds the
// eight partials of a row.  The sum of row  hrow(w', l) = 8 w' + 4 ((l >> 2) & 1) + (l >> 4)  comes back on every lane.
__device__ __forceinline__ int kh_q2h_row(int wave, int lane) { return 8 * wave + 4 * ((lane >> 2) & 1) + (lane >> 4); }
__device__ __forceinline__ bool kh_q2h_writer(int lane) { return (lane & 11) == 0; }
__device__ __forceinline__ cplx kh_q2h_rowsum(const KhQ2hLds &s, int wave, int lane, double scale) {
    const cplx v = s.part[lane >> 3][8 * wave + (lane & 7)];
    const double dx = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x, scale, 0.0, 0, 0, 0);
    const double dy = __builtin_amdgcn_mfma_f64_4x4x4f64(v.y, scale, 0.0, 0, 0, 0);
    return c_make(dx + dpp_move<KH_DPP_ROR8>(dx), dy + dpp_move<KH_DPP_ROR8>(dy));
}
// sum over the wave of a value that is zero except on the 8 writer lanes (0, 4, 16, 20, 32, 36, 48, 52); valid on lane 0
__device__ __forceinline__ double kh_q2h_writers_sum(double v) {
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);  // lanes {i, 16+i, 32+i, 48+i} -> class i
    return d + dpp_move<KH_DPP_ROR4>(d);  // classes 0 and 4 (on lanes 0.. 3 and 4 .. 7)
}

// y = M x for all rows: both stages; the result row kh_q2h_row(wave, lane) on every lane.  Two barriers.
__device__ __forceinline__ cplx kh_q2h_product(const KhQ2hLds &s, const cplx (&m)[5], const cplx *x, int wave, int lane,
                                               double scale) {
    cplx acc, tr;
    kh_q2h_halves(m, x, wave, lane, acc, tr);
    kh_q2h_publish(s, wave, lane, acc, tr);
    __syncthreads();
    const cplx y = kh_q2h_rowsum(s, wave, lane, scale);
    __syncthreads();  // (the partial vectors may be overwritten)
    return y;
}

// the series' rows of degree m -> LDS (as kh_q2_load_rows)
__device__ __forceinline__ void kh_q2h_load_rows(const KhSweepArgs &p, const KhQ2hLds &s, int m, int tid) {
    __syncthreads();
    if (tid < KH_Q2_ROWS) {
        const double *r = p.q2_rows + ((size_t)m * KH_Q2_ROWS + tid) * 2;
        s.inv2[tid] = make_double2(r[0], r[1]);
    }
    if (tid == KH_Q2_ROWS) s.inv2[KH_Q2_ROWS] = make_double2(p.q2_c0[m], 0.0);
    __syncthreads();
}

// state <- exp(f A dt) state, two terms per phase (kh_q2_expm_action in the packed layout).  On entry buf[cur] holds
// the state (all rows written, barrier passed); on exit buf[cur] holds the new state.  `state`: row
// kh_q2h_row(wave, lane) of it on every lane.
template <class Epilogue>
__device__ __forceinline__ int kh_q2h_expm_action(const KhQ2hLds &s, const cplx (&a)[5], const cplx (&b)[5], cplx &state,
                                                  int &cur, cplx *store_in, int N, double fre, double fim, double dt,
                                                  int nsub, int m, int wave, int lane, Epilogue epilogue) {
    const int row = kh_q2h_row(wave, lane);
    const bool writer = kh_q2h_writer(lane);
    const double h = nsub == 1 ? dt : dt / nsub;
    const double f2h2 = (fre * fre - fim * fim) * h * h;
    const int phases = (m + 1) >> 1;
    if (store_in != nullptr && wave == 0 && lane < N) store_in[lane] = s.buf[cur][lane];
    for (int sub = 0; sub < nsub; ++sub) {
        const double hr = h * s.inv2[0].x, c0 = s.inv2[KH_Q2_ROWS].x;
        cplx sacc = c_make(hr * state.x, hr * state.y);
        state = c_make(c0 * state.x, c0 * state.y);
        if (phases == 1) {
            if (writer) s.sbuf[row] = sacc;
            __syncthreads();
        }
        for (int ph = 0; ph < phases; ++ph) {
            const double c2 = f2h2 * s.inv2[ph].y;
            const bool last = (ph + 1 == phases);
            cplx acc, tr;
            kh_q2h_halves(b, s.buf[cur], wave, lane, acc, tr);
            if (last) {
                // the one A product (on s) rides in the same two stages: c2 B t + f A s, combined before the waves meet
                cplx acc_a, tr_a;
                kh_q2h_halves(a, s.sbuf, wave, lane, acc_a, tr_a);
                const cplx f = c_make(fre, fim);
                const cplx fa = c_mul(f, acc_a), ft = c_mul(f, tr_a);
                acc = c_make(fma(c2, acc.x, fa.x), fma(c2, acc.y, fa.y));
                tr = c_make(fma(c2, tr.x, ft.x), fma(c2, tr.y, ft.y));
            }
            kh_q2h_publish(s, wave, lane, acc, tr);
            __syncthreads();
            if (!last) {
                const cplx t2 = kh_q2h_rowsum(s, wave, lane, c2);
                state.x += t2.x;
                state.y += t2.y;
                const double hn = h * s.inv2[ph + 1].x;
                sacc.x = fma(hn, t2.x, sacc.x);
                sacc.y = fma(hn, t2.y, sacc.y);
                if (writer) {
                    s.buf[cur ^ 1][row] = t2;
                    if (ph + 2 == phases) s.sbuf[row] = sacc;
                }
            } else {
                const cplx t2 = kh_q2h_rowsum(s, wave, lane, 1.0);
                state.x += t2.x;
                state.y += t2.y;
                if (writer) s.buf[cur ^ 1][row] = state;
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
// hpk: [K*5] packed H0, H1, P0, P1, P2 of every objective (Hermitian: the same for both directions)
__global__ void __launch_bounds__(KH_Q2H_THREADS, 4)
kh_q2h_sweep_store(KhSweepArgs p, const cplx *const *__restrict__ hpk, const double *__restrict__ pulses,
                   const cplx *__restrict__ state_in, cplx *__restrict__ store, cplx *__restrict__ state_out, int direction) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KhQ2hLds s = kh_q2h_carve(smem);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid <= KH_MAX_DEGREE) s.deg[tid] = p.q2_theta[tid];
    const int row = kh_q2h_row(wave, lane);
    const bool writer = kh_q2h_writer(lane);
    const int N = p.N, nt = p.nt;
    double matvecs = 0.0;
    for (int k = blockIdx.x; k < p.K; k += gridDim.x) {
        const cplx *const *pk = hpk + (size_t)k * 5;
        __syncthreads();  // previous objective's readers are done with LDS
        kh_q2h_stage(pk[3], tid, s.p1);
        kh_q2h_stage(pk[4], tid, s.p2);
        cplx h1[5], a[5], b[5];
        kh_q2h_load(pk[1], tid, lane, h1);
        const double nrm0 = kh_uniform(p.op_norms[(size_t)k * 2]), nrm1 = kh_uniform(p.op_norms[(size_t)k * 2 + 1]);
        cplx state = row < N ? state_in[(size_t)k * N + row] : c_make(0.0, 0.0);
        int cur = 0;
        if (writer) s.buf[0][row] = state;
        __syncthreads();
        const int n0 = direction > 0 ? 0 : nt - 2;
        double eps_next = pulses[n0], dt_next = p.dt[n0];
        KhDegreeCache dc = {12, 1.0, 0.0};
        int m_rows = -1;
        double eps_prev = 0.0;
        for (int step0 = 0; step0 < nt - 1; step0 += KH_Q2_REFRESH) {
            // restart A = H0, B = P0 (outside the interval loop: one definition of the tiles in its body)
            kh_q2h_load(pk[0], tid, lane, a);
            kh_q2h_load(pk[2], tid, lane, b);
            eps_prev = 0.0;
            const int step_stop = step0 + KH_Q2_REFRESH < nt - 1 ? step0 + KH_Q2_REFRESH : nt - 1;
            for (int step = step0; step < step_stop; ++step) {
                const int n = direction > 0 ? step : nt - 2 - step;
                const double eps = kh_uniform(eps_next), dt = kh_uniform(dt_next);
                if (step + 1 < nt - 1) {
                    const int nn = direction > 0 ? n + 1 : n - 1;
                    dt_next = p.dt[nn];
                    eps_next = pulses[nn];
                }
                int nsub, m;
                kh_degree_cached((nrm0 + fabs(eps) * nrm1) * dt, s.deg, p.theta_max, p.inv_theta_max, dc, &nsub, &m);
                if (m != m_rows) {
                    kh_q2h_load_rows(p, s, m, tid);
                    m_rows = m;
                }
                kh_q2h_advance(s, tid, lane, eps, eps_prev, h1, a, b);
                eps_prev = eps;
                cplx *store_in = store == nullptr ? nullptr : store + ((size_t)k * nt + (direction > 0 ? n : n + 1)) * N;
                matvecs += kh_q2h_expm_action(s, a, b, state, cur, store_in, N, p.fre, p.fim, dt, nsub, m, wave, lane, [] {});
            }
        }
        if (store != nullptr && wave == 0 && lane < N)
            store[((size_t)k * nt + (direction > 0 ? nt - 1 : 0)) * N + lane] = s.buf[cur][lane];
        if (state_out != nullptr && wave == 0 && lane < N) state_out[(size_t)k * N + lane] = s.buf[cur][lane];
    }
    if (tid == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}

// ---------------------------------------------------------------------------
// forward sweep with sequential pulse update (optimize.py:444-508): ONE launch, grid == K <= 2 x #CUs, first order,
// partial sums on the adjoint side (H1 is Hermitian here: <chi|H1 phi> = <H1 chi|phi>)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(KH_Q2H_THREADS, 4)
kh_q2h_forward_update(KhSweepArgs p, const cplx *const *__restrict__ hpk, KhUpdateArgs u, KhExchange ex) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KhQ2hLds s = kh_q2h_carve(smem);
    double(*red)[8] = (double(*)[8])s.red;  // [parity][wave]
    double(*D_sh)[2] = (double(*)[2])s.D;   // [parity][value, ok]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid <= KH_MAX_DEGREE) s.deg[tid] = p.q2_theta[tid];
    const int row = kh_q2h_row(wave, lane);
    const bool writer = kh_q2h_writer(lane);
    const int N = p.N, nt = p.nt;
    const int k = blockIdx.x;
    double matvecs = 0.0;

    const cplx *const *pk = hpk + (size_t)k * 5;
    kh_q2h_stage(pk[3], tid, s.p1);
    kh_q2h_stage(pk[4], tid, s.p2);
    cplx h1[5], a[5], b[5];
    kh_q2h_load(pk[1], tid, lane, h1);
    double eps_prev = 0.0;
    const double nrm0 = kh_uniform(p.op_norms[(size_t)k * 2]), nrm1 = kh_uniform(p.op_norms[(size_t)k * 2 + 1]);
    const double chi_norm = kh_uniform(u.chi_norms[k]);

    cplx state = row < N ? u.phi[(size_t)k * N + row] : c_make(0.0, 0.0);
    int cur = 0;
    if (writer) s.buf[0][row] = state;
    cplx chi = c_make(0.0, 0.0);
    auto load_chi = [&](int n) { chi = row < N ? u.chi_store[((size_t)k * nt + n) * N + row] : c_make(0.0, 0.0); };
    // this wave's piece of Im(mu <w|state>) -> red[par][wave]  (w, state: one row per lane, eightfold; writers count)
    auto piece = [&](int par, cplx w) {
        cplx ov = c_make(0.0, 0.0);
        if (writer) c_fma_conj(ov, w, state);
        const double v = kh_q2h_writers_sum(u.mu_re * ov.y + u.mu_im * ov.x);
        if (lane == 0) red[par][wave] = v;
    };
    auto partial_total = [&](int par) {
        double acc = 0.0;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) acc += red[par][w8];
        return chi_norm * acc;
    };

    // interval 0: w = H1 chi(t_0) like every other one, from chib
    load_chi(0);
    if (writer) s.chib[row] = chi;
    __syncthreads();
    cplx w = kh_q2h_product(s, h1, s.chib, wave, lane, u.adj_sign);
    matvecs += 1.0;
    piece(0, w);
    if (1 < nt - 1) {
        load_chi(1);
        if (writer) s.chib[row] = chi;
    }
    __syncthreads();

    double dt_next = kh_uniform(p.dt[0]), guess_next = kh_uniform(u.guess[0]), shape_next = kh_uniform(u.shape[0]);
    const double lam = kh_uniform(u.lambda[0]);
    int m_rows = -1;
    double stepw_next = kh_uniform(shape_next / lam);
    KhDegreeCache dc = {12, 1.0, 0.0};
    double g_a_loc = 0.0;

    for (int nr = 0, n_stop; nr < nt - 1; nr = n_stop) {
        kh_q2h_load(pk[0], tid, lane, a);  // restart A = H0, B = P0 (see kh_q2_forward_update)
        kh_q2h_load(pk[2], tid, lane, b);
        eps_prev = 0.0;
        n_stop = (nr / KH_Q2_REFRESH + 1) * KH_Q2_REFRESH;
        n_stop = n_stop < nt - 1 ? n_stop : nt - 1;
        for (int n = nr; n < n_stop; ++n) {
            const int par = n & 1;
            // ---- cross-objective sum (optimize.py:470) ----
            double part[1] = {0.0};
            if (wave == 0) {
                part[0] = partial_total(par);
                kh_exchange_publish(ex, n, k, 1, lane, part);
            }
            // w = H1 chi(t_{n+1}) in the shadow of the exchange (chib holds chi(t_{n+1})): both stages, two barriers --
            // wave 0 takes part between its publication and its first poll
            if (n + 1 < nt - 1) {
                w = kh_q2h_product(s, h1, s.chib, wave, lane, u.adj_sign);
                matvecs += 1.0;
            }
            if (wave == 0) {
                double D[1];
                const bool ok = kh_exchange_collect<1, KH_GATHER_CHUNKS_WIDE, true>(ex, n, k, 1, lane, part, D);
                if (lane == 0) {
                    D_sh[par][0] = D[0];
                    D_sh[par][1] = ok ? 1.0 : 0.0;
                }
            }
            const double dt = dt_next, guess = guess_next, stepw = stepw_next;
            double dt_ld = 0.0, guess_ld = 0.0, shape_ld = 0.0;
            if (n + 1 < nt - 1) {
                dt_ld = p.dt[n + 1];
                guess_ld = u.guess[n + 1];
                shape_ld = u.shape[n + 1];
            }
            __syncthreads();
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
            if (m != m_rows) {
                kh_q2h_load_rows(p, s, m, tid);
                m_rows = m;
            }
            kh_q2h_advance(s, tid, lane, eps, eps_prev, h1, a, b);
            eps_prev = eps;
            stepw_next = kh_uniform(shape_next / lam);
            if (n + 2 < nt - 1) load_chi(n + 2);  // lands during the phases; goes to LDS in the epilogue
            auto epilogue = [&] {
                if (n + 1 < nt - 1) {
                    piece((n + 1) & 1, w);
                    if (n + 2 < nt - 1 && writer) s.chib[row] = chi;
                }
            };
            matvecs += kh_q2h_expm_action(s, a, b, state, cur, (cplx *)nullptr, N, p.fre, p.fim, dt, nsub, m, wave, lane,
                                          epilogue);
        }
    }
    if (wave == 0 && lane < N) u.phi[(size_t)k * N + lane] = s.buf[cur][lane];
    if (k == 0 && tid == 0) u.g_a[0] = g_a_loc;
    if (tid == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}
