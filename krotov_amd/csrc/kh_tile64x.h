// krotov_amd/csrc/kh_tile64x.h -- register-tile sweeps for N <= 64 with FIVE TO EIGHT controls (round 6)
//
// kh_tile64.h keeps all 1 + L operator tiles of an objective on the CU (registers, two of them in LDS) and stops at four
// controls; with more the problem used to fall to the generic kernels, which re-assemble the interval's generator from
// 1 + L operators streamed from the memory side (576 KiB per objective and interval at L = 8: 25 of the backward sweep's
// 40 us per interval) and run the series from LDS at a quarter of the register-tile rate.  Here, as in kh_tile64.h, one
// 512-thread workgroup per objective holds the generator in registers (lane = row x column group, 8 elements) -- and of
// the operators as many as the CU has room for:
//   * H0 and the controls 0, 1, 2 in registers (4 x 32 VGPRs), the controls 3, 4 in LDS (2 x 64 KiB, each lane reads only
//     its own slots: conflict-free, no barrier), the controls 5 ... L - 1 STREAMED once per interval from lane-order,
//     zero-padded copies (kh_tx_permute at engine creation: a wave-level load is 1 KiB contiguous; 64 KiB per streamed
//     operator, objective and interval -- at L = 8, K = 256: 48 MiB per interval, which the Infinity Cache holds;
//     when they are fetched: kh_tx_build below);
//   * the update sums on the adjoint side (first order): <chi_k | H_lk phi_k> = <H_lk^+ chi_k | phi_k> with the left
//     factor formed for the whole co-state store in front of the sweep (kh_generic.h: kh_gen_adjoint_side, the generic
//     family's pre-pass, u.adj_store = [L][K][nt][N]) -- wave l takes control l: one dot product with the state the
//     series left in LDS, no operator product and no cross-wave reduction; the same wave gathers control l's sum from
//     the other workgroups (kh_common.h), so with L <= 8 = the workgroup's waves everything runs side by side;
//   * one instantiation per number of controls (LT = 5 .. 8): which tiles are streamed is static code.
//
//   optimize.py:444-508 (forward sweep with sequential update), :393-418 (backward sweep) for 5 <= L <= 8, N <= 64,
//   dense operators, at most one objective per CU.  Second order, more objectives than CUs, one launch per interval:
//   the generic kernels as before.
#pragma once
#include "kh_tile64.h"

#define KH_TX_THREADS 512
#define KH_TX_REG 4  // operator tiles in registers: H0 and the controls 0 .. 2
#define KH_TX_LDS 2  // operator tiles in LDS: the controls 3, 4
#define KH_TX_MIN_L 5

__host__ __device__ inline size_t kh_tx_lds_bytes() { return (size_t)KH_TX_LDS * KH_TILE_N * KH_TILE_N * sizeof(cplx); }

// row-major N x N -> lane order of the 512-thread register tile, zero beyond N:
// out[j * 512 + tid] = in[row 8 wave + (lane >> 3)][column (lane & 7) + 8 j], j < 8
__global__ void kh_tx_permute(const cplx *__restrict__ in, cplx *__restrict__ out, int N)
#if KH_DEFINES(KH_TU_MAIN)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row = KhTile<1>::row_in(wave, lane, 0), cg = KhTileLanes::cg(lane);
    for (int j = blockIdx.x; j < 8; j += gridDim.x) {
        const int col = cg + 8 * j;
        out[(size_t)j * KH_TX_THREADS + tid] = (row < N && col < N) ? in[(size_t)row * N + col] : c_make(0.0, 0.0);
    }
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif

// this lane's 8 elements of an operator in lane order (a control the objective does not have: the engine's zero tile).
// The tile pointers come out of a table in memory: without the address-space cast these are FLAT loads (which also count
// as LDS operations and need a 64-bit address register pair each).
typedef double kh_tx_d2 __attribute__((ext_vector_type(2)));
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(1))) kh_tx_d2 *kh_tx_gptr;
#else
typedef const kh_tx_d2 *kh_tx_gptr;  // (the host pass only parses the kernels)
#endif
__device__ __forceinline__ void kh_tx_load(const cplx *tab, int tid, cplx (&a)[8]) {
    const kh_tx_gptr g = (kh_tx_gptr)tab + (unsigned)kh_launder(tid);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const kh_tx_d2 v = g[j * KH_TX_THREADS];
        a[j] = c_make(v.x, v.y);
    }
}
// ... of a STREAMED operator: a lane whose row (tid >> 3) is beyond N holds zeros and does not fetch them
__device__ __forceinline__ void kh_tx_load_rows(const cplx *tab, int tid, int N, cplx (&a)[8]) {
    if ((tid >> 3) < N) {
        kh_tx_load(tab, tid, a);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = c_make(0.0, 0.0);
    }
}

struct KhTxOps {
    cplx reg[KH_TX_REG][8];
    cplx *lds;  // this lane's column of the LDS-resident tiles: element (i, j) at lds[(i * 8 + j) * 512]

    __device__ __forceinline__ void load(const cplx *const *tab_k, int tid, cplx *lds_base) {
        lds = lds_base + tid;
#pragma unroll
        for (int o = 0; o < KH_TX_REG; ++o) kh_tx_load(tab_k[o], tid, reg[o]);
#pragma unroll
        for (int i = 0; i < KH_TX_LDS; ++i) {
            const kh_tx_gptr src = (kh_tx_gptr)tab_k[KH_TX_REG + i] + tid;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const kh_tx_d2 v = src[j * KH_TX_THREADS];
                lds[(size_t)(i * 8 + j) * KH_TX_THREADS] = c_make(v.x, v.y);
            }
        }
    }
    // a = H0 + sum_{l < 5} eps_l H_l (the resident operators) [+ e5 t5]; two elements at a time: all sixteen LDS reads
    // in flight together would cost 64 registers that the kernel does not have
    template <int NPRE>  // + e5 t5 (NPRE >= 1) + e6 t6 (NPRE == 2): the tiles fetched ahead
    __device__ __forceinline__ void build_resident(const double *eps, const cplx (&t5)[8], const cplx (&t6)[8], cplx (&a)[1][8]) const {
        double e[KH_TX_REG - 1 + KH_TX_LDS + 2];
#pragma unroll
        for (int l = 0; l < KH_TX_REG - 1 + KH_TX_LDS + NPRE; ++l) e[l] = kh_uniform(eps[l]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            cplx v = reg[0][j];
#pragma unroll
            for (int l = 0; l < KH_TX_REG - 1; ++l) {
                v.x = fma(e[l], reg[1 + l][j].x, v.x);
                v.y = fma(e[l], reg[1 + l][j].y, v.y);
            }
#pragma unroll
            for (int i = 0; i < KH_TX_LDS; ++i) {
                const cplx hl = lds[(size_t)(i * 8 + j) * KH_TX_THREADS];
                v.x = fma(e[KH_TX_REG - 1 + i], hl.x, v.x);
                v.y = fma(e[KH_TX_REG - 1 + i], hl.y, v.y);
            }
            if constexpr (NPRE >= 1) {
                v.x = fma(e[5], t5[j].x, v.x);
                v.y = fma(e[5], t5[j].y, v.y);
            }
            if constexpr (NPRE >= 2) {
                v.x = fma(e[6], t6[j].x, v.x);
                v.y = fma(e[6], t6[j].y, v.y);
            }
            asm volatile("" : "+v"(v.x), "+v"(v.y));  // (see kh_tx_pin)
            a[0][j] = v;
            if (j & 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
};

// The generator is used only inside the series' loops (at least one sub-step of at least one term, but the compiler does
// not know that): left alone, its multiply-adds are SUNK into that branch, behind all the loads they consume -- 24 global and
// 16 LDS loads in flight at once, 160 registers, the resident tiles in scratch.  An empty asm that "modifies" the generator
// pins its assembly where it is written.
__device__ __forceinline__ void kh_tx_pin(cplx (&a)[1][8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(a[0][j].x), "+v"(a[0][j].y));
}

// a += w t
__device__ __forceinline__ void kh_tx_axpy(double w, const cplx (&t)[8], cplx (&a)[1][8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[0][j].x = fma(w, t[j].x, a[0][j].x);
        a[0][j].y = fma(w, t[j].y, a[0][j].y);
    }
}

// The streamed controls 5, 6, 7.  Register budget: four resident tiles + the generator + the series' broadcast vector leave
// no room for a tile that rides through the series: the tiles are fetched at the top of a step (plain sweeps: two of them)
// or one after the other while the generator is assembled.
// NPRE of the streamed tiles (controls 5, 6) are fetched at the top of the interval, the others while the generator is
// assembled.  Measured (profiles/r06/ab_tile64x.txt, K = 256, N = 64): the streamed tiles are a burst that all workgroups
// issue at the same moment -- the update sweep re-synchronises them every interval --, 16 MiB per streamed control, and
// what an interval costs on top of the five-control kernel is that burst's transfer time (+2.3 / +4.3 / +6.1 us for the
// first / second / third streamed control: the first still comes out of the XCDs' L2); fetching ahead of the exchange
// does not hide it (the gathering waves' polls queue up behind the tiles: loads return in order) and costs +4.5 us at
// L = 7.  The plain sweeps' workgroups drift apart and stream at the memory side's rate (48 MiB in 8.3 us at L = 8).
#ifndef KH_TX_STORE_PRE
#define KH_TX_STORE_PRE 2
#endif
#ifndef KH_TX_UPDATE_PRE
#define KH_TX_UPDATE_PRE 0
#endif
template <int LT, int NPRE>
__device__ __forceinline__ void kh_tx_prefetch(const cplx *const *tab_k, int tid, int N, cplx (&ts0)[8], cplx (&ts1)[8]) {
    static_assert(LT >= KH_TX_MIN_L && LT <= 8 && KH_MAX_L == 8, "at most three streamed controls");
    if constexpr (LT > 5 && NPRE >= 1) kh_tx_load_rows(tab_k[6], tid, N, ts0);
    if constexpr (LT > 6 && NPRE >= 2) kh_tx_load_rows(tab_k[7], tid, N, ts1);
}
template <int LT, int NPRE>
__device__ __forceinline__ void kh_tx_build(const KhTxOps &h, const cplx *const *tab_k, const double *eps, int tid, int N,
                                            cplx (&ts0)[8], cplx (&ts1)[8], cplx (&a)[1][8]) {
    constexpr int HAVE = (LT - 5) < NPRE ? (LT - 5) : NPRE;  // tiles that are on their way already
    h.template build_resident<HAVE>(eps, ts0, ts1, a);
#pragma unroll
    for (int l = 5 + HAVE; l < LT; ++l) {
        __builtin_amdgcn_sched_barrier(0);
        kh_tx_load_rows(tab_k[1 + l], tid, N, ts0);
        kh_tx_axpy(kh_uniform(eps[l]), ts0, a);
        kh_tx_pin(a);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------
// plain propagation with storage (backward sweep / iteration-0 forward sweep); objectives in turns
// ---------------------------------------------------------------------------
// tabs: [K * (1 + L)] lane-order copies of this direction's operators
template <int LT>
__global__ void __launch_bounds__(KH_TX_THREADS)
kh_tx_sweep_store(KhSweepArgs p, const cplx *const *__restrict__ tabs, const double *__restrict__ pulses,
                  const cplx *__restrict__ state_in, cplx *__restrict__ store, cplx *__restrict__ state_out, int direction) {
    __shared__ __attribute__((aligned(16))) cplx buf[2][KH_TILE_N];
    __shared__ __attribute__((aligned(16))) double inv_sh[KH_MAX_DEGREE + 2];
    __shared__ __attribute__((aligned(16))) double deg_sh[KH_MAX_DEGREE + 2];
    __shared__ __attribute__((aligned(16))) double eps_sh[2][KH_MAX_L];  // by step parity
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool writer = KhTile<1>::writer(lane);
    if (tid <= KH_MAX_DEGREE) deg_sh[tid] = p.q2_theta[tid];
    const int N = p.N, nt = p.nt;
    constexpr int L = LT;
    const int row = KhTile<1>::row(wave, lane, 0);
    double matvecs = 0.0;
    for (int k = blockIdx.x; k < p.K; k += gridDim.x) {
        const cplx *const *tab_k = tabs + (size_t)k * (1 + L);
        double nrm[1 + LT];
#pragma unroll
        for (int o = 0; o <= LT; ++o) nrm[o] = kh_uniform(p.op_norms[(size_t)k * (1 + L) + o]);
        __syncthreads();  // previous objective's readers are done with buf, eps_sh and the LDS tiles
        KhTxOps h;
        h.load(tab_k, tid, kh_tile_dyn_lds);
        cplx state[1];
        state[0] = row < N ? state_in[(size_t)k * N + row] : c_make(0.0, 0.0);
        int cur = 0;
        if (writer) buf[0][row] = state[0];
        const int n0 = direction > 0 ? 0 : nt - 2;
        if (tid < L) eps_sh[0][tid] = pulses[(size_t)tid * (nt - 1) + n0];
        __syncthreads();
        if (store != nullptr && wave == 0 && lane < N)
            store[((size_t)k * nt + (direction > 0 ? 0 : nt - 1)) * N + lane] = buf[0][lane];
        double dt_next = kh_uniform(p.dt[n0]);
        int m_hint = -1;
        for (int step = 0; step < nt - 1; ++step) {
            cplx ts0[8], ts1[8];
            kh_tx_prefetch<LT, KH_TX_STORE_PRE>(tab_k, tid, N, ts0, ts1);
            const int n = direction > 0 ? step : nt - 2 - step;
            const int sp = step & 1;
            const double dt = dt_next;
            double theta = nrm[0];
#pragma unroll
            for (int l = 0; l < LT; ++l) theta += fabs(kh_uniform(eps_sh[sp][l])) * nrm[1 + l];
            if (step + 1 < nt - 1) {  // the next step's scalars: in flight during this one
                const int nn = direction > 0 ? n + 1 : n - 1;
                dt_next = kh_uniform(p.dt[nn]);
                if (tid < L) eps_sh[sp ^ 1][tid] = pulses[(size_t)tid * (nt - 1) + nn];
            }
            int nsub, m;
            kh_degree_lookup(theta * dt, deg_sh, p.theta_max, p.inv_theta_max, m_hint < 1 ? 12 : m_hint, &nsub, &m);
            if (m != m_hint) kh_tile_load_ratios(p, inv_sh, m, tid);  // (workgroup-uniform, rare)
            m_hint = m;
            cplx a[1][8];
            kh_tx_build<LT, KH_TX_STORE_PRE>(h, tab_k, eps_sh[sp], tid, N, ts0, ts1, a);
            matvecs += kh_tile_expm_action<1>(a, state, buf, inv_sh, cur, p.fre, p.fim, dt, nsub, m, wave, lane);
            // buf[cur] now holds the new state (and the barrier that published it also published eps_sh[sp ^ 1])
            if (store != nullptr && wave == 0 && lane < N)
                store[((size_t)k * nt + (direction > 0 ? n + 1 : n)) * N + lane] = buf[cur][lane];
        }
        if (state_out != nullptr && wave == 0 && lane < N) state_out[(size_t)k * N + lane] = buf[cur][lane];
    }
    if (tid == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}

// ---------------------------------------------------------------------------
// forward sweep with sequential pulse update (optimize.py:444-508), first order: ONE launch, grid == K <= #CUs,
// u.adj_store = H_lk^+ chi_k(t_n) for every control, objective and time (kh_gen_adjoint_side)
// ---------------------------------------------------------------------------
template <int LT>
__global__ void __launch_bounds__(KH_TX_THREADS)
kh_tx_forward_update(KhSweepArgs p, const cplx *const *__restrict__ tabs, KhUpdateArgs u, KhExchange ex) {
    __shared__ __attribute__((aligned(16))) cplx buf[2][KH_TILE_N];
    __shared__ __attribute__((aligned(16))) double inv_sh[KH_MAX_DEGREE + 2];
    __shared__ __attribute__((aligned(16))) double deg_sh[KH_MAX_DEGREE + 2];
    __shared__ __attribute__((aligned(16))) double red[KH_MAX_L];  // wave l's sum for control l
    __shared__ __attribute__((aligned(16))) double D_sh[KH_MAX_L], ok_sh[KH_MAX_L], eps_sh[KH_MAX_L], g_a_sh[KH_MAX_L];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool writer = KhTile<1>::writer(lane);
    const int N = p.N, nt = p.nt;
    constexpr int L = LT;
    const int k = blockIdx.x;
    const int row = KhTile<1>::row(wave, lane, 0);
    const cplx *const *tab_k = tabs + (size_t)k * (1 + L);
    double nrm[1 + LT];
#pragma unroll
    for (int o = 0; o <= LT; ++o) nrm[o] = kh_uniform(p.op_norms[(size_t)k * (1 + L) + o]);
    const double chi_norm = kh_uniform(u.chi_norms[k]);
    if (tid <= KH_MAX_DEGREE) deg_sh[tid] = p.q2_theta[tid];
    if (tid < KH_MAX_L) g_a_sh[tid] = 0.0;
    KhTxOps h;
    h.load(tab_k, tid, kh_tile_dyn_lds);
    cplx state[1];
    state[0] = row < N ? u.phi[(size_t)k * N + row] : c_make(0.0, 0.0);
    int cur = 0;
    if (writer) buf[0][row] = state[0];
    __syncthreads();
    double matvecs = 0.0;

    // wave l < L, lane = row: V_lk(t_n)[row] = (H_lk^+ chi_k(t_n))[row], fetched one interval ahead
    const bool has_control = wave < L && p.ops[(size_t)k * (1 + L) + 1 + (wave < L ? wave : 0)] != nullptr;
    cplx vb = c_make(0.0, 0.0);
    auto load_bra = [&](int n) {
        vb = (has_control && lane < N) ? u.adj_store[(((size_t)wave * p.K + k) * nt + n) * N + lane] : c_make(0.0, 0.0);
    };
    // ||chi_k|| Im(mu <V_lk(t_n) | phi_k(t_n)>) -> red[l]; phi in buf[cur] (all rows written, barrier passed)
    auto sums = [&]() {
        if (wave < L) {
            const cplx phi = buf[cur][lane];
            cplx ov = c_make(0.0, 0.0);
            c_fma_conj(ov, vb, phi);
            const double v = sum64_mfma(u.mu_re * ov.y + u.mu_im * ov.x);
            if (lane == 0) red[wave] = chi_norm * v;
        }
        matvecs += (double)L;
    };
    if (u.n_begin < nt - 1) {
        load_bra(u.n_begin);
        sums();
    }
    __syncthreads();
    int m_hint = -1;
    const double my_lambda = tid < LT ? u.lambda[tid] : 1.0;

    for (int n = u.n_begin; n < u.n_end; ++n) {
        const int par = n & 1;
        cplx ts0[8], ts1[8];
        kh_tx_prefetch<LT, KH_TX_UPDATE_PRE>(tab_k, tid, N, ts0, ts1);
        double my_guess = 0.0, my_stepw = 0.0;  // (in flight while the sums are exchanged)
        if (tid < LT) {
            my_guess = u.guess[(size_t)tid * (nt - 1) + n];
            my_stepw = u.shape[(size_t)tid * (nt - 1) + n] / my_lambda;
        }
        if (n + 1 < nt - 1) load_bra(n + 1);  // lands while this interval is processed
        // ---- cross-objective sum (optimize.py:470): wave 0 publishes, wave l gathers control l ----
        if (wave == 0) {
            double part[KH_MAX_L];
#pragma unroll
            for (int l = 0; l < KH_MAX_L; ++l) part[l] = l < L ? red[l] : 0.0;
            if (ex.G == 1) {
                if (lane == 0)
                    for (int l = 0; l < L; ++l) {
                        D_sh[l] = part[l];
                        ok_sh[l] = 1.0;
                    }
            }
            if (ex.G > 1) kh_publish(ex, par, k, L, lane, part, (unsigned)(n + 1));
        }
        if (ex.G > 1 && wave < L) {
            double Dl = 0.0;
            const bool ok = kh_gather_one<KH_GATHER_CHUNKS>(ex, par, L, wave, (unsigned)(n + 1), lane, Dl);
            if (lane == 0) {
                D_sh[wave] = Dl;
                ok_sh[wave] = ok ? 1.0 : 0.0;
            }
        }
        const double dt = kh_uniform(p.dt[n]);
        __syncthreads();
        if (ex.world > 1) {  // objectives sharded over GPUs: the GPUs' sums through the peer windows
            if (wave == 0) {
                double D[KH_MAX_L];
                bool ok = true;
#pragma unroll
                for (int l = 0; l < KH_MAX_L; ++l) {
                    D[l] = l < L ? D_sh[l] : 0.0;
                    ok = ok && (l >= L || ok_sh[l] != 0.0);
                }
                const unsigned int epoch = ex.epoch_base + (unsigned)(n + 1);
                if (ok) {
                    if (k == 0 && n != ex.fail_at) kh_p2p_publish(ex, par, L, lane, D, epoch);
                    ok = kh_p2p_gather<KH_MAX_L>(ex, par, L, epoch, lane, D);
                }
                if (lane == 0)
                    for (int l = 0; l < L; ++l) {
                        D_sh[l] = D[l];
                        ok_sh[l] = ok ? 1.0 : 0.0;
                    }
            }
            __syncthreads();
        }
        {
            bool all_ok = true;
            for (int l = 0; l < L; ++l) all_ok = all_ok && ok_sh[l] != 0.0;
            if (!all_ok) return;
        }
        // ---- pulse update (optimize.py:471-477): thread l = control l, its guess and step width fetched before the exchange ----
        if (tid < LT) {
            const double d1 = D_sh[tid];
            const double eps = my_guess + my_stepw * d1;
            eps_sh[tid] = eps;
            g_a_sh[tid] += my_stepw * (d1 * d1) * dt;
            if (k == 0) u.opt[(size_t)tid * (nt - 1) + n] = eps;
        }
        __syncthreads();  // eps_sh
        // ---- propagate over interval n with the updated pulses (optimize.py:479-491) ----
        double theta = nrm[0];
#pragma unroll
        for (int l = 0; l < LT; ++l) theta += fabs(kh_uniform(eps_sh[l])) * nrm[1 + l];
        int nsub, m;
        kh_degree_lookup(theta * dt, deg_sh, p.theta_max, p.inv_theta_max, m_hint < 1 ? 12 : m_hint, &nsub, &m);
        if (m != m_hint) kh_tile_load_ratios(p, inv_sh, m, tid);  // (workgroup-uniform, rare)
        m_hint = m;
        cplx a[1][8];
        kh_tx_build<LT, KH_TX_UPDATE_PRE>(h, tab_k, eps_sh, tid, N, ts0, ts1, a);
        matvecs += kh_tile_expm_action<1>(a, state, buf, inv_sh, cur, p.fre, p.fim, dt, nsub, m, wave, lane);
        // ---- partial sums of the next interval (state is in buf[cur], barrier passed) ----
        if (n + 1 < nt - 1) {
            sums();
            __syncthreads();  // every wave's sum is in red[] before wave 0 publishes it; D_sh / ok_sh / eps_sh are free
        }
    }
    // running state back to the engine workspace
    if (wave == 0 && lane < N) u.phi[(size_t)k * N + lane] = buf[cur][lane];
    if (k == 0 && tid < L) u.g_a[tid] = g_a_sh[tid];
    if (tid == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}
