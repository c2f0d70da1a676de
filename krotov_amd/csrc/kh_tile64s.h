// krotov_amd/csrc/kh_tile64s.h -- update sweep for MORE objectives than a GPU keeps co-resident (N <= 64)
//
// The register-tile kernels (kh_tile64.h, kh_tile64q2.h) keep the operators of ONE objective in the registers of a
// workgroup for the whole sweep; that bounds a GPU at one (two, with one control) objective per CU, and an ensemble of
// 1 000 members used to fall to one LAUNCH per interval (48 us per interval at K = 1024).  Here ONE persistent launch
// covers the sweep: G co-resident workgroups, workgroup w owns the objectives w, w + G, w + 2G, ... and walks through
// them in every interval -- the running states stay in LDS (1 KiB each), the operator tiles are STREAMED from the
// memory side for every objective and interval (128 KiB per objective with one control: at K = 1024 that is 128 MiB
// per interval, which the Infinity Cache holds), the workgroup's pieces of the cross-objective sums are added in a
// fixed order before they go through the same in-kernel exchange as everywhere else (kh_common.h).
//
//   optimize.py:444-508 (the forward sweep with sequential update) for K > #co-resident workgroups on one GPU.
//
// Where an interval goes (K = 1024, one control, 256 workgroups x 4 objectives, scripts/timing_stream.py): 23 us in the
// products (5.7 us per objective: 13 phases whose LDS reads -- every wave reads the whole vector, 64 KiB per phase and
// workgroup -- and FMAs do not overlap between barriers), 6 us waiting for tiles, 3.4 us in the exchange.
// Measured and not kept (docs/HISTORY.md R4.10): two workgroups per CU at 128 VGPRs, a second tile set in registers that
// fetches the next objective's tiles during the products, two objectives per phase in one workgroup, fetching Hermitian
// operators from their upper block triangle only.
#pragma once
#include "kh_tile64.h"

#define KH_STREAM_MMAX 16  // objectives per workgroup (LDS: 2 KiB each)
#ifndef KH_TIMING_WG
#define KH_TIMING_WG 0
#endif
#ifndef KH_TIMING_STRIDE
#define KH_TIMING_STRIDE 8
#endif

// kh_tile_load_op (RPT = 1) for a loader that runs for every objective and interval: no per-element branches.  (The
// guarded loads of kh_tile_load_op compile to one branch per element whose address reload from scratch waits for
// vmcnt(0): sixteen dependent round trips per tile set -- irrelevant where the tiles are fetched once per sweep.)
// N64 (N = 64, an instantiation of its own): one address per tile, the eight elements at immediate offsets; otherwise
// clamped indices, zeros selected.
template <bool N64>
__device__ __forceinline__ void kh_stream_load_op(const cplx *op, int N, int wave, int lane, cplx (&a)[1][8]) {
    const int cg = KhTileLanes::cg(lane), row = KhTile<1>::row_in(wave, lane, 0);
    if (op == nullptr) {  // (uniform: a control that this objective does not have)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[0][j] = c_make(0.0, 0.0);
        return;
    }
    if constexpr (N64) {
        const cplx *src = op + (unsigned)kh_launder(row * KH_TILE_N + cg);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[0][j] = src[8 * j];
    } else {
        const int rc = row < N ? row : N - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = cg + 8 * j, cc = col < N ? col : N - 1;
            const cplx v = op[(unsigned)(rc * N + cc)];
            const bool in = row < N && col < N;
            a[0][j] = c_make(in ? v.x : 0.0, in ? v.y : 0.0);
        }
    }
}

template <int LT, bool SO, bool N64>
__global__ void __launch_bounds__(512, 2)
kh_stream_forward_update(KhSweepArgs p, KhUpdateArgs u, KhExchange ex) {
    constexpr int WAVES = 8;
    typedef KhTileOps<1, LT, 0> Tiles;
    __shared__ __attribute__((aligned(16))) cplx bufs[KH_STREAM_MMAX][2][KH_TILE_N];
    __shared__ __attribute__((aligned(16))) double red[WAVES][LT];
    __shared__ __attribute__((aligned(16))) double D_sh[2][LT + 1];
    __shared__ __attribute__((aligned(16))) double ok_sh[2][LT];
    __shared__ __attribute__((aligned(16))) double inv_sh[KH_MAX_DEGREE + 2];
    __shared__ __attribute__((aligned(16))) double deg_sh[KH_MAX_DEGREE + 2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, cg = KhTileLanes::cg(lane);
    if (tid <= KH_MAX_DEGREE) deg_sh[tid] = p.q2_theta[tid];
    const int N = N64 ? KH_TILE_N : p.N, nt = p.nt, K = p.K;
    const int w = blockIdx.x, G = gridDim.x;
    const int mw = w < K ? (K - w + G - 1) / G : 0;  // this workgroup's objectives: w + j G, j < mw
    const int row = KhTile<1>::row(wave, lane, 0);
    double matvecs = 0.0;
    unsigned int curbits = 0;  // bit j: which of bufs[j][.] holds objective j's state

    for (int j = 0; j < mw; ++j) {
        const int k = w + j * G;
        if (tid < KH_TILE_N) bufs[j][0][tid] = tid < N ? u.phi[(size_t)k * N + tid] : c_make(0.0, 0.0);
    }
    __syncthreads();

    auto load_tiles = [&](Tiles &h, int k) {
        const cplx *const *ops_k = p.ops + (size_t)k * (1 + LT);
#pragma unroll
        for (int o = 0; o <= LT; ++o) {
            kh_stream_load_op<N64>(ops_k[o], N, wave, lane, h.reg[o]);
        }
    };
    // wave-level pieces of  Im(mu <bra | H_l phi>)  of one objective -> red[wave][l]; phi in x (LDS); chi (second order:
    // chi + hs (phi - phi_prev), optimize.py:468-469) replicated over the row's lanes -- kh_tile64.h, partial_pieces
    auto pieces = [&](const Tiles &h, const cplx *x, const cplx &bra) {
        cplx xv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) xv[c] = x[cg + 8 * c];
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            cplx y[1];
            h.matvec_part(1 + l, xv, y);
            cplx ov = c_make(0.0, 0.0);
            c_fma_conj(ov, bra, y[0]);
            const double v = sum64_mfma(u.mu_re * ov.y + u.mu_im * ov.x);
            if (lane == 0) red[wave][l] = v;
        }
        matvecs += LT;
    };
    // (after a barrier) part[l] += ||chi_k|| * the waves' pieces, in every thread
    auto add_total = [&](int k, double (&part)[LT]) {
        const double chi_norm = kh_uniform(u.chi_norms[k]);
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            double acc = 0.0;
#pragma unroll
            for (int ww = 0; ww < WAVES; ++ww) acc += red[ww][l];
            part[l] += chi_norm * acc;
        }
    };

    Tiles h;
    // the workgroup's pieces of the first interval's sums
    double part[LT], g_a_loc[LT], eps[LT];
#pragma unroll
    for (int l = 0; l < LT; ++l) part[l] = g_a_loc[l] = eps[l] = 0.0;
    if (u.n_begin < nt - 1) {
        for (int j = 0; j < mw; ++j) {
            const int k = w + j * G;
            load_tiles(h, k);
            cplx bra = row < N ? u.chi_store[((size_t)k * nt + u.n_begin) * N + row] : c_make(0.0, 0.0);
            if constexpr (SO) {
                const cplx prev = row < N ? u.fw_prev[((size_t)k * nt + u.n_begin) * N + row] : c_make(0.0, 0.0);
                const double hs = 0.5 * u.sigma[u.n_begin] / kh_uniform(u.chi_norms[k]);
                const cplx phi_row = bufs[j][0][row];
                bra = c_make(fma(hs, phi_row.x - prev.x, bra.x), fma(hs, phi_row.y - prev.y, bra.y));
            }
            pieces(h, bufs[j][0], bra);
            __syncthreads();
            add_total(k, part);
            __syncthreads();  // (red is free again)
        }
    }
    int m_loaded = -1;
    double dt = 0.0;

    // objective j of this workgroup over interval n
    auto step = [&](Tiles &hc, int n, int j) -> bool {
        const int k = w + j * G;
#ifdef KH_TIMING
        const long long tk0 = wall_clock64();
        long long tkg = tk0;
#endif
        if (j > 0) load_tiles(hc, k);
        if (j == 0) {
            const int par = n & 1;
            // ---- cross-objective sum (optimize.py:470) ----
            // The first objective's tiles are on their way while the sums are exchanged -- except in the waves that
            // gather: their polls would queue up behind the tiles (loads return in order).  Those fetch theirs behind it.
            constexpr int CH = KH_GATHER_CHUNKS;
            if (LT > 1) {
                if (wave == 0) kh_exchange_publish(ex, n, w, LT, lane, part);
                if (wave >= LT) load_tiles(hc, k);
                if (wave < LT) {
                    double Dl = 0.0;
                    bool ok = true;
                    if (ex.G > 1)
                        ok = kh_gather_one<CH>(ex, par, LT, wave, (unsigned)(n + 1), lane, Dl);
                    else {
#pragma unroll
                        for (int l = 0; l < LT; ++l) Dl = wave == l ? part[l] : Dl;
                    }
                    if (lane == 0) {
                        D_sh[par][wave] = Dl;
                        ok_sh[par][wave] = ok ? 1.0 : 0.0;
                    }
                    load_tiles(hc, k);
                }
            } else if (wave == 0) {
                double D[LT];
                const bool ok = kh_exchange<LT, CH, false>(ex, n, w, LT, lane, part, D);
                if (lane == 0) {
#pragma unroll
                    for (int l = 0; l < LT; ++l) {
                        D_sh[par][l] = D[l];
                        ok_sh[par][l] = ok ? 1.0 : 0.0;
                    }
                }
                load_tiles(hc, k);
            } else {
                load_tiles(hc, k);
            }
#ifdef KH_TIMING
            tkg = wall_clock64();
#endif
            dt = kh_uniform(p.dt[n]);
            double guess[LT], stp[LT];
#pragma unroll
            for (int l = 0; l < LT; ++l) {
                guess[l] = kh_uniform(u.guess[(size_t)l * (nt - 1) + n]);
                stp[l] = kh_uniform(u.shape[(size_t)l * (nt - 1) + n]) / kh_uniform(u.lambda[l]);
            }
            __syncthreads();
            bool all_ok = true;
#pragma unroll
            for (int l = 0; l < LT; ++l) all_ok = all_ok && ok_sh[par][l] != 0.0;
            if (!all_ok) return false;
            // ---- pulse update (optimize.py:471-477) ----
#pragma unroll
            for (int l = 0; l < LT; ++l) {
                const double d1 = D_sh[par][l];
                eps[l] = kh_uniform(guess[l] + stp[l] * d1);
                g_a_loc[l] = kh_uniform(g_a_loc[l] + stp[l] * (d1 * d1) * dt);
                part[l] = 0.0;
            }
            if (w == 0 && tid == 0) {
#pragma unroll
                for (int l = 0; l < LT; ++l) u.opt[(size_t)l * (nt - 1) + n] = eps[l];
            }
        }
#ifdef KH_TIMING
        const long long tk1 = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long tk2 = wall_clock64();
#endif
        // ---- objective k over interval n with the updated pulse (optimize.py:479-491) ----
        int cur = (curbits >> j) & 1;
        cplx bra_raw = c_make(0.0, 0.0), prev = c_make(0.0, 0.0);
        double hs = 0.0;
        if (n + 1 < nt - 1) {  // lands while the series runs
            bra_raw = row < N ? u.chi_store[((size_t)k * nt + n + 1) * N + row] : c_make(0.0, 0.0);
            if constexpr (SO) {
                prev = row < N ? u.fw_prev[((size_t)k * nt + n + 1) * N + row] : c_make(0.0, 0.0);
                hs = 0.5 * u.sigma[n + 1] / kh_uniform(u.chi_norms[k]);
            }
        }
        const double *norms_k = p.op_norms + (size_t)k * (1 + LT);
        double theta = kh_uniform(norms_k[0]);
#pragma unroll
        for (int l = 0; l < LT; ++l) theta += fabs(eps[l]) * kh_uniform(norms_k[1 + l]);
        int nsub, m;
        kh_degree_lookup(theta * dt, deg_sh, p.theta_max, p.inv_theta_max, m_loaded < 1 ? 12 : m_loaded, &nsub, &m);
        if (m != m_loaded) kh_tile_load_ratios(p, inv_sh, m, tid);  // (workgroup-uniform, rare)
        m_loaded = m;
        if constexpr (SO) {
            if (wave == 0 && lane < N) u.fw_store[((size_t)k * nt + n) * N + lane] = bufs[j][cur][lane];
        }
        cplx state[1];
        state[0] = bufs[j][cur][row];
        // the generator takes the drift's registers: H_0 is fetched again for the next interval anyway
        hc.build(eps, hc.reg[0]);
        matvecs += kh_tile_expm_action<1>(hc.reg[0], state, bufs[j], inv_sh, cur, p.fre, p.fim, dt, nsub, m, wave, lane);
        curbits = (curbits & ~(1u << j)) | ((unsigned)cur << j);
        if (n + 1 < nt - 1) {
            cplx bra = bra_raw;
            if constexpr (SO)
                bra = c_make(fma(hs, state[0].x - prev.x, bra_raw.x), fma(hs, state[0].y - prev.y, bra_raw.y));
            pieces(hc, bufs[j][cur], bra);
            __syncthreads();
            add_total(k, part);
        }
#ifdef KH_TIMING
        if (w == KH_TIMING_WG && tid == 0 && p.stats != nullptr) {  // 10-ns ticks: exchange (+ issue) | tiles' wait | products
            p.stats[1] += (double)(tkg - tk0);   // publish + gather (wave 0)
            p.stats[2] += (double)(tk2 - tkg);   // scalars, barrier, pulse update, tiles' issue and wait
            p.stats[3] += (double)(wall_clock64() - tk2);
        }
        // (KH_TRACE=1: 10-ns ticks that workgroups 0, KH_TIMING_STRIDE, ... spent outside the exchange, as differences to workgroup 0)
        if (w % KH_TIMING_STRIDE == 0 && w / KH_TIMING_STRIDE < 64 && tid == 0 && p.stats != nullptr)
            p.stats[4 + w / KH_TIMING_STRIDE] += (double)(wall_clock64() - tk1);
#endif
        return true;
    };

    if (mw > 0) {
        for (int n = u.n_begin; n < u.n_end; ++n)
            for (int j = 0; j < mw; ++j)
                if (!step(h, n, j)) return;
    }
    for (int j = 0; j < mw; ++j) {
        const int k = w + j * G;
        const int cur = (curbits >> j) & 1;
        if (wave == 0 && lane < N) {
            u.phi[(size_t)k * N + lane] = bufs[j][cur][lane];
            if constexpr (SO) u.fw_store[((size_t)k * nt + u.n_end) * N + lane] = bufs[j][cur][lane];
        }
    }
    if (w == 0 && tid == 0)
        for (int l = 0; l < LT; ++l) u.g_a[l] = g_a_loc[l];
    if (tid == 0 && p.stats != nullptr) atomicAdd(p.stats, matvecs);
}

