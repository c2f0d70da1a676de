// Cooperative shared-operator kernels, second generation (round 3): rows over waves instead of k over waves.
//
// kh_coop.h splits the k range of a workgroup's 16 rows over 8 waves; every round (one term of the series for all
// objectives of the column group) then ends in a cross-wave sum: DPP sums -> LDS write -> barrier -> LDS read ->
// 8-lane sum -> owner arithmetic -> publication, ~2 200 cycles of a 4 450-cycle round with no memory operation and no
// matrix-core instruction in them.  Here a workgroup has FOUR waves (one per SIMD, so up to 512
// registers each) and wave w owns the row block [16 g + 4 w, 16 g + 4 w + 4) for the WHOLE k range:
//   * a round needs no LDS and no barrier: the wave fetches the column group's block (N x 2 objectives, 1 KiB per
//     16-row group, the consumer's lane order of kh_coop.h), multiplies -- one v_mfma_f64_4x4x4_4b per group and
//     operand part, the four blocks of the instruction taking four consecutive k-steps --, adds the four blocks with
//     two row rotations, and every lane then holds ONE real component (row lane >> 4, column lane & 3 of
//     [re c0, re c1, im c0, im c1]) of the result, four times replicated;
//   * the series arithmetic is real and component-wise (the coefficients are real), so the lanes carry the state, the
//     sum s and the published term as one double each; only the product with f = -+i at the end of a step pairs a
//     lane with its re/im partner (one DPP move);
//   * a lane publishes its component with ONE 16-byte store ({tag | hi}, {tag | lo}) into the ring of kh_coop.h;
//   * B = A^2 lives in registers for the whole k range (N / 16 complex numbers per lane: 100 VGPRs at N = 400), A in
//     LDS (read back only by the lane that wrote it), both advanced from interval to interval as in kh_coop.h.
// The price: every wave fetches the whole block (4 x the L1 -> register traffic of the k-split form).
// Two objectives per workgroup, one control, the A^2 chain; plain sweeps only.
//
// RESULT (MI355X, config 4, backward sweep; -DKH_WITH_C4W build, KH_COOP4W=1; parity tests of the plain sweeps pass):
// 35.1 ms against 22.9 ms with kh_coop.h (29.2 ms with a head start of 28 x 64 cycles before a round's fetch).  Per round
// (-DKH_TIMING, scripts/timing_c4.py): fetch 4 428 cycles (62 % of the rounds find a stale block and poll on), matrix
// cores + block sums 1 300, between rounds 2 190.  A quarter of the fetch (-DKH_C4_X_QUARTER) saves 2.5 ms, no matrix-core
// instructions (-DKH_C4_X_NOMFMA) 3.2 ms: neither is what the round waits for.  With ONE wave per SIMD nothing overlaps:
// a round's ~700 instructions (26 loads, tag checks, operand moves out of the accumulation registers -- the fragment
// does not fit next to the fetched block in 256 architectural VGPRs --, 52 MFMAs, sums, publication) issue one at a time,
// 4+ cycles each, and the exchange's round trip is fully exposed behind them; the k-split kernels hide one wave's
// latency behind the other seven.  First version: 57.8 ms -- a short-circuit `&&` chain over the 52 tag compares had
// compiled to 26 nested branches with exec-mask saves spilled to VGPR lanes; tag checks are bitwise since.
// Not part of the product library.
#pragma once
// (included from krotov_hip.hip in a -DKH_WITH_C4W build, after krotov_amd/csrc/kh_coop.h)

#define KH_C4_THREADS 256
#define KH_C4_WAVES 4
#define KH_C4_TABLE_PAD 272

__host__ __device__ inline int kh_c4_groups(int N) { return (N + 15) / 16; }
__host__ __device__ inline size_t kh_c4_table_stride(int NG) { return (size_t)KH_C4_WAVES * NG * 64 + KH_C4_TABLE_PAD; }
__host__ __device__ inline size_t kh_c4_table_elems(int G, int NG) { return (size_t)G * kh_c4_table_stride(NG); }

// fragment order: element of (row block g, wave w, group j, lane) = F[16 g + 4 w + (lane & 3)][16 j + 4 ((lane >> 2) & 3)
// + (lane >> 4)] -- the A operand of v_mfma_f64_4x4x4_4b whose blocks are the group's four k-steps -- at
// [g stride + (w NG + j) 64 + lane], zero beyond N
__global__ void kh_c4_permute_kernel(const cplx *__restrict__ in, cplx *__restrict__ out, int N, int G, int NG) {
    const size_t total = (size_t)G * KH_C4_WAVES * NG * 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const size_t t = idx >> 6;
        const int j = (int)(t % NG), w = (int)((t / NG) % KH_C4_WAVES), g = (int)(t / NG / KH_C4_WAVES);
        const int row = 16 * g + 4 * w + (lane & 3), col = 16 * j + 4 * ((lane >> 2) & 3) + (lane >> 4);
        out[(size_t)g * kh_c4_table_stride(NG) + (size_t)(w * NG + j) * 64 + lane] =
            (row < N && col < N) ? in[(size_t)row * N + col] : c_make(0.0, 0.0);
    }
}
// bit j of word (g, w) behind the table: group j of that wave's fragment has a non-zero element
__global__ void kh_c4_mask_kernel(const cplx *__restrict__ tab, unsigned int *__restrict__ mask, int NG) {
    const int lane = threadIdx.x;  // one wave per (row block, wave)
    const cplx *src = tab + (size_t)(blockIdx.x / KH_C4_WAVES) * kh_c4_table_stride(NG) +
                      (size_t)(blockIdx.x % KH_C4_WAVES) * NG * 64 + lane;
    unsigned int m = 0;
    for (int j = 0; j < NG; ++j) {
        const cplx v = src[(size_t)j * 64];
        if (__ballot(v.x != 0.0 || v.y != 0.0) != 0ull) m |= 1u << j;
    }
    if (lane == 0) mask[blockIdx.x] = m;
}
__device__ __forceinline__ unsigned int kh_c4_frag_mask(const cplx *op, int G, int g, int wave, int NG) {
    if (op == nullptr) return 0u;
    const unsigned int *m = (const unsigned int *)(op + kh_c4_table_elems(G, NG));
    return __builtin_amdgcn_readfirstlane(m[g * KH_C4_WAVES + wave]);
}

struct KhC4Lds {
    double red[2][KH_C4_WAVES];  // the waves' pieces of the update sum, by interval parity
    double D[2][2];              // reduced sum + ok flag, by interval parity
    double deg[KH_MAX_DEGREE + 2];
    double coef[2][2 * KH_Q2_ROWS + 2];  // the interval's series rows, by interval parity: the rounds have no barrier, so a
                                         // wave may enter the next interval while another still reads this one's rows
    int abort;
    int local;
#ifdef KH_TIMING
    double tim[8];  // wave 0 of workgroup 0: [0] fetch (issue -> all fresh), [1] matrix cores + block sums, [2] between rounds, [3] stamp, [4] slow-path entries
#endif
    __attribute__((aligned(16))) double frag[1];  // [KH_C4_WAVES][NG][64] complex: the A fragment (dynamic size)
};
// (the fragment is laid out for MAXG groups per wave -- the kernels' template parameter -- so that no access needs a
// bounds check: groups beyond the real ones hold zeros)
__host__ __device__ inline size_t kh_c4_lds_bytes(int MAXG) {
    return sizeof(KhC4Lds) + sizeof(cplx) * (size_t)KH_C4_WAVES * MAXG * 64;
}

struct KhC4Masks {
    unsigned int h0, h1, p0, p1, p2;
};
__device__ __forceinline__ KhC4Masks kh_c4_masks(const KhCoopArgs &c, int g, int wave, int NG) {
    KhC4Masks m;
    m.h0 = kh_c4_frag_mask(c.fops[0], c.G, g, wave, NG);
    m.h1 = kh_c4_frag_mask(c.fops[1], c.G, g, wave, NG);
    m.p0 = kh_c4_frag_mask(c.sq[0], c.G, g, wave, NG);
    m.p1 = kh_c4_frag_mask(c.sq[1], c.G, g, wave, NG);
    m.p2 = kh_c4_frag_mask(c.sq[2], c.G, g, wave, NG);
    return m;
}
__device__ __forceinline__ KhCoopSrc kh_c4_src(const cplx *op, int g, int wave, int lane, int NG) {
    KhCoopSrc r;
    r.p = op == nullptr ? nullptr
                        : (const __attribute__((address_space(1))) kh_d2 *)(op + (size_t)g * kh_c4_table_stride(NG) +
                                                                            (size_t)wave * NG * 64 + lane);
    return r;
}

#define KH_C4_CHUNK 8
// r (+)= e * (fragment-ordered table), groups in chunks of KH_C4_CHUNK: all loads of a chunk before its first use, a
// chunk without a non-zero group is skipped
template <int MAXG, bool ASSIGN>
__device__ __forceinline__ void kh_c4_reg_axpy(const cplx *op, double e, int g, int wave, int lane, int NG,
                                               cplx (&r)[MAXG], unsigned int mask) {
    const KhCoopSrc src = kh_c4_src(op, g, wave, lane, NG);
#pragma unroll
    for (int c0 = 0; c0 < MAXG; c0 += KH_C4_CHUNK) {
        const bool live = !src.null() && ((mask >> c0) & ((1u << KH_C4_CHUNK) - 1u)) != 0u;  // (mask bits >= NG are 0)
        if (live) {
            cplx v[KH_C4_CHUNK];
#pragma unroll
            for (int j = 0; j < KH_C4_CHUNK; ++j) v[j] = src[(size_t)(c0 + j < NG ? c0 + j : NG - 1) * 64];
#pragma unroll
            for (int j = 0; j < KH_C4_CHUNK; ++j) {
                if (c0 + j < MAXG) {
                    const double ej = c0 + j < NG ? e : 0.0;  // (a clamped read beyond the last group adds nothing)
                    if constexpr (ASSIGN) {
                        r[c0 + j] = c_make(ej * v[j].x, ej * v[j].y);
                    } else {
                        r[c0 + j].x = fma(ej, v[j].x, r[c0 + j].x);
                        r[c0 + j].y = fma(ej, v[j].y, r[c0 + j].y);
                    }
                }
            }
        } else if constexpr (ASSIGN) {
#pragma unroll
            for (int j = 0; j < KH_C4_CHUNK; ++j)
                if (c0 + j < MAXG) r[c0 + j] = c_make(0.0, 0.0);
        }
    }
}
// the same for the LDS fragment (a: this lane's element of group j at a[j * 64]; MAXG groups per wave)
template <int MAXG, bool ASSIGN>
__device__ __forceinline__ void kh_c4_lds_axpy(const cplx *op, double e, int g, int wave, int lane, int NG, cplx *a,
                                               unsigned int mask) {
    const KhCoopSrc src = kh_c4_src(op, g, wave, lane, NG);
#pragma unroll
    for (int c0 = 0; c0 < MAXG; c0 += KH_C4_CHUNK) {
        const bool live = !src.null() && ((mask >> c0) & ((1u << KH_C4_CHUNK) - 1u)) != 0u;
        if (live) {
            cplx v[KH_C4_CHUNK], t[KH_C4_CHUNK];
#pragma unroll
            for (int j = 0; j < KH_C4_CHUNK; ++j) {
                if (c0 + j < MAXG) {
                    v[j] = src[(size_t)(c0 + j < NG ? c0 + j : NG - 1) * 64];
                    t[j] = ASSIGN ? c_make(0.0, 0.0) : a[(size_t)(c0 + j) * 64];
                }
            }
#pragma unroll
            for (int j = 0; j < KH_C4_CHUNK; ++j) {
                if (c0 + j < MAXG) {
                    const double ej = c0 + j < NG ? e : 0.0;
                    t[j].x = fma(ej, v[j].x, t[j].x);
                    t[j].y = fma(ej, v[j].y, t[j].y);
                    a[(size_t)(c0 + j) * 64] = t[j];
                }
            }
        } else if constexpr (ASSIGN) {
#pragma unroll
            for (int j = 0; j < KH_C4_CHUNK; ++j)
                if (c0 + j < MAXG) a[(size_t)(c0 + j) * 64] = c_make(0.0, 0.0);
        }
    }
}

typedef unsigned int kh_u32x4s __attribute__((ext_vector_type(4)));
// this lane's component of round `rid`: one 16-byte store (two tagged granules) at the consumer's lane position
__device__ __forceinline__ void kh_c4_publish(const KhCoopArgs &c, __amdgpu_buffer_rsrc_t rsrc, unsigned int rid, int y, int g,
                                              int wave, int lane, double value) {
    const unsigned int tag = c.epoch_base + rid;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(value);
    kh_u32x4s v;
    v.x = (unsigned int)(bits >> 32);
    v.y = tag;
    v.z = (unsigned int)(bits & 0xffffffffull);
    v.w = tag;
    const unsigned int off = (unsigned int)((((size_t)(rid % KH_COOP_RING) * c.Y + y) * (size_t)c.G + g) * 1024u) +
                             16u * (unsigned int)(16 * (lane >> 4) + 4 * wave + (lane & 3));
    if (c.local)
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)off, 0, KH_CPOL_SC0);
    else
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)off, 0, KH_CPOL_SC1);
}

// One round for this wave's four rows.  `e` <- the lane's real component (row lane >> 4, column lane & 3 of
// [re c0, re c1, im c0, im c1]) of  F . T_rid, replicated over the four lanes (lane >> 2) & 3.  F = breg (registers) or,
// with FROM_LDS, the fragment in LDS.  Returns false if the exchange timed out (abort flag raised).
template <int MAXG, bool FROM_LDS>
__device__ __forceinline__ bool kh_c4_round(const KhCoopArgs &c, const KhExchange &ex, __amdgpu_buffer_rsrc_t rsrc,
                                            unsigned int rid, int y, int jmax, const cplx (&breg)[MAXG],
                                            const cplx *afrag, KhC4Lds &s, int lane, double &e) {
    // jmax (per lane): groups j < jmax hold a row < N for this lane's position in the group.  Everything below is
    // unconditional over the MAXG groups -- a group beyond jmax is fetched from beyond the buffer (the load returns
    // zeros without touching memory), counts as fresh and multiplies zeros: no branch, no exec juggling per group.
    const unsigned int epoch = c.epoch_base + rid;
    const unsigned int ring_off = (unsigned int)((((size_t)(rid % KH_COOP_RING) * c.Y + y) * (size_t)c.G) * 1024u) + 16u * lane;
    cplx fa[FROM_LDS ? MAXG : 1];
    if constexpr (FROM_LDS) {  // the operator elements leave LDS while the block is on its way
#pragma unroll
        for (int j = 0; j < MAXG; ++j) fa[j] = afrag[(size_t)j * 64];
    }
    kh_u64 gq[MAXG][2];
#ifdef KH_TIMING
    const long long tq0 = clock64();
#endif
    for (int d = 0; d < c.first_poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
#ifdef KH_C4_X_QUARTER  // (timing experiment: wrong results) every wave fetches a quarter of the block only
    const int wave_x = threadIdx.x >> 6;
#define KH_C4_J(j) (((j) & 3) == wave_x ? (j) : 1000)
#else
#define KH_C4_J(j) (j)
#endif
    if (c.local) {
#pragma unroll
        for (int j = 0; j < MAXG; ++j)
            kh_coop_load2<KH_CPOL_SC1>(rsrc, KH_C4_J(j) < jmax ? ring_off + 1024u * j : 0xfffffff0u, gq[j][0], gq[j][1]);
    } else {
#pragma unroll
        for (int j = 0; j < MAXG; ++j)
            kh_coop_load2<KH_CPOL_SC0>(rsrc, KH_C4_J(j) < jmax ? ring_off + 1024u * j : 0xfffffff0u, gq[j][0], gq[j][1]);
    }
    // (bitwise, not &&: a short-circuit chain over 26 groups compiles to 26 nested branches with exec juggling)
    unsigned int bad = 0u;
#pragma unroll
    for (int j = 0; j < MAXG; ++j)
        bad |= (((unsigned int)(gq[j][0] >> 32) ^ epoch) | ((unsigned int)(gq[j][1] >> 32) ^ epoch)) & (KH_C4_J(j) < jmax ? ~0u : 0u);
#ifdef KH_TIMING
    if (threadIdx.x == 0 && blockIdx.x == 0 && !__all(bad == 0u)) s.tim[4] += 1.0;
#endif
    if (!__all(bad == 0u)) {  // only what was stale, bypassing the caches, until everything carries the round's tag
        long long t0 = 0;
        unsigned int spins = 0;
        for (;;) {
            kh_compiler_fence();
#pragma unroll
            for (int j = 0; j < MAXG; ++j) {
                const unsigned int stale = (((unsigned int)(gq[j][0] >> 32) ^ epoch) | ((unsigned int)(gq[j][1] >> 32) ^ epoch)) &
                                           (KH_C4_J(j) < jmax ? ~0u : 0u);
                if (__any(stale != 0u))  // (wave-uniform: the group is fetched again by the whole wave)
                    kh_coop_load2<KH_CPOL_SC1>(rsrc, KH_C4_J(j) < jmax ? ring_off + 1024u * j : 0xfffffff0u, gq[j][0], gq[j][1]);
            }
            bad = 0u;
#pragma unroll
            for (int j = 0; j < MAXG; ++j)
                bad |= (((unsigned int)(gq[j][0] >> 32) ^ epoch) | ((unsigned int)(gq[j][1] >> 32) ^ epoch)) & (KH_C4_J(j) < jmax ? ~0u : 0u);
            if (__all(bad == 0u)) break;
            __builtin_amdgcn_s_sleep(2);
            if (spins == 0) t0 = wall_clock64();
            if ((++spins & 63u) == 0) {
                const bool gave_up = (wall_clock64() - t0 > ex.timeout_ticks) ||
                                     (__hip_atomic_load(ex.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u);
                if (__any(gave_up)) {
                    if (lane == 0) {
                        __hip_atomic_store(ex.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s.abort = 1;
                    }
                    return false;
                }
            }
        }
    }
#ifdef KH_TIMING
    const long long tq1 = clock64();
#endif
    // four accumulators (two per part): a dependent v_mfma_f64_4x4x4 issues every ~47 cycles, an independent one
    // every ~19
    double ar0 = 0.0, ai0 = 0.0, ar1 = 0.0, ai1 = 0.0;
#pragma unroll
    for (int j = 0; j < MAXG; ++j) {
        const double v = __hiloint2double((int)(unsigned int)(gq[j][0] & 0xffffffffull), (int)(unsigned int)(gq[j][1] & 0xffffffffull));
        const cplx f = FROM_LDS ? fa[FROM_LDS ? j : 0] : breg[j];
#ifdef KH_C4_X_NOMFMA  // (timing experiment: wrong results)
        ar0 = fma(f.x, v, ar0);
        ai0 = fma(f.y, v, ai0);
        continue;
#endif
        if (j & 1) {
            ar1 = __builtin_amdgcn_mfma_f64_4x4x4f64(f.x, v, ar1, 0, 0, 0);
            ai1 = __builtin_amdgcn_mfma_f64_4x4x4f64(f.y, v, ai1, 0, 0, 0);
        } else {
            ar0 = __builtin_amdgcn_mfma_f64_4x4x4f64(f.x, v, ar0, 0, 0, 0);
            ai0 = __builtin_amdgcn_mfma_f64_4x4x4f64(f.y, v, ai0, 0, 0, 0);
        }
    }
    // Re(F) X = [Fr Xr | Fr Xi], Im(F) X = [Fi Xr | Fi Xi]: F X = first + (-1, +1) * (second with its column pairs swapped);
    // then the four blocks (k-steps) of the instruction: D lane = 16 row + 4 block + column
    const double sgn = (lane & 2) ? 1.0 : -1.0;
    double r = fma(sgn, dpp_move<KH_DPP_XOR2>(ai0 + ai1), ar0 + ar1);
    r += dpp_move<KH_DPP_ROR8>(r);
    r += dpp_move<KH_DPP_ROR4>(r);
    e = r;
#ifdef KH_TIMING
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const long long tq2 = clock64();
        s.tim[0] += (double)(tq1 - tq0);
        s.tim[1] += (double)(tq2 - tq1);
        if (s.tim[3] != 0.0) s.tim[2] += (double)tq0 - s.tim[3];
        s.tim[3] = (double)tq2;
    }
#endif
    return true;
}

#define KH_C4_REFRESH 64
// B = P0, A = H0 (eps' = 0)
template <int MAXG>
__device__ __forceinline__ void kh_c4_restart(const KhCoopArgs &c, const KhC4Masks &mk, int g, int wave, int lane, int NG,
                                              cplx *afrag, cplx (&breg)[MAXG], double &eps_prev) {
    kh_c4_reg_axpy<MAXG, true>(c.sq[0], 1.0, g, wave, lane, NG, breg, mk.p0);
    kh_c4_lds_axpy<MAXG, true>(c.fops[0], 1.0, g, wave, lane, NG, afrag, mk.h0);
    eps_prev = 0.0;
}

// state <- exp(f A dt) state on the A^2 chain (kh_coop_expm_action_sq); all quantities are the lane's real component.
// p1pre: this lane's groups of the P1 table, fetched by the caller while it waited for eps (or NULL)
template <int MAXG>
__device__ __forceinline__ bool kh_c4_expm_action_sq(const KhCoopArgs &c, const KhExchange &ex, __amdgpu_buffer_rsrc_t rsrc,
                                                     const KhC4Masks &mk, double eps, double &eps_prev, cplx *afrag,
                                                     cplx (&breg)[MAXG], double &state, unsigned int &rid, KhC4Lds &s,
                                                     int NG, int jmax, int y, int g, bool publishes,
                                                     double fre, double fim, double dt, int nsub, int m, int tid,
                                                     int wave, int lane, int par, const cplx (*p1pre)[MAXG] = nullptr) {
    const double h = kh_uniform(nsub == 1 ? dt : dt / nsub);
    const double f2h2 = kh_uniform((fre * fre - fim * fim) * h * h);
    const int phases = (m + 1) >> 1;
    const bool series = c.ser_rows != nullptr;
    if (series) {
        const double *grow = c.ser_rows + (size_t)m * KH_Q2_ROWS * 2;
        if (tid < 2 * phases) s.coef[par][tid] = grow[tid];
        if (tid == 2 * phases) s.coef[par][tid] = c.ser_c0[m];
    }
    {   // fragments: B += (eps - eps') P1 + (eps^2 - eps'^2) P2 (registers), A += (eps - eps') H1 (LDS)
        const double e1 = eps - eps_prev, e2 = e1 * (eps + eps_prev);
        if (p1pre != nullptr) {
#pragma unroll
            for (int j = 0; j < MAXG; ++j) {
                breg[j].x = fma(e1, (*p1pre)[j].x, breg[j].x);
                breg[j].y = fma(e1, (*p1pre)[j].y, breg[j].y);
            }
        } else {
            kh_c4_reg_axpy<MAXG, false>(c.sq[1], e1, g, wave, lane, NG, breg, mk.p1);
        }
        kh_c4_reg_axpy<MAXG, false>(c.sq[2], e2, g, wave, lane, NG, breg, mk.p2);
        kh_c4_lds_axpy<MAXG, false>(c.fops[1], e1, g, wave, lane, NG, afrag, mk.h1);
        eps_prev = kh_uniform(eps);
    }
    __syncthreads();  // the series rows are in LDS (and every wave has left the previous interval's rounds)
    const double *rows = series ? s.coef[par] : nullptr;
    const double c_0 = kh_uniform(series ? s.coef[par][2 * phases] : 1.0), c_1 = kh_uniform(series ? s.coef[par][0] : 1.0);
    const bool is_im = (lane & 2) != 0;
    for (int sub = 0; sub < nsub; ++sub) {
        double sacc = h * c_1 * state;
        state = c_0 * state;
        for (int ph = 0; ph < phases; ++ph) {
            const double r2 = kh_uniform(rows != nullptr ? rows[2 * ph + 1] : kh_inv_table[2 * ph + 1] * kh_inv_table[2 * ph + 2]);
            const double r1n = kh_uniform(ph + 1 < phases ? (rows != nullptr ? rows[2 * ph + 2] : kh_inv_table[2 * ph + 3]) : 0.0);
            double w;
            if (!kh_c4_round<MAXG, false>(c, ex, rsrc, rid, y, jmax, breg, afrag, s, lane, w)) return false;
            const double t2 = f2h2 * r2 * w;
            state += t2;
            const bool last = ph + 1 == phases;
            if (!last) sacc = fma(h * r1n, t2, sacc);
            if (publishes) kh_c4_publish(c, rsrc, rid + 1, y, g, wave, lane, last ? sacc : t2);
            ++rid;
        }
        double w;
        if (!kh_c4_round<MAXG, true>(c, ex, rsrc, rid, y, jmax, breg, afrag, s, lane, w)) return false;
        const double partner = dpp_move<KH_DPP_XOR2>(w);  // the other part (re <-> im) of the same element
        state += is_im ? fma(fim, partner, fre * w) : fma(-fim, partner, fre * w);
        if (publishes) kh_c4_publish(c, rsrc, rid + 1, y, g, wave, lane, state);
        ++rid;
    }
    return true;
}

// (all threads; contains barriers) s.local <- 1 iff the G workgroups of column group y report one XCC id
__device__ __forceinline__ void kh_c4_check_placement(const KhCoopArgs &c, const KhExchange &ex, KhC4Lds &s, int g, int y,
                                                      int tid) {
    if (tid == 0) s.local = 0;
    if (c.xcd_rows <= 0 || c.xcc == nullptr) {
        __syncthreads();
        return;
    }
    if (tid == 0) {
        const unsigned int mine = kh_xcc_id() + 1u;
        __hip_atomic_store(c.xcc + (size_t)y * c.G + g, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool same = true;
        const long long t0 = wall_clock64();
        for (int i = 0; i < c.G && same; ++i) {
            unsigned int v;
            while ((v = __hip_atomic_load(c.xcc + (size_t)y * c.G + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
                if (wall_clock64() - t0 > ex.timeout_ticks) {
                    v = 0xffffffffu;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            same = v == mine;
        }
        s.local = same ? 1 : 0;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// plain propagation with storage (backward sweep / iteration-0 forward sweep)
// ---------------------------------------------------------------------------
template <int MAXG>
__global__ void __launch_bounds__(KH_C4_THREADS)
kh_c4_sweep_store(KhSweepArgs p, KhCoopArgs c_in, KhExchange ex, const double *__restrict__ pulses,
                  const cplx *__restrict__ state_in, cplx *__restrict__ store, cplx *__restrict__ state_out, int direction) {
    extern __shared__ __attribute__((aligned(16))) char kh_c4_smem[];
    KhC4Lds &s = *(KhC4Lds *)kh_c4_smem;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int g, y;
    if (!kh_coop_place(c_in, g, y)) return;
    const int N = p.N, nt = p.nt, NG = c_in.ks;
    if (tid <= KH_MAX_DEGREE) s.deg[tid] = c_in.ser_theta != nullptr ? c_in.ser_theta[tid] : p.deg_theta[tid];
    if (tid == 0) s.abort = 0;
#ifdef KH_TIMING
    if (tid < 8) s.tim[tid] = 0.0;
#endif
    kh_c4_check_placement(c_in, ex, s, g, y, tid);
    KhCoopArgs c = c_in;
    c.local = __builtin_amdgcn_readfirstlane(s.local);
    // this lane's component: row, objective, part
    const int row = 16 * g + 4 * wave + (lane >> 4), k = 2 * y + (lane & 1);
    const bool is_im = (lane & 2) != 0, first = ((lane >> 2) & 3) == 0;
    const bool publishes = first && row < N, has_state = publishes && k < p.K;
    // groups whose row at this lane's position (4 block + k-step) exists: 16 j + 4 ((lane >> 2) & 3) + (lane >> 4) < N
    const int jmax = (N - (4 * ((lane >> 2) & 3) + (lane >> 4)) + 15) / 16;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)c.vbuf, 0, (int)((size_t)KH_COOP_RING * c.Y * c.G * 1024u), 0x00020000);
    const size_t comp = is_im ? 1 : 0;
    double state = (row < N && k < p.K) ? ((const double *)state_in)[((size_t)k * N + row) * 2 + comp] : 0.0;
    unsigned int rid = 1;
    if (publishes) kh_c4_publish(c, rsrc, rid, y, g, wave, lane, state);
    if (has_state && store != nullptr)
        ((double *)store)[(((size_t)k * nt + (direction > 0 ? 0 : nt - 1)) * N + row) * 2 + comp] = state;
    cplx *afrag = (cplx *)s.frag + (size_t)wave * MAXG * 64 + lane;
#pragma unroll
    for (int j = 0; j < MAXG; ++j) afrag[(size_t)j * 64] = c_make(0.0, 0.0);  // (groups beyond NG stay zero)
    cplx breg[MAXG];
#pragma unroll
    for (int j = 0; j < MAXG; ++j) breg[j] = c_make(0.0, 0.0);
    double eps_prev = 0.0;
    const KhC4Masks mk = kh_c4_masks(c, g, wave, NG);
    int rounds = 0, m_hint = 12;
    __syncthreads();
    for (int step = 0; step < nt - 1; ++step) {
        const int n = direction > 0 ? step : nt - 2 - step;
        if (step % KH_C4_REFRESH == 0) kh_c4_restart<MAXG>(c, mk, g, wave, lane, NG, afrag, breg, eps_prev);
        const double eps = kh_uniform(pulses[n]);
        const double dt = kh_uniform(p.dt[n]);
        const double theta = kh_uniform(p.op_norms[0] + fabs(eps) * p.op_norms[1]);
        int nsub, m;
        kh_degree_lookup(theta * dt, s.deg, p.theta_max, p.inv_theta_max, m_hint, &nsub, &m);
        m_hint = m;
        if (!kh_c4_expm_action_sq<MAXG>(c, ex, rsrc, mk, eps, eps_prev, afrag, breg, state, rid, s, NG, jmax, y, g,
                                        publishes, p.fre, p.fim, dt, nsub, m, tid, wave, lane, step & 1))
            return;
        rounds += nsub * (((m + 1) >> 1) + 1);
        if (has_state && store != nullptr)
            ((double *)store)[(((size_t)k * nt + (direction > 0 ? n + 1 : n)) * N + row) * 2 + comp] = state;
    }
    if (has_state && state_out != nullptr) ((double *)state_out)[((size_t)k * N + row) * 2 + comp] = state;
    if (g == 0 && tid == 0 && p.stats != nullptr) {
        const int cols = min(2, p.K - 2 * y);
        atomicAdd(p.stats, (double)rounds * cols);
#ifdef KH_TIMING
        if (y == 0) {  // cycles per round: fetch | matrix cores + sums | between rounds (incl. the fragment updates); slow-path share
            p.stats[1] = s.tim[0] / rounds;
            p.stats[2] = s.tim[1] / rounds;
            p.stats[3] = s.tim[2] / rounds + 1e6 * (double)(long long)(100.0 * s.tim[4] / rounds);
        }
#endif
    }
}
