// Cooperative sweep kernels for objectives that SHARE one operator list and whose state dimension
// is too large for a register tile (64 < N <= 512): BASELINE config 4, 16 density matrices
// under one 400 x 400 Liouvillian.
//
// With shared operators one Taylor term for all objectives is a genuine dense product
//     W (N x K)  =  A (N x N)  .  T (N x K),        A = op_0 + sum_l eps_l op_l,
// so it goes to the fp64 matrix cores: v_mfma_f64_16x16x4_f64 (16 objectives per workgroup) or v_mfma_f64_4x4x4_4b
// (4 or 2), complex arithmetic as four (two) real products.
//
// Decomposition: workgroup (g, y) owns rows [16 g, 16 g + 16) of A and `cols` objectives (16, 4 or 2: as few as keeps
// the grid co-resident, and -- with 2 or 4 -- every column group on an XCD of its own, kh_coop_place).  Its 8 waves
// split the k range; the partial blocks are summed through LDS.  Every term needs the whole previous term, i.e. the
// blocks of all row workgroups of the column group: they are exchanged through a ring in global memory made of
// epoch-tagged 8-byte granules {epoch:32 | half of a double:32} -- the data is its own "ready" flag, so a round costs
// ONE memory round trip instead of data + flag.  A workgroup can be at most one round ahead of the slowest reader (it
// needs everybody's block of round r before it can publish r + 1), so any ring of >= 2 blocks is race free; the ring is
// KH_COOP_RING deep so that a block's lines have left the (per-XCD, not cross-XCD coherent) L2 before they are reused.
// Layouts and scopes: kh_coop_slot (16 objectives), kh_coop_slot4 / kh_coop_publish (4 and 2: the consumer's lane
// order, 16-byte loads).
//
// One control: the series runs on B = A^2 (kh_coop_expm_action_sq): B lives in registers, A in LDS, both advanced
// from interval to interval instead of rebuilt; operator tables are kept in fragment order with a zero-slot mask
// (kh_coop_permute_kernel, kh_coop_mask_kernel).  Several controls: term by term on the LDS fragment of A.
//
// All workgroups must be co-resident; spins are bounded by the exchange timeout and raise the engine's abort flag.
// History, measurements and what bounds a round: DESIGN.md section 3.5.
#pragma once

#include "kh_common.h"
#include "kh_generic.h"

#define KH_COOP_THREADS 512
#define KH_COOP_WAVES 8
#define KH_COOP_COLS 16     // MFMA N: a workgroup handles c.cols <= 16 objectives (the rest of the tile is zero)
// owner threads: tid < 16 COLS owns element (row tid / COLS, column tid % COLS) of the block (one wave for COLS = 4)
#define KH_COOP_RING 32      // blocks in the exchange ring (as allocated; in use across XCDs)
// ... of which a column group that sits on ONE XCD uses the first few: its L2 is coherent for its own workgroups, so
// the race-free minimum (2) would do, and a short ring keeps the L2 for the operator tables -- with 32 blocks in turn a
// group's ring took 0.8 MB of the 4 MB next to 3.3 MB of P1 slices, and every interval's table read missed somewhere
// (config 4: backward 18.4 -> 17.7 ms, update 20.7 -> 19.7 ms)
#ifndef KH_COOP_RING_LOCAL
#define KH_COOP_RING_LOCAL 4
#endif
#ifndef KH_COOP_CHUNK
#define KH_COOP_CHUNK 4     // slots whose loads are in flight together in a fragment update (MAXKS is a multiple)
#endif
#define KH_COOP_MAX_L 2     // controls (the update-sum exchange keeps 16 registers per control in flight)

typedef double kh_d4 __attribute__((ext_vector_type(4)));

// Owner threads.  COLS = 16 / 4: thread tid < 16 COLS owns element (row tid / COLS, column tid % COLS) of the block.
// COLS = 2: the owners are spread over all waves -- wave w, lane 16 j owns row 2 w + j / 2, column j % 2 -- so that the
// cross-wave sum of a round is ONE LDS read per lane and a sum over 8 neighbouring lanes (kh_coop_round4) instead of
// sixteen dependent reads in one wave.
template <int COLS>
__device__ __forceinline__ bool kh_coop_is_owner(int tid) {
    if constexpr (COLS == 2) return (tid & 15) == 0;
    return tid < 16 * COLS;
}
template <int COLS>
__device__ __forceinline__ void kh_coop_owner_element(int tid, int &r, int &col) {
    if constexpr (COLS == 2) {
        r = 2 * (tid >> 6) + ((tid >> 5) & 1);
        col = (tid >> 4) & 1;
    } else {
        r = tid / COLS;
        col = tid % COLS;
    }
}

typedef unsigned int kh_u32x4 __attribute__((ext_vector_type(4)));
template <class T>
__device__ __forceinline__ T *kh_uniform_ptr(T *p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v);
    const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}
struct KhCoopArgs {
    kh_u64 *vbuf;             // cols = 16: [KH_COOP_RING][Y][G*16][4][16] granules; cols = 4: [KH_COOP_RING][Y][G]
                              // [2][64][2], a 16-row group in the CONSUMER's lane order (kh_coop_slot4)
    unsigned int epoch_base;  // rounds of earlier launches (tags are monotonic: the buffer is never cleared)
    int G, Y;                 // row blocks, column groups
    int cols;                 // objectives per column group (the kernels' COLS): 16 or 4.  Fewer columns = more
                              // workgroups, each fetching a narrower block per term with fewer loads
    int ks;                   // k-steps (of 4 columns) per wave: 32 * ks >= N
    int first_poll_delay;     // s_sleep units (64 cycles) before a round's first fetch
    int xcd_rows;             // > 0: one-dimensional grid of 8 * xcd_rows blocks, block b = 8 g + y (see kh_coop_place)
    int local;                // (set in the kernel after kh_coop_check_placement: a wave-uniform copy of KhCoopLds::local)
    unsigned int ring_mask;   // blocks of the ring in use - 1 (set with `local`: KH_COOP_RING_LOCAL or KH_COOP_RING)
    unsigned int *xcc;        // [Y * G] XCC id + 1 of every workgroup (placement check, zeroed with vbuf)
    const cplx *const *fops;  // [1 + L] this direction's (shared) operators in fragment order
    const double *ser_theta;  // A^2 chain only: degree thresholds, c_0 and rows {c_{2p+1}/c_{2p}, c_{2p+2}/c_{2p}} of the series
    const double *ser_c0;     // in use (kh_common.h: Chebyshev form for generators with an (almost) imaginary spectrum),
    const double *ser_rows;   // or NULL: Taylor
    const cplx *const *sq;    // one control only: P0 = H0 H0, P1 = H0 H1 + H1 H0, P2 = H1 H1 of this direction's
                              // operators (A^2 = P0 + eps P1 + eps^2 P2) in fragment order, or NULL: term-by-term series
    const cplx *tab[5];       // (set in the kernel, kh_coop_resolve_tables: fops[0], fops[1], sq[0], sq[1], sq[2] read ONCE,
                              // as wave-uniform values -- read where they are used, each is a dependent vector load in
                              // front of the table read it addresses, once per interval and table)
};

struct KhCoopLds {
    double red[2][KH_COOP_WAVES][KH_COOP_MAX_L];  // owner waves' pieces of the update sums, by interval parity
    double D[2][KH_COOP_MAX_L + 1];        // reduced sums + ok flag, by interval parity
    double deg[KH_MAX_DEGREE + 2];
    double coef[2 * KH_Q2_ROWS + 2];  // the current interval's series rows (kh_coop_expm_action_sq)
    int abort;
    int local;  // this column group's workgroups all sit on ONE XCD: term blocks are exchanged through its L2
#ifdef KH_TIMING
    double tim[12];  // [7] owners' sums, [8] second barrier, [9] between rounds (caller), [10] fragment updates, [11] stamp
#endif
    __attribute__((aligned(16))) double frag[1];  // [ks][KH_COOP_THREADS] complex operator fragment of the current interval (dynamic size), then
                     // the per-wave partial blocks: [WAVES][8][64] (16 objectives per workgroup: re regs 0-3, im 4-7)
                     // or [WAVES][2][64] (4 objectives: element (row r, column c) of the block at [4 r + c])
};

#define KH_COOP_PART2 66  // cols = 2: a wave's 64 partial values, padded (the cross-wave read strides over waves)
__host__ __device__ inline size_t kh_coop_lds_bytes(int ks, int cols = KH_COOP_COLS) {
    return sizeof(KhCoopLds) + sizeof(double) * 2 * (size_t)ks * KH_COOP_THREADS +
           sizeof(double) * KH_COOP_WAVES * (cols == 2 ? 2 * KH_COOP_PART2 : cols == 4 ? 2 * 64 : 8 * 64);
}
__device__ __forceinline__ double *kh_coop_part(KhCoopLds &s, int ks) { return s.frag + 2 * (size_t)ks * KH_COOP_THREADS; }

// 4 objectives per workgroup: the wave's share of the k range, in groups of 16 columns of the operator (balanced:
// the first `groups % 8` waves take one group more)
__host__ __device__ inline int kh_coop4_groups(int N) { return (N + 15) / 16; }
__host__ __device__ inline int kh_coop4_slots(int N) { return 4 * ((kh_coop4_groups(N) + KH_COOP_WAVES - 1) / KH_COOP_WAVES); }
__host__ __device__ inline void kh_coop4_share(int N, int wave, int *start, int *count) {
    const int groups = kh_coop4_groups(N), base = groups / KH_COOP_WAVES, extra = groups % KH_COOP_WAVES;
    *count = base + (wave < extra ? 1 : 0);
    *start = wave * base + (wave < extra ? wave : extra);
}

// A thread's view of the fragment: element q is the complex number at f[q T] (one 16-byte LDS access)
struct KhCoopFrag {
    cplx *f;  // (cplx *)s.frag + tid
    __device__ __forceinline__ double &re(int q) const { return f[(size_t)q * KH_COOP_THREADS].x; }
    __device__ __forceinline__ double &im(int q) const { return f[(size_t)q * KH_COOP_THREADS].y; }
    __device__ __forceinline__ cplx get(int q) const { return f[(size_t)q * KH_COOP_THREADS]; }
    __device__ __forceinline__ bool nz(int) const { return true; }
};

// The same view of a fragment held in registers (slot index known at compile time after unrolling)
template <int MAXKS, bool MASKED = false>
struct KhCoopRegFrag {
    const cplx (&v)[MAXKS];
    unsigned int mask;  // MASKED: bit q clear = slot q is zero in every lane of the wave (its products are skipped)
    __device__ __forceinline__ double re(int q) const { return v[q].x; }
    __device__ __forceinline__ double im(int q) const { return v[q].y; }
    __device__ __forceinline__ cplx get(int q) const { return v[q]; }
    __device__ __forceinline__ bool nz(int q) const { return !MASKED || ((mask >> q) & 1u) != 0u; }
};

// Operators are re-laid out once, at engine creation, in "fragment order": the element that lane `lane` of
// wave `wave` of row block g holds for k-step q -- row 16 g + (lane & 15), column (wave ks + q) 4 + (lane >> 4),
// the MFMA A-operand layout -- sits at [((g WAVES + wave) ks + q) 64 + lane], zero beyond N.  A fragment
// (re)build then reads 1 KiB per wave-level load, fully coalesced, without bounds checks; from the row-major
// matrix the same load touched 16 half-used lines (measured: 4.2 -> see DESIGN.md us per table per interval).
// With 4 objectives per workgroup (cols == 4) the four 4x4x4 blocks of v_mfma_f64_4x4x4_4b take four consecutive
// k-steps instead of four row groups: slot q = 4 gi + rb of a wave is the A operand for its gi-th group of 16
// columns and the row group rb -- row 16 g + 4 rb + (lane & 3), column 16 (start + gi) + 4 ((lane >> 2) & 3) +
// (lane >> 4).  The vector block of a group then IS the B operand as loaded (one element per lane, no replication
// across blocks, no ds_bpermute): see kh_coop_round.
// Row block g starts at g * kh_coop_table_stride(ks): the fragment size plus KH_COOP_TABLE_PAD elements, so that the
// row blocks -- read by the workgroups of an XCD at the same moment, same (wave, slot) offset -- do not start a
// power of two apart and fall onto one L2 channel each.
#ifndef KH_COOP_TABLE_PAD
#define KH_COOP_TABLE_PAD 272  // 4096 + 256 bytes
#endif
__host__ __device__ inline size_t kh_coop_table_stride(int ks) { return (size_t)KH_COOP_WAVES * ks * 64 + KH_COOP_TABLE_PAD; }
__host__ __device__ inline size_t kh_coop_table_elems(int G, int ks) { return (size_t)G * kh_coop_table_stride(ks); }
__global__ void kh_coop_permute_kernel(const cplx *__restrict__ in, cplx *__restrict__ out, int N, int G, int ks,
                                       int cols)
#if KH_DEFINES(KH_TU_MAIN)
{
    const size_t total = (size_t)G * KH_COOP_WAVES * ks * 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const size_t t = idx >> 6;
        const int q = (int)(t % ks), wave = (int)((t / ks) % KH_COOP_WAVES), g = (int)(t / ks / KH_COOP_WAVES);
        int row, col;
        if (cols <= 4) {
            int start, count;
            kh_coop4_share(N, wave, &start, &count);
            const int gi = q >> 2, rb = q & 3;
            row = g * 16 + 4 * rb + (lane & 3);
            col = gi < count ? 16 * (start + gi) + 4 * ((lane >> 2) & 3) + (lane >> 4) : N;
        } else {
            row = g * 16 + (lane & 15);
            col = (wave * ks + q) * 4 + (lane >> 4);
        }
        out[(size_t)g * kh_coop_table_stride(ks) + (size_t)(wave * ks + q) * 64 + lane] =
            (row < N && col < N) ? in[(size_t)row * N + col] : c_make(0.0, 0.0);
    }
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif

// Zero slots.  Operators of physical models are often sparse in places (a control Hamiltonian's commutator
// superoperator has a few entries per row; so has its square), and every table is zero-padded from N to the waves'
// share of 16-column groups.  A slot -- the 64 elements one wave loads at once -- that is zero in all lanes is
// marked once, at engine creation, in a 32-bit word per (row block, wave) stored BEHIND the table; fragment loads,
// updates and (for the control operators' products) matrix-core instructions skip such slots.  Exact: skipped
// terms are exact zeros.
__global__ void kh_coop_mask_kernel(const cplx *__restrict__ tab, unsigned int *__restrict__ mask, int ks)
#if KH_DEFINES(KH_TU_MAIN)
{
    const int lane = threadIdx.x;  // one wave per (row block, wave)
    const cplx *src = tab + (size_t)(blockIdx.x / KH_COOP_WAVES) * kh_coop_table_stride(ks) +
                      (size_t)(blockIdx.x % KH_COOP_WAVES) * ks * 64 + lane;
    unsigned int m = 0;
    for (int q = 0; q < ks; ++q) {
        const cplx v = src[(size_t)q * 64];
        if (__ballot(v.x != 0.0 || v.y != 0.0) != 0ull) m |= 1u << q;
    }
    if (lane == 0) mask[blockIdx.x] = m;
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif
__device__ __forceinline__ unsigned int kh_coop_frag_mask(const cplx *op, int G, int g, int wave, int ks) {
    if (op == nullptr) return 0u;
    const unsigned int *m = (const unsigned int *)(op + kh_coop_table_elems(G, ks));
    return __builtin_amdgcn_readfirstlane(m[g * KH_COOP_WAVES + wave]);
}

__device__ __forceinline__ void kh_coop_resolve_tables(KhCoopArgs &c) {
#pragma unroll
    for (int i = 0; i < 5; ++i) c.tab[i] = nullptr;
    if (c.sq != nullptr) {
        c.tab[0] = kh_uniform_ptr(c.fops[0]);
        c.tab[1] = kh_uniform_ptr(c.fops[1]);
#pragma unroll
        for (int i = 0; i < 3; ++i) c.tab[2 + i] = kh_uniform_ptr(c.sq[i]);
    }
}

// Workgroup placement.  The term block of a column group is written by its G workgroups and read by the same G
// workgroups, every round.  If they all sit on ONE XCD, the block can stay in that XCD's L2: producers store with
// workgroup scope (the line is kept in L2), consumers load with agent scope (L1 bypassed, L2-served) -- no trip
// through the memory side.  Blocks are dispatched round-robin over the 8 XCDs (block b on XCD b % 8: observed, not
// promised), so the launch uses a one-dimensional grid with b = 8 g + y: column group y lands on XCD y.  Nothing
// relies on that: every workgroup publishes the XCC id it really runs on (safe protocol), and a column group uses
// the L2 form only if all its members report the same one -- otherwise the memory-side form below.
#define KH_HW_REG_XCC_ID 20
__device__ __forceinline__ unsigned int kh_xcc_id() {
    return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | KH_HW_REG_XCC_ID) & 0xf;
}
__device__ __forceinline__ bool kh_coop_place(const KhCoopArgs &c, int &g, int &y) {
    if (c.xcd_rows > 0) {
        g = blockIdx.x >> 3;
        y = blockIdx.x & 7;
        return g < c.G && y < c.Y;
    }
    g = blockIdx.x;
    y = blockIdx.y;
    return true;
}
// (all threads; contains barriers) s.local <- 1 iff the G workgroups of column group y report one XCC id
__device__ __forceinline__ void kh_coop_check_placement(const KhCoopArgs &c, const KhExchange &ex, KhCoopLds &s, int g,
                                                        int y, int tid) {
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
                if (wall_clock64() - t0 > ex.timeout_ticks) {  // (somebody is not there: the rounds will notice too)
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

// this lane's elements of row block g of a fragment-ordered operator (NULL: zero operator)
// (The table pointers come out of a pointer table in memory, so the compiler knows no address space for them and
// would read the tables with FLAT loads -- which also count against the LDS counter and so serialise with the LDS
// traffic of a fragment update.  They are global memory: say so.)
typedef double kh_d2 __attribute__((ext_vector_type(2)));
// A lane's view of a fragment-ordered table: ONE buffer resource per (table, row block) in SGPRs and one 32-bit lane
// offset shared by all tables; slot q is the scalar offset 1024 q.  (64-bit lane addresses per table and slot, once
// the table pointers are wave-uniform values, were hoisted out of the interval loop by the compiler: 140 spilled
// registers.)  A NULL table reads as zeros (no records).
struct KhCoopSrc {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned int voff;
    bool is_null;
    __device__ __forceinline__ bool null() const { return is_null; }
    __device__ __forceinline__ cplx operator[](size_t i) const {
        const kh_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)(unsigned int)(i * sizeof(cplx)), 0);
        return c_make(__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z));
    }
};
__device__ __forceinline__ KhCoopSrc kh_coop_frag_src(const cplx *op_in, int g, int wave, int lane, int ks) {
    const cplx *op = kh_uniform_ptr(op_in);
    KhCoopSrc r;
    r.is_null = op == nullptr;
    const cplx *base = op == nullptr ? nullptr : op + (size_t)__builtin_amdgcn_readfirstlane(g) * kh_coop_table_stride(ks);
    r.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0,
                                               op == nullptr ? 0 : (int)(sizeof(cplx) * KH_COOP_WAVES * ks * 64), 0x00020000);
    r.voff = (unsigned int)(sizeof(cplx) * ((size_t)wave * ks * 64 + lane));
    return r;
}

template <int MAXKS>
__device__ __forceinline__ void kh_coop_load_frag(const cplx *op, int g, int wave, int lane, int ks,
                                                  const KhCoopFrag &f, unsigned int mask = ~0u) {
    const KhCoopSrc src = kh_coop_frag_src(op, g, wave, lane, ks);
#pragma unroll
    for (int q = 0; q < MAXKS; ++q) {
        if (q < ks) {
            f.f[(size_t)q * KH_COOP_THREADS] =
                (!src.null() && ((mask >> q) & 1u)) ? src[(size_t)q * 64] : c_make(0.0, 0.0);
        }
    }
}

// a += eps * op  (same fragment layout)
template <int MAXKS>
__device__ __forceinline__ void kh_coop_axpy_frag(const cplx *op, double eps, int g, int wave, int lane, int ks,
                                                  const KhCoopFrag &a, unsigned int mask = ~0u) {
    const KhCoopSrc src = kh_coop_frag_src(op, g, wave, lane, ks);
    if (src.null()) return;
    // in chunks of four slots, loads (table and LDS) before uses -- see kh_coop_reg_axpy
#pragma unroll
    for (int c0 = 0; c0 < MAXKS; c0 += 4) {
        if (c0 < ks && ((mask >> c0) & 0xfu) != 0u) {
            cplx v[4], t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = src[(size_t)(c0 + j < ks ? c0 + j : ks - 1) * 64];
                t[j] = a.get(c0 + j < MAXKS ? c0 + j : MAXKS - 1);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (c0 + j < ks) {
                    t[j].x = fma(eps, v[j].x, t[j].x);
                    t[j].y = fma(eps, v[j].y, t[j].y);
                    a.f[(size_t)(c0 + j) * KH_COOP_THREADS] = t[j];
                }
            }
        }
    }
}

// granule i (re hi, re lo, im hi, im lo) of element (row, col) is slot(...)[i * 16]: the 16 columns
// of one granule index are contiguous, so a wave's 8-byte accesses cover whole 128-byte lines
__device__ __forceinline__ kh_u64 *kh_coop_slot(const KhCoopArgs &c, unsigned int rid, int y, int row, int col) {
    return c.vbuf +
           (((size_t)(rid & c.ring_mask) * c.Y + y) * ((size_t)c.G * 16) + row) * (4 * KH_COOP_COLS) + col;
}

// 4 objectives per workgroup: a 16-row group of the block is 2 KiB laid out as the consumer's wave reads it --
// [half: re, im][lane][2 granules: hi, lo], lane (hi, b, lo) <-> row 4 b + hi, column lo (kh_coop_round4) -- so
// one 16-byte load per lane and half fetches eight whole 128-byte lines (the first layout, 4 columns of the
// 16-column rows above, touched 64 lines per group with 32 useful bytes each: the round's fetch was bound by the
// number of line requests a CU can issue, 2048 per round, not by latency).
__device__ __forceinline__ int kh_coop4_lane(int row_in_group, int col) {
    return 16 * (row_in_group & 3) + 4 * (row_in_group >> 2) + col;
}
__device__ __forceinline__ size_t kh_coop_group4(const KhCoopArgs &c, unsigned int rid, int y, int group) {
    return (((size_t)(rid & c.ring_mask) * c.Y + y) * (size_t)c.G + group) * 256;  // (granules)
}
__device__ __forceinline__ kh_u64 *kh_coop_slot4(const KhCoopArgs &c, unsigned int rid, int y, int group) {
    return c.vbuf + kh_coop_group4(c, rid, y, group);
}
#define KH_CPOL_SC0 1           // workgroup scope (may be served by the L1 / a stale L2 line)
#define KH_CPOL_SC1 16          // agent scope
// two granules (16 bytes) of the lane's element; each granule carries its own tag, so only 8-byte atomicity is used.
// (Not marked volatile: the compiler turns that into system scope, sc0 sc1 -- a slower path.  A polling loop puts a
// compiler barrier, kh_compiler_fence, before every pass instead.)
__device__ __forceinline__ void kh_compiler_fence() { asm volatile("" ::: "memory"); }
template <int CPOL>
__device__ __forceinline__ void kh_coop_load2(__amdgpu_buffer_rsrc_t rsrc, unsigned int byte_off, kh_u64 &a, kh_u64 &b) {
    const kh_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, CPOL);
    a = (kh_u64)v.x | ((kh_u64)v.y << 32);
    b = (kh_u64)v.z | ((kh_u64)v.w << 32);
}

// owner thread: element (row, col) of round `rid`
template <int COLS>
__device__ __forceinline__ void kh_coop_publish(const KhCoopArgs &c, unsigned int rid, int y, int row, int col,
                                                cplx v, bool local = false) {
    const kh_u64 tag = (kh_u64)(c.epoch_base + rid) << 32;
    const kh_u64 re = (kh_u64)__double_as_longlong(v.x), im = (kh_u64)__double_as_longlong(v.y);
#ifdef KH_COOP_X_NOSTORE
    if (rid > 1) return;
#endif
#ifdef KH_COOP_STRESS  // (protocol test build: publications delayed pseudo-randomly per workgroup and round -- results must not change)
    {
        const unsigned int hsh = (blockIdx.x * 2654435761u) ^ (rid * 40503u);
        const int naps = (int)((hsh >> 7) % 23u);
        for (int d = 0; d < naps; ++d) __builtin_amdgcn_s_sleep(16);
    }
#endif
    if constexpr (COLS == 2 || COLS == 4) {
        // The element's two granules per part are adjacent: ONE 16-byte store per part ({hi | tag}, {lo | tag}; each
        // 8-byte granule still carries its own tag) instead of two 8-byte ones.  COLS = 2: a group is 1 KiB,
        // [lane][2 granules], lane column n = re of objective n / im of objective n - 2; COLS = 4: 2 KiB,
        // [half: re, im][lane][2 granules].
        constexpr unsigned int GB = COLS == 4 ? 2048u : 1024u;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)c.vbuf, 0, (int)((size_t)KH_COOP_RING * c.Y * c.G * GB), 0x00020000);
        const unsigned int off = (unsigned int)(kh_coop_group4(c, rid, y, row >> 4) * 8 / (2048u / GB)) +
                                 16u * (unsigned int)kh_coop4_lane(row & 15, col);
        const unsigned int t32 = (unsigned int)(tag >> 32);
        kh_u32x4 vr, vi;
        vr.x = (unsigned int)(re >> 32);
        vr.y = t32;
        vr.z = (unsigned int)(re & 0xffffffffull);
        vr.w = t32;
        vi.x = (unsigned int)(im >> 32);
        vi.y = t32;
        vi.z = (unsigned int)(im & 0xffffffffull);
        vi.w = t32;
        constexpr unsigned int IM = COLS == 4 ? 1024u : 32u;  // bytes from the re granules to the im granules
        if (local) {
            __builtin_amdgcn_raw_buffer_store_b128(vr, rsrc, (int)off, 0, KH_CPOL_SC0);
            __builtin_amdgcn_raw_buffer_store_b128(vi, rsrc, (int)(off + IM), 0, KH_CPOL_SC0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(vr, rsrc, (int)off, 0, KH_CPOL_SC1);
            __builtin_amdgcn_raw_buffer_store_b128(vi, rsrc, (int)(off + IM), 0, KH_CPOL_SC1);
        }
        return;
    }
    kh_u64 *g = kh_coop_slot(c, rid, y, row, col);
    if (local) {  // (kept in the XCD's L2: kh_coop_place)
        __hip_atomic_store(g + 0 * KH_COOP_COLS, tag | (re >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(g + 1 * KH_COOP_COLS, tag | (re & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(g + 2 * KH_COOP_COLS, tag | (im >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(g + 3 * KH_COOP_COLS, tag | (im & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    __hip_atomic_store(g + 0 * KH_COOP_COLS, tag | (re >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(g + 1 * KH_COOP_COLS, tag | (re & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(g + 2 * KH_COOP_COLS, tag | (im >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(g + 3 * KH_COOP_COLS, tag | (im & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One round: W = F . T_rid for this workgroup's 16 rows.  Every wave fetches its k slice of round
// `rid` in the MFMA B-operand layout (element [k = (wave ks + q) 4 + (lane >> 4)][column lane & 15])
// -- all granule loads of a polling pass in flight together, one memory round trip -- polls until
// every granule carries the round's tag, and multiplies the slice on the matrix cores.  The 8
// partial blocks are summed through LDS; owner threads get their element in `w`.  Contains two
// __syncthreads; a timeout raises s.abort before the first.
// COLS (16 or 4) objectives per workgroup.  With 4, a load instruction fetches FOUR k-steps of the narrow
// block (lane -> k offset lane / 4, column lane % 4) instead of one with 48 idle lanes -- the round is
// bound by the number of wave-level loads the CU issues (8 waves x 52 at N = 400), not by their bytes --
// and the elements are moved into the MFMA operand layout with ds_bpermute.
template <int MAXKS, int COLS, class Frag>
__device__ __forceinline__ void kh_coop_round16(const KhCoopArgs &c, const KhExchange &ex, unsigned int rid, int y,
                                              int N, const Frag &f, KhCoopLds &s, int tid, int wave, int lane,
                                              cplx &w) {
    double *part = kh_coop_part(s, c.ks);
    constexpr int KPL = 16 / COLS;               // k-steps per load group
    constexpr int NG = (MAXKS + KPL - 1) / KPL;  // load groups
    const unsigned int epoch = c.epoch_base + rid;
    const int lcol = lane % COLS, lk = lane / COLS;  // this lane's element of a load group: k offset, column
    kh_u64 g[NG][4];
    const long long t0 = wall_clock64();
#ifdef KH_TIMING
    const long long tq0 = clock64();
#endif
    unsigned int spins = 0;
    // A pass issued before the slowest producer's stores have reached the memory side costs a whole
    // extra round trip: give them a head start, and re-fetch only what was stale afterwards.
    for (int d = 0; d < c.first_poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
    // first pass through L2 (fast; may see a stale line); padding: tag ok, value +0.0
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int kstep = j * KPL + (lk >> 2);  // k-step of this lane's element within the wave's slice
        const int row = (wave * c.ks) * 4 + j * 4 * KPL + lk;
#pragma unroll
        for (int i = 0; i < 4; ++i) g[j][i] = (kh_u64)epoch << 32;
        if (kstep < c.ks && row < N) {
            const kh_u64 *sl = kh_coop_slot(c, rid, y, row, lcol);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                g[j][i] = __hip_atomic_load(sl + i * KH_COOP_COLS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    // the usual case -- everything fresh -- must stay cheap: one pass of compares, no reload code
    bool all_fresh = true;
#pragma unroll
    for (int j = 0; j < NG; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) all_fresh = all_fresh && ((unsigned int)(g[j][i] >> 32) == epoch);
#ifdef KH_TIMING
    {
        const int stale_lanes = __popcll(__ballot(!all_fresh));
        if (tid == 0 && blockIdx.x == 0) s.tim[5] += (double)stale_lanes;
    }
    const long long tqf = clock64();
    if (tid == 0 && blockIdx.x == 0) s.tim[4] += (double)(tqf - tq0);
#endif
    // otherwise: only what was stale, bypassing L2, until everything carries the round's tag
    while (!__all(all_fresh)) {
#ifdef KH_TIMING
        if (tid == 0 && blockIdx.x == 0) s.tim[6] += 1.0;
#endif
        bool ok = true;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int kstep = j * KPL + (lk >> 2);
            const int row = (wave * c.ks) * 4 + j * 4 * KPL + lk;
            if (kstep < c.ks && row < N) {
                const kh_u64 *sl = kh_coop_slot(c, rid, y, row, lcol);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if ((unsigned int)(g[j][i] >> 32) != epoch)
                        g[j][i] = __hip_atomic_load(sl + i * KH_COOP_COLS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int j = 0; j < NG; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) ok = ok && ((unsigned int)(g[j][i] >> 32) == epoch);
        all_fresh = ok;
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 63u) == 0) {
            const bool gave_up =
                (wall_clock64() - t0 > ex.timeout_ticks) ||
                (__hip_atomic_load(ex.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u);
            if (__any(gave_up)) {
                if (lane == 0) {
                    __hip_atomic_store(ex.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s.abort = 1;
                }
                break;
            }
        }
    }
#ifdef KH_TIMING
    const long long tq1 = clock64();
#endif
    // COLS = 16: v_mfma_f64_16x16x4 (A: lane -> [row lane & 15][k lane >> 4]; B: [k lane >> 4][column lane & 15];
    //   C/D: column lane & 15, row (lane >> 4) + 4 reg).
    // COLS = 4: v_mfma_f64_4x4x4_4b, four independent 4x4x4 blocks per instruction, one per group of four rows
    //   (layout probed with scripts/ubench_mfma4.hip: A lane 16 k + 4 blk + row -- the SAME lanes as the 16x16x4
    //   A operand with row index 4 blk + row --, B lane 16 k + 4 blk + column, D lane 16 row + 4 blk + column).
    //   Exactly the 16 x 4 x 4 product of a k-step at a quarter of the 16x16x4 issue time.
    kh_d4 acc_r = {0.0, 0.0, 0.0, 0.0}, acc_i = {0.0, 0.0, 0.0, 0.0};
    double acc4_r = 0.0, acc4_i = 0.0;
#pragma unroll
    for (int q = 0; q < MAXKS; ++q) {
        if (q < c.ks) {
            const int j = q / KPL;
            int h[4];  // the payload halves: re hi, re lo, im hi, im lo
#pragma unroll
            for (int i = 0; i < 4; ++i) h[i] = (int)(unsigned int)(g[j][i] & 0xffffffffull);
            if constexpr (COLS != 16) {
                // element [k-step q, k = lane >> 4][column lane & 3] sits in load lane (k offset) * COLS + column
                const int src = (((q % KPL) * 4 + (lane >> 4)) * COLS + (lane & (COLS - 1))) * 4;  // byte address
#pragma unroll
                for (int i = 0; i < 4; ++i) h[i] = __builtin_amdgcn_ds_bpermute(src, h[i]);
            }
            const double vr = __hiloint2double(h[0], h[1]), vi = __hiloint2double(h[2], h[3]);
            const double fr = f.re(q), fi = f.im(q);
            if constexpr (COLS == 4) {
                acc4_r = __builtin_amdgcn_mfma_f64_4x4x4f64(fr, vr, acc4_r, 0, 0, 0);
                acc4_r = __builtin_amdgcn_mfma_f64_4x4x4f64(fi, -vi, acc4_r, 0, 0, 0);
                acc4_i = __builtin_amdgcn_mfma_f64_4x4x4f64(fr, vi, acc4_i, 0, 0, 0);
                acc4_i = __builtin_amdgcn_mfma_f64_4x4x4f64(fi, vr, acc4_i, 0, 0, 0);
            } else {
                acc_r = __builtin_amdgcn_mfma_f64_16x16x4f64(fr, vr, acc_r, 0, 0, 0);
                acc_r = __builtin_amdgcn_mfma_f64_16x16x4f64(fi, -vi, acc_r, 0, 0, 0);
                acc_i = __builtin_amdgcn_mfma_f64_16x16x4f64(fr, vi, acc_i, 0, 0, 0);
                acc_i = __builtin_amdgcn_mfma_f64_16x16x4f64(fi, vr, acc_i, 0, 0, 0);
            }
        }
    }
    if constexpr (COLS == 4) {
        part[(wave * 8 + 0) * 64 + lane] = acc4_r;
        part[(wave * 8 + 1) * 64 + lane] = acc4_i;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            part[(wave * 8 + i) * 64 + lane] = acc_r[i];
            part[(wave * 8 + 4 + i) * 64 + lane] = acc_i[i];
        }
    }
#ifdef KH_TIMING
    const long long tq2 = clock64();
#endif
    __syncthreads();
#ifdef KH_TIMING
    const long long tq3 = clock64();
    if (tid == 0 && blockIdx.x == 0) {
        s.tim[0] += (double)(tq1 - tq0);
        s.tim[1] += (double)(tq2 - tq1);
        s.tim[2] += (double)(tq3 - tq2);
        s.tim[3] += (double)spins;
    }
#endif
    w = c_make(0.0, 0.0);
    if (tid < 16 * COLS) {
        const int r = tid / COLS, oc = tid % COLS;  // owner of element (row r, column oc)
        if constexpr (COLS == 4) {
            const int src = 16 * (r & 3) + 4 * (r >> 2) + oc;  // D lane of row 4 blk + row', column
#pragma unroll
            for (int wv = 0; wv < KH_COOP_WAVES; ++wv) {
                w.x += part[(wv * 8 + 0) * 64 + src];
                w.y += part[(wv * 8 + 1) * 64 + src];
            }
        } else {
            const int src = (r & 3) * 16 + oc, reg = r >> 2;
#pragma unroll
            for (int wv = 0; wv < KH_COOP_WAVES; ++wv) {
                w.x += part[(wv * 8 + reg) * 64 + src];
                w.y += part[(wv * 8 + 4 + reg) * 64 + src];
            }
        }
    }
    __syncthreads();  // part[] is free for the next round
}

// The same round for 4 objectives per workgroup.  A wave's share of the k range is `count` groups of 16 columns
// (kh_coop4_share); a group's 16 x 4 block of the term is ONE element per lane -- lane (hi, b, lo) <- row
// 16 (start + gi) + 4 b + hi, column lo -- which is exactly the B operand of v_mfma_f64_4x4x4_4b when its four
// 4x4x4 blocks b take four consecutive k-steps (the operator fragments are stored to match, kh_coop_permute_kernel):
// no replication across blocks, no ds_bpermute (208 of them per wave and round in the first version: 1.2 us).
// Four row groups rb = 0..3 need four MFMA sets per group; the blocks' partial sums are added with two row
// rotations, and the wave's 16 x 4 block goes to LDS as 64 values per component: element (row r, column c) at 4 r + c.
struct KhNoHook {
    __device__ __forceinline__ void operator()() const {}
};
// hook: the caller's work in the shadow of the block fetch -- called once the fetch's loads are issued, before the first
// look at their tags; hook2: called when the blocks have arrived, in front of the matrix-core phase (loads issued there
// arrive under it).  The plain sweeps' fragment updates: kh_coop_expm_action_sq_ahead.
template <int MAXKS, int COLS, class Frag, class Hook = KhNoHook, class Hook2 = KhNoHook>
__device__ __forceinline__ void kh_coop_round4(const KhCoopArgs &c, const KhExchange &ex, unsigned int rid, int y, int N,
                                               const Frag &f, KhCoopLds &s, int tid, int wave, int lane, cplx &w,
                                               const Hook &hook = Hook(), const Hook2 &hook2 = Hook2()) {
    constexpr int MAXG = MAXKS / 4;  // groups per wave
    // (2 objectives: double-buffered by round parity -- the next round's barrier orders the reuse)
    double *part = kh_coop_part(s, c.ks) + (COLS == 2 ? (rid & 1u) * (KH_COOP_WAVES * KH_COOP_PART2) : 0);
    const unsigned int epoch = c.epoch_base + rid;
    int start, count;
    kh_coop4_share(N, wave, &start, &count);
    count = __builtin_amdgcn_readfirstlane(count);
    // operator elements of the first group: read from LDS while the block is on its way (the rest, one group ahead
    // of the matrix-core instructions that use them: left to itself the compiler waits for every read in turn)
    cplx fq[2][4];  // (two buffers, by group parity)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) fq[0][rb] = f.get(rb);
    const int roff = 4 * ((lane >> 2) & 3) + (lane >> 4);  // this lane's row of a group (column lane & 3)
    constexpr int NGR = COLS == 4 ? 4 : 2;         // granules of a lane's operand element
    constexpr unsigned int GB = COLS == 4 ? 2048u : 1024u;  // bytes of a group in the ring
    kh_u64 g[MAXG][NGR];
    long long t0 = 0;  // (taken when the slow path is entered: s_memrealtime is not free)
#ifdef KH_TIMING
    const long long tq0 = clock64();
#endif
    unsigned int spins = 0;
    for (int d = 0; d < c.first_poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
    // The whole ring as one buffer resource (KH_COOP_RING * Y * G * 2 KiB < 2 GiB: checked at engine creation);
    // lane offsets inside a group are compile-time + 16 lane.
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)c.vbuf, 0, (int)((size_t)KH_COOP_RING * c.Y * c.G * GB), 0x00020000);
    const unsigned int ring_off = (unsigned int)(kh_coop_group4(c, rid, y, 0) * 8 / (2048u / GB)) + 16u * lane;
    // first pass through L2 (fast; may see a stale line).  No branch per group: a group this lane has no row in (beyond
    // the wave's share, or beyond N in the last one) is fetched from beyond the buffer -- the load returns zeros without
    // touching memory: value +0.0 -- and counts as fresh.
    const int jv = min(count, (N - roff - 16 * start + 15) >> 4);  // this lane's groups j < jv exist
#ifndef KH_COOP_X_NOLOAD
    if (c.local) {  // one XCD: its L2 has the producers' stores (agent-scope loads bypass only the L1)
#pragma unroll
        for (int j = 0; j < MAXG; ++j) {
            const unsigned int off = j < jv ? ring_off + GB * (start + j) : 0xffffff00u;
            kh_coop_load2<KH_CPOL_SC1>(rsrc, off, g[j][0], g[j][1]);
            if constexpr (COLS == 4) kh_coop_load2<KH_CPOL_SC1>(rsrc, off + (j < jv ? 1024u : 0u), g[j][2], g[j][3]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < MAXG; ++j) {
            const unsigned int off = j < jv ? ring_off + GB * (start + j) : 0xffffff00u;
            kh_coop_load2<KH_CPOL_SC0>(rsrc, off, g[j][0], g[j][1]);
            if constexpr (COLS == 4) kh_coop_load2<KH_CPOL_SC0>(rsrc, off + (j < jv ? 1024u : 0u), g[j][2], g[j][3]);
        }
    }
#else
#pragma unroll
    for (int j = 0; j < MAXG; ++j)
#pragma unroll
        for (int i = 0; i < NGR; ++i) g[j][i] = (kh_u64)epoch << 32;
#endif
    hook();
    // (bitwise, not &&: a short-circuit chain over the tags compiles to nested branches)
    unsigned int bad_tags = 0u;
#pragma unroll
    for (int j = 0; j < MAXG; ++j) {
        unsigned int bj = 0u;
#pragma unroll
        for (int i = 0; i < NGR; ++i) bj |= (unsigned int)(g[j][i] >> 32) ^ epoch;
        bad_tags |= j < jv ? bj : 0u;
    }
    bool all_fresh = bad_tags == 0u;
#ifdef KH_TIMING
    {
        const int stale_lanes = __popcll(__ballot(!all_fresh));
        if (tid == 0 && blockIdx.x == 0) s.tim[5] += (double)stale_lanes;
    }
    const long long tqf = clock64();
    if (tid == 0 && blockIdx.x == 0) s.tim[4] += (double)(tqf - tq0);
#endif
#ifdef KH_COOP_X_NOPOLL  // (timing experiment: wrong results)
    all_fresh = true;
#endif
    // otherwise: only what was stale, bypassing L2, until everything carries the round's tag
    while (!__all(all_fresh)) {
#ifdef KH_TIMING
        if (tid == 0 && blockIdx.x == 0) s.tim[6] += 1.0;
#endif
        kh_compiler_fence();
        bool ok = true;
#pragma unroll
        for (int j = 0; j < MAXG; ++j) {
            if (j < jv) {
                const unsigned int off = ring_off + GB * (start + j);
                if ((unsigned int)(g[j][0] >> 32) != epoch || (unsigned int)(g[j][1] >> 32) != epoch)
                    kh_coop_load2<KH_CPOL_SC1>(rsrc, off, g[j][0], g[j][1]);
                if constexpr (COLS == 4) {
                    if ((unsigned int)(g[j][2] >> 32) != epoch || (unsigned int)(g[j][3] >> 32) != epoch)
                        kh_coop_load2<KH_CPOL_SC1>(rsrc, off + 1024u, g[j][2], g[j][3]);
                }
            }
        }
        unsigned int bad2 = 0u;
#pragma unroll
        for (int j = 0; j < MAXG; ++j) {
            unsigned int bj = 0u;
#pragma unroll
            for (int i = 0; i < NGR; ++i) bj |= (unsigned int)(g[j][i] >> 32) ^ epoch;
            bad2 |= j < jv ? bj : 0u;
        }
        ok = bad2 == 0u;
        all_fresh = ok;
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(2);
        if (spins == 0) t0 = wall_clock64();
        if ((++spins & 63u) == 0) {
            const bool gave_up =
                (wall_clock64() - t0 > ex.timeout_ticks) ||
                (__hip_atomic_load(ex.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u);
            if (__any(gave_up)) {
                if (lane == 0) {
                    __hip_atomic_store(ex.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s.abort = 1;
                }
                break;
            }
        }
    }
#ifdef KH_TIMING
    const long long tq1 = clock64();
#endif
    hook2();
    double ar[4] = {0.0, 0.0, 0.0, 0.0}, ai[4] = {0.0, 0.0, 0.0, 0.0};  // one accumulator pair per row group
    if constexpr (COLS == 4) {
#pragma unroll
        for (int j = 0; j < MAXG; ++j) {
            if (j < count) {
                if (j + 1 < MAXG) {
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb) fq[(j + 1) & 1][rb] = f.get(4 * (j + 1) + rb);
                }
                const double vr = __hiloint2double((int)(unsigned int)(g[j][0] & 0xffffffffull), (int)(unsigned int)(g[j][1] & 0xffffffffull));
                const double vi = __hiloint2double((int)(unsigned int)(g[j][2] & 0xffffffffull), (int)(unsigned int)(g[j][3] & 0xffffffffull));
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    if (!f.nz(4 * j + rb)) continue;
                    const double fr = fq[j & 1][rb].x, fi = fq[j & 1][rb].y;
                    ar[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(fr, vr, ar[rb], 0, 0, 0);
                    ar[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(fi, -vi, ar[rb], 0, 0, 0);
                    ai[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(fr, vi, ai[rb], 0, 0, 0);
                    ai[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(fi, vr, ai[rb], 0, 0, 0);
                }
            }
        }
        // D lane = 16 (row within the group) + 4 (block = k-step) + column: add the four blocks, one lane of four writes
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            ar[rb] += dpp_move<KH_DPP_ROR8>(ar[rb]);
            ar[rb] += dpp_move<KH_DPP_ROR4>(ar[rb]);
            ai[rb] += dpp_move<KH_DPP_ROR8>(ai[rb]);
            ai[rb] += dpp_move<KH_DPP_ROR4>(ai[rb]);
            if (((lane >> 2) & 3) == 0) {
                const int e = 16 * rb + 4 * (lane >> 4) + (lane & 3);
                part[(wave * 2 + 0) * 64 + e] = ar[rb];
                part[(wave * 2 + 1) * 64 + e] = ai[rb];
            }
        }
    } else {
        // 2 objectives: the operand's four columns are [re c0, re c1, im c0, im c1], so Re(F) and Im(F) need ONE
        // MFMA each per group and row block (half the matrix-core work per objective of the 4-column form, which
        // spends four); Re(F) X gives [Fr Xr | Fr Xi], Im(F) X gives [Fi Xr | Fi Xi]: the result is
        // [Fr Xr - Fi Xi | Fr Xi + Fi Xr] = first + (-1, +1) * (second with its column pairs swapped).
#pragma unroll
        for (int j = 0; j < MAXG; ++j) {
            if (j < count) {
                if (j + 1 < MAXG) {
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb) fq[(j + 1) & 1][rb] = f.get(4 * (j + 1) + rb);
                }
                const double v = __hiloint2double((int)(unsigned int)(g[j][0] & 0xffffffffull), (int)(unsigned int)(g[j][1] & 0xffffffffull));
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    if (!f.nz(4 * j + rb)) continue;
#ifdef KH_COOP_X_NOMFMA  // (timing experiment: wrong results)
                    ar[rb] += v * fq[j & 1][rb].x;
                    continue;
#endif
                    ar[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(fq[j & 1][rb].x, v, ar[rb], 0, 0, 0);
                    ai[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(fq[j & 1][rb].y, v, ai[rb], 0, 0, 0);
                }
            }
        }
        const double sgn = (lane & 2) ? 1.0 : -1.0;
#ifdef KH_COOP_X_NOSUM  // (timing experiment: wrong results) no block sums, no LDS partial sums, no barrier
        w = c_make(ar[0] + ar[1] + ar[2] + ar[3], ai[0] + ai[1] + ai[2] + ai[3]);
        return;
#endif
        // (the four row blocks' chains step by step side by side, not one after the other: each step is a DPP move whose
        // result the next instruction needs)
        double e4[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) e4[rb] = fma(sgn, dpp_move<KH_DPP_XOR2>(ai[rb]), ar[rb]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) e4[rb] += dpp_move<KH_DPP_ROR8>(e4[rb]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) e4[rb] += dpp_move<KH_DPP_ROR4>(e4[rb]);
        if (((lane >> 2) & 3) == 0) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) part[wave * KH_COOP_PART2 + 16 * rb + 4 * (lane >> 4) + (lane & 3)] = e4[rb];
        }
    }
#ifdef KH_TIMING
    const long long tq2 = clock64();
#endif
    __syncthreads();
#ifdef KH_TIMING
    const long long tq3 = clock64();
    if (tid == 0 && blockIdx.x == 0) {
        s.tim[0] += (double)(tq1 - tq0);
        s.tim[1] += (double)(tq2 - tq1);
        s.tim[2] += (double)(tq3 - tq2);
        s.tim[3] += (double)spins;
    }
#endif
    w = c_make(0.0, 0.0);
    if constexpr (COLS == 4) {
        if (tid < 64) {  // owner of element (row tid / 4, column tid % 4)
#pragma unroll
            for (int wv = 0; wv < KH_COOP_WAVES; ++wv) {
                w.x += part[(wv * 2 + 0) * 64 + tid];
                w.y += part[(wv * 2 + 1) * 64 + tid];
            }
        }
    } else {
        // element [4 row + n] of the wave vectors (n: re c0, re c1, im c0, im c1).  Lane (group lane >> 3, wave
        // lane & 7) reads ONE partial value of element 4 (2 wave + group / 4) + (group / 2) % 2 + 2 (group % 2); the
        // eight waves' values sit in eight neighbouring lanes (sum8), and the im partner of a re group is the next
        // group, 8 lanes on in the same 16-lane row.  Owners: lanes 0, 16, 32, 48 (kh_coop_owner_element).
        const int grp = lane >> 3;
        const int e = 4 * (2 * wave + (grp >> 2)) + ((grp >> 1) & 1) + 2 * (grp & 1);
        const double v = sum8(part[(lane & 7) * KH_COOP_PART2 + e]);
        const double partner = dpp_move<KH_DPP_ROR8>(v);
        if ((lane & 15) == 0) w = c_make(v, partner);
    }
#ifdef KH_TIMING
    const long long tq4 = clock64();
#endif
    if constexpr (COLS != 2) __syncthreads();  // part[] is free for the next round
#ifdef KH_TIMING
    if (tid == 0 && blockIdx.x == 0) {
        const long long tq5 = clock64();
        s.tim[7] += (double)(tq4 - tq3);
        s.tim[8] += (double)(tq5 - tq4);
        if (s.tim[11] != 0.0) s.tim[9] += (double)tq0 - s.tim[11];
        s.tim[11] = (double)tq5;
    }
#endif
}

template <int MAXKS, int COLS, class Frag, class Hook = KhNoHook, class Hook2 = KhNoHook>
__device__ __forceinline__ void kh_coop_round(const KhCoopArgs &c, const KhExchange &ex, unsigned int rid, int y,
                                              int N, const Frag &f, KhCoopLds &s, int tid, int wave, int lane,
                                              cplx &w, const Hook &hook = Hook(), const Hook2 &hook2 = Hook2()) {
    if constexpr (COLS <= 4)
        kh_coop_round4<MAXKS, COLS>(c, ex, rid, y, N, f, s, tid, wave, lane, w, hook, hook2);
    else {
        hook2();  // (16 objectives per workgroup: in the open, in front of the round)
        hook();
        kh_coop_round16<MAXKS, COLS>(c, ex, rid, y, N, f, s, tid, wave, lane, w);
    }
}

// register fragment <- / += eps * (fragment-ordered operator)
template <int MAXKS>
__device__ __forceinline__ void kh_coop_reg_load(const cplx *op, int g, int wave, int lane, int ks, cplx (&r)[MAXKS],
                                                 unsigned int mask = ~0u) {
    const KhCoopSrc src = kh_coop_frag_src(op, g, wave, lane, ks);
#pragma unroll
    for (int q = 0; q < MAXKS; ++q)
        r[q] = (q < ks && !src.null() && ((mask >> q) & 1u)) ? src[(size_t)q * 64] : c_make(0.0, 0.0);
}
template <int MAXKS>
__device__ __forceinline__ void kh_coop_reg_axpy(const cplx *op, double eps, int g, int wave, int lane, int ks,
                                                 cplx (&r)[MAXKS], unsigned int mask = ~0u) {
    const KhCoopSrc src = kh_coop_frag_src(op, g, wave, lane, ks);
    if (src.null()) return;
    // Slots are read in chunks of KH_COOP_CHUNK: all loads of a chunk are issued before the first is used.  One slot at
    // a time -- what a per-slot mask test compiles to, every load in a basic block of its own -- exposes the full L2
    // latency once per slot: 16 x 0.3 us = the 4.8 us per interval the dense table read used to cost (the L2 hit rate
    // is 97 %, the traffic 21 GB/s per CU: neither bandwidth nor misses).  A chunk is skipped if all its slots are zero
    // (tables are zero-padded to whole chunks; zero slots inside a chunk are read and add nothing).
#pragma unroll
    for (int c0 = 0; c0 < MAXKS; c0 += KH_COOP_CHUNK) {
        if (c0 < ks && ((mask >> c0) & ((1u << KH_COOP_CHUNK) - 1u)) != 0u) {
            cplx v[KH_COOP_CHUNK];
#pragma unroll
            for (int j = 0; j < KH_COOP_CHUNK; ++j)  // (slots >= ks do not exist: clamped address, value dropped)
                v[j] = src[(size_t)(c0 + j < ks ? c0 + j : ks - 1) * 64];
#pragma unroll
            for (int j = 0; j < KH_COOP_CHUNK; ++j) {
                if (c0 + j >= ks) v[j] = c_make(0.0, 0.0);
                r[c0 + j].x = fma(eps, v[j].x, r[c0 + j].x);
                r[c0 + j].y = fma(eps, v[j].y, r[c0 + j].y);
            }
        }
    }
}

// r += er * opr and (LDS fragment) a += ea * opa, chunk by chunk, BOTH tables' reads of a chunk (and the fragment's LDS reads)
// issued before any is used: for a banded control operator the non-zero chunk of P2 and of H1 is the same chunk of the
// same wave -- one trip to L2 instead of two in a row (a chunk is read if either table has something in it: the tables
// are zero-padded, a zero chunk adds nothing)
template <int MAXKS>
__device__ __forceinline__ void kh_coop_axpy_pair(const cplx *opr, double er, cplx (&r)[MAXKS], const cplx *opa, double ea,
                                                  const KhCoopFrag &a, int g, int wave, int lane, int ks, unsigned int mask) {
    const KhCoopSrc sr = kh_coop_frag_src(opr, g, wave, lane, ks), sa = kh_coop_frag_src(opa, g, wave, lane, ks);
#pragma unroll
    for (int c0 = 0; c0 < MAXKS; c0 += 4) {
        if (c0 < ks && ((mask >> c0) & 0xfu) != 0u) {
            cplx vr[4], va[4], t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // (slots >= ks do not exist: clamped address, value dropped; a NULL table reads zeros)
                const int q = c0 + j < ks ? c0 + j : ks - 1;
                vr[j] = sr[(size_t)q * 64];
                va[j] = sa[(size_t)q * 64];
                t[j] = a.get(c0 + j < MAXKS ? c0 + j : MAXKS - 1);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (c0 + j < ks) {
                    r[c0 + j].x = fma(er, vr[j].x, r[c0 + j].x);
                    r[c0 + j].y = fma(er, vr[j].y, r[c0 + j].y);
                    a.f[(size_t)(c0 + j) * KH_COOP_THREADS] = c_make(fma(ea, va[j].x, t[j].x), fma(ea, va[j].y, t[j].y));
                }
            }
        }
    }
}

// A = op_0 + sum_l eps_l op_l (fragments rebuilt from L2 once per interval; ops: fragment-ordered copies)
template <int MAXKS>
__device__ __forceinline__ void kh_coop_build(const cplx *const *ops, const double *eps, int L, int g, int wave,
                                              int lane, int ks, const KhCoopFrag &a) {
    kh_coop_load_frag<MAXKS>(ops[0], g, wave, lane, ks, a);
#pragma unroll
    for (int l = 0; l < KH_COOP_MAX_L; ++l)
        if (l < L) kh_coop_axpy_frag<MAXKS>(ops[1 + l], eps[l], g, wave, lane, ks, a);
}

// state <- exp(f A dt) state, term by term; round `rid` holds the state on entry and on exit.
// Owner threads carry `state`; rid is advanced by the number of rounds.  Returns false if the
// exchange timed out.
template <int MAXKS, int COLS>
__device__ __forceinline__ bool kh_coop_expm_action(const KhCoopArgs &c, const KhExchange &ex,
                                                    const KhCoopFrag &a, cplx &state, unsigned int &rid,
                                                    KhCoopLds &s, int N, int y, int row, int col,
                                                    bool owner_valid, double fre, double fim, double dt, int nsub,
                                                    int m, int tid, int wave, int lane) {
    const double h = nsub == 1 ? dt : dt / nsub;
    for (int sub = 0; sub < nsub; ++sub) {
        for (int j = 1; j <= m; ++j) {
            cplx w;
            kh_coop_round<MAXKS, COLS>(c, ex, rid, y, N, a, s, tid, wave, lane, w);
            if (s.abort) return false;  // (raised before the barriers inside kh_coop_round)
            if (kh_coop_is_owner<COLS>(tid)) {
                const double hj = h / j;
                const cplx t = c_mul(c_make(fre * hj, fim * hj), w);
                state.x += t.x;
                state.y += t.y;
                if (owner_valid) kh_coop_publish<COLS>(c, rid + 1, y, row, col, j == m ? state : t, c.local != 0);
            }
            ++rid;
        }
    }
    return true;
}

// The same with the square of the generator as the chain operator (one control: A^2 = P0 + eps P1 +
// eps^2 P2 with three fixed matrices staged by kh_engine_create).  The even terms are a chain of products
// with B = A^2, t_{2p+2} = f^2 h^2 / ((2p+1)(2p+2)) B t_{2p}; the odd terms only enter the state sum and A is
// linear, sum_p t_{2p+1} = f A s with s = sum_p h/(2p+1) t_{2p}: every owner accumulates its element of s
// while the even terms go by, the last B round publishes s instead of a term, and ONE round with the A
// fragment finishes the step.  ceil(m/2) + 1 rounds (each one cross-workgroup exchange) instead of m; the
// fragment in LDS is rebuilt twice per step (B, then A) from L2.
// The fragments are not rebuilt per interval but advanced: B (registers: it is the operand of all rounds of the
// interval but one, and reading it from LDS costs 131 KiB of LDS traffic per round -- 0.5 us, as much as the
// matrix-core instructions themselves and badly overlapped with them) += (eps - eps') P1 + (eps^2 - eps'^2) P2 and
// A (LDS: one round per interval) += (eps - eps') H1 -- three table reads per interval instead of five; the caller restarts them from
// P0 / H0 (kh_coop_sq_restart) every KH_COOP_REFRESH intervals, so rounding cannot drift.
#define KH_COOP_REFRESH 64
struct KhCoopSqMasks {
    unsigned int h0, h1, p0, p1, p2;  // zero-slot masks of this wave's fragments (kh_coop_mask_kernel)
};
__device__ __forceinline__ KhCoopSqMasks kh_coop_sq_masks(const KhCoopArgs &c, int g, int wave) {
    KhCoopSqMasks m = {0u, 0u, 0u, 0u, 0u};
    if (c.sq != nullptr) {
        m.h0 = kh_coop_frag_mask(c.tab[0], c.G, g, wave, c.ks);
        m.h1 = kh_coop_frag_mask(c.tab[1], c.G, g, wave, c.ks);
        m.p0 = kh_coop_frag_mask(c.tab[2], c.G, g, wave, c.ks);
        m.p1 = kh_coop_frag_mask(c.tab[3], c.G, g, wave, c.ks);
        m.p2 = kh_coop_frag_mask(c.tab[4], c.G, g, wave, c.ks);
    }
    return m;
}
template <int MAXKS>
__device__ __forceinline__ void kh_coop_sq_restart(const KhCoopArgs &c, const KhCoopSqMasks &mk, int g, int wave,
                                                   int lane, const KhCoopFrag &a, cplx (&breg)[MAXKS],
                                                   double &eps_prev) {
    kh_coop_reg_load<MAXKS>(c.tab[2], g, wave, lane, c.ks, breg, mk.p0);
    kh_coop_load_frag<MAXKS>(c.tab[0], g, wave, lane, c.ks, a, mk.h0);
    eps_prev = 0.0;
}

template <int MAXKS, int COLS>
__device__ __forceinline__ bool kh_coop_expm_action_sq(const KhCoopArgs &c, const KhExchange &ex,
                                                       const KhCoopSqMasks &mk, double eps, double &eps_prev, const KhCoopFrag &a,
                                                       cplx (&breg)[MAXKS],
                                                       cplx &state, unsigned int &rid, KhCoopLds &s, int N, int y,
                                                       int g, int row, int col, bool owner_valid, double fre,
                                                       double fim, double dt, int nsub, int m, int tid, int wave,
                                                       int lane, const cplx (*p1pre)[MAXKS] = nullptr) {
    // p1pre: this lane's slots of the P1 table, fetched by the caller while it waited for eps (zero slots hold zeros)
    // (wave-uniform scalars are kept in SGPRs: the VGPR file is full of operator fragments)
    const double h = kh_uniform(nsub == 1 ? dt : dt / nsub);
    const double f2h2 = kh_uniform((fre * fre - fim * fim) * h * h);  // f is purely real or purely imaginary
    const int phases = (m + 1) >> 1;
    // series coefficients of degree m: sum_j c_j (f h A)^j, rows[p] = {c_{2p+1}/c_{2p}, c_{2p+2}/c_{2p}} (p = 0: c_1, c_2)
    // (copied to LDS once per interval: a scalar load from the table in every phase sat exposed in the round's chain)
    const bool series = c.ser_rows != nullptr;
    if (series) {
        const double *grow = c.ser_rows + (size_t)m * KH_Q2_ROWS * 2;
        if (tid < 2 * phases) s.coef[tid] = grow[tid];
        if (tid == 2 * phases) s.coef[tid] = c.ser_c0[m];
        __syncthreads();
    }
    const double *rows = series ? s.coef : nullptr;
    const double c_0 = kh_uniform(series ? s.coef[2 * phases] : 1.0), c_1 = kh_uniform(series ? s.coef[0] : 1.0);
#ifndef KH_COOP_X_NOREBUILD  // (timing experiment: wrong results)
    {
#ifdef KH_TIMING
        const long long tr0 = clock64();
#endif
        const double e1 = eps - eps_prev, e2 = e1 * (eps + eps_prev);
        if (p1pre != nullptr) {
            // (the prefetched P1 behind the two sparse tables: their reads are on their way while its 128 FMAs run)
            kh_coop_axpy_pair<MAXKS>(c.tab[4], e2, breg, c.tab[1], e1, a, g, wave, lane, c.ks, mk.p2 | mk.h1);
#pragma unroll
            for (int q = 0; q < MAXKS; ++q) {
                breg[q].x = fma(e1, (*p1pre)[q].x, breg[q].x);
                breg[q].y = fma(e1, (*p1pre)[q].y, breg[q].y);
            }
        } else {
            kh_coop_reg_axpy<MAXKS>(c.tab[3], e1, g, wave, lane, c.ks, breg, mk.p1);
            kh_coop_reg_axpy<MAXKS>(c.tab[4], e2, g, wave, lane, c.ks, breg, mk.p2);
            kh_coop_axpy_frag<MAXKS>(c.tab[1], e1, g, wave, lane, c.ks, a, mk.h1);
        }
        eps_prev = kh_uniform(eps);
#ifdef KH_TIMING
        if (tid == 0 && blockIdx.x == 0) {
            const long long tr1 = clock64();
            s.tim[10] += (double)(tr1 - tr0);
            if (s.tim[11] != 0.0) s.tim[11] += (double)(tr1 - tr0);  // (not counted as "between rounds")
        }
#endif
    }
#endif
    const KhCoopRegFrag<MAXKS> bf = {breg, ~0u};
    for (int sub = 0; sub < nsub; ++sub) {
        cplx sacc = c_make(h * c_1 * state.x, h * c_1 * state.y);
        state = c_make(c_0 * state.x, c_0 * state.y);  // (the published block of this round is the unscaled state)
        for (int ph = 0; ph < phases; ++ph) {
            // (the coefficients are fetched before the round, not between its end and the owners' stores; a timed-out
            // round is noticed after the phases: the later rounds give up at once on the abort flag)
            // (read before the round, used after it)
            // (plain loads, NOT kh_uniform: a read-first-lane would wait for the LDS read here, in front of the round's
            // block fetch; as it is the read's latency hides under the round)
            const double r2 = rows != nullptr ? rows[2 * ph + 1] : kh_inv_table[2 * ph + 1] * kh_inv_table[2 * ph + 2];
            const double r1n = ph + 1 < phases ? (rows != nullptr ? rows[2 * ph + 2] : kh_inv_table[2 * ph + 3]) : 0.0;
            cplx w;
            kh_coop_round<MAXKS, COLS>(c, ex, rid, y, N, bf, s, tid, wave, lane, w);
            const double c2 = f2h2 * r2, hn = h * r1n;
            if (kh_coop_is_owner<COLS>(tid)) {
                const cplx t2 = c_make(c2 * w.x, c2 * w.y);
                state.x += t2.x;
                state.y += t2.y;
                const bool last = (ph + 1 == phases);
                if (!last) {
                    sacc.x = fma(hn, t2.x, sacc.x);
                    sacc.y = fma(hn, t2.y, sacc.y);
                }
                if (owner_valid) kh_coop_publish<COLS>(c, rid + 1, y, row, col, last ? sacc : t2, c.local != 0);
            }
            ++rid;
        }
        cplx w;
        kh_coop_round<MAXKS, COLS>(c, ex, rid, y, N, a, s, tid, wave, lane, w);
        if (s.abort) return false;
        if (kh_coop_is_owner<COLS>(tid)) {
            const cplx odd = c_mul(c_make(fre, fim), w);
            state.x += odd.x;
            state.y += odd.y;
            if (owner_valid) kh_coop_publish<COLS>(c, rid + 1, y, row, col, state, c.local != 0);
        }
        ++rid;
    }
    return true;
}

// The same step for the PLAIN sweeps, whose pulse values are known in advance: the fragment updates leave the interval's
// serial chain.  The dense P1 table -- 102 KB per workgroup: 1 600 cycles of the CU's vector-memory path, whatever its
// latency -- arrives a quarter per round UNDER the matrix-core phases of the step's last four B rounds (requested when a
// round's blocks are in: hook2); B (registers) is moved to the NEXT interval's pulse value inside the step's last round
// -- the A round, which does not read B -- between the issue of that round's block fetch and the first look at the tags
// (hook); A (LDS) follows one round later, inside the next step's first round (a B round, which does not read A).
// eps_b / eps_a: the pulse values B and A stand at (0 after kh_coop_sq_restart); a B that is not at `eps` on entry --
// first interval, restart -- is brought there in the open, as in kh_coop_expm_action_sq.
template <int MAXKS, int COLS>
__device__ __forceinline__ bool kh_coop_expm_action_sq_ahead(const KhCoopArgs &c, const KhExchange &ex, const KhCoopSqMasks &mk,
                                                             double eps, double eps_next, bool have_next, double &eps_b,
                                                             double &eps_a, const KhCoopFrag &a, cplx (&breg)[MAXKS],
                                                             cplx &state, unsigned int &rid, KhCoopLds &s, int N, int y, int g,
                                                             int row, int col, bool owner_valid, double fre, double fim,
                                                             double dt, int nsub, int m, int tid, int wave, int lane) {
    static_assert(MAXKS % 4 == 0, "quarters");
    const double h = kh_uniform(nsub == 1 ? dt : dt / nsub);
    const double f2h2 = kh_uniform((fre * fre - fim * fim) * h * h);
    const int phases = (m + 1) >> 1;
    const bool series = c.ser_rows != nullptr;
    if (series) {
        const double *grow = c.ser_rows + (size_t)m * KH_Q2_ROWS * 2;
        if (tid < 2 * phases) s.coef[tid] = grow[tid];
        if (tid == 2 * phases) s.coef[tid] = c.ser_c0[m];
        __syncthreads();
    }
    const double *rows = series ? s.coef : nullptr;
    const double c_0 = kh_uniform(series ? s.coef[2 * phases] : 1.0), c_1 = kh_uniform(series ? s.coef[0] : 1.0);
    if (eps_b != eps) {  // (wave-uniform; rare)
        const double e1 = eps - eps_b, e2 = e1 * (eps + eps_b);
        kh_coop_reg_axpy<MAXKS>(c.tab[3], e1, g, wave, lane, c.ks, breg, mk.p1);
        kh_coop_reg_axpy<MAXKS>(c.tab[4], e2, g, wave, lane, c.ks, breg, mk.p2);
        eps_b = kh_uniform(eps);
    }
    const auto advance_a = [&]() {
        if (eps_a != eps) {
            kh_coop_axpy_frag<MAXKS>(c.tab[1], eps - eps_a, g, wave, lane, c.ks, a, mk.h1);
            eps_a = kh_uniform(eps);
        }
    };
    const KhCoopRegFrag<MAXKS> bf = {breg, ~0u};
    const KhCoopSrc src1 = kh_coop_frag_src(c.tab[3], g, wave, lane, c.ks);
    constexpr int QS = MAXKS / 4;  // slots of a quarter
    for (int sub = 0; sub < nsub; ++sub) {
        const bool ahead = have_next && sub + 1 == nsub;
        const bool staged = ahead && phases >= 4;  // (otherwise: all of P1 requested in front of the A round)
        cplx p1n[MAXKS];
#pragma unroll
        for (int q = 0; q < MAXKS; ++q) p1n[q] = c_make(0.0, 0.0);
        cplx sacc = c_make(h * c_1 * state.x, h * c_1 * state.y);
        state = c_make(c_0 * state.x, c_0 * state.y);
        // one B round; hook2 (called when the round's blocks are in, in front of its matrix-core phase) requests a quarter
        // of P1 in the step's last four B rounds -- peeled below, so that the registers the quarter lands in are constants
        const auto b_round = [&](int ph, const auto &hook2) {
            const double r2 = rows != nullptr ? rows[2 * ph + 1] : kh_inv_table[2 * ph + 1] * kh_inv_table[2 * ph + 2];
            const double r1n = ph + 1 < phases ? (rows != nullptr ? rows[2 * ph + 2] : kh_inv_table[2 * ph + 3]) : 0.0;
            cplx w;
            kh_coop_round<MAXKS, COLS>(c, ex, rid, y, N, bf, s, tid, wave, lane, w, advance_a, hook2);
            const double c2 = f2h2 * r2, hn = h * r1n;
            if (kh_coop_is_owner<COLS>(tid)) {
                const cplx t2 = c_make(c2 * w.x, c2 * w.y);
                state.x += t2.x;
                state.y += t2.y;
                const bool last = (ph + 1 == phases);
                if (!last) {
                    sacc.x = fma(hn, t2.x, sacc.x);
                    sacc.y = fma(hn, t2.y, sacc.y);
                }
                if (owner_valid) kh_coop_publish<COLS>(c, rid + 1, y, row, col, last ? sacc : t2, c.local != 0);
            }
            ++rid;
        };
        const int plain_rounds = staged ? phases - 4 : phases;
        for (int ph = 0; ph < plain_rounds; ++ph) b_round(ph, KhNoHook());
        if (staged) {
            b_round(phases - 4, [&]() {
#pragma unroll
                for (int j = 0; j < QS; ++j) p1n[0 * QS + j] = src1[(size_t)(0 * QS + j) * 64];
            });
            b_round(phases - 3, [&]() {
#pragma unroll
                for (int j = 0; j < QS; ++j) p1n[1 * QS + j] = src1[(size_t)(1 * QS + j) * 64];
            });
            b_round(phases - 2, [&]() {
#pragma unroll
                for (int j = 0; j < QS; ++j) p1n[2 * QS + j] = src1[(size_t)(2 * QS + j) * 64];
            });
            b_round(phases - 1, [&]() {
#pragma unroll
                for (int j = 0; j < QS; ++j) p1n[3 * QS + j] = src1[(size_t)(3 * QS + j) * 64];
            });
        }
        advance_a();  // (a step without B rounds: degree 0 -- not reached by the tables, kept for safety)
        if (ahead && !staged) kh_coop_reg_load<MAXKS>(c.tab[3], g, wave, lane, c.ks, p1n, mk.p1);
        const auto advance_b = [&]() {
            if (ahead) {
                const double e1 = eps_next - eps_b, e2 = e1 * (eps_next + eps_b);
#pragma unroll
                for (int q = 0; q < MAXKS; ++q) {
                    breg[q].x = fma(e1, p1n[q].x, breg[q].x);
                    breg[q].y = fma(e1, p1n[q].y, breg[q].y);
                }
                kh_coop_reg_axpy<MAXKS>(c.tab[4], e2, g, wave, lane, c.ks, breg, mk.p2);
                eps_b = kh_uniform(eps_next);
            }
        };
        cplx w;
        kh_coop_round<MAXKS, COLS>(c, ex, rid, y, N, a, s, tid, wave, lane, w, advance_b);
        if (s.abort) return false;
        if (kh_coop_is_owner<COLS>(tid)) {
            const cplx odd = c_mul(c_make(fre, fim), w);
            state.x += odd.x;
            state.y += odd.y;
            if (owner_valid) kh_coop_publish<COLS>(c, rid + 1, y, row, col, state, c.local != 0);
        }
        ++rid;
    }
    return true;
}

// ---------------------------------------------------------------------------
// Update sums on the adjoint side (one control, first order)
// ---------------------------------------------------------------------------
// <chi_k(t_n) | H_1 phi_k(t_n)> = <H_1^+ chi_k(t_n) | phi_k(t_n)>, and the left factor does not depend on the running
// forward state: V = H_1^+ X for ALL stored co-states X = [chi_k(t_n)]_{k,n} (N x K nt) is ONE dense product in front
// of the update sweep -- off its serial chain -- instead of one cross-workgroup round per time interval inside it
// (1.9 us x (nt - 1), plus the 64 transient registers of the control operator's fragment).  In the sweep every owner
// thread then reads its element of V next to its element of the state: no round, no fragment.
//
// kh_coop_adj_mask_kernel marks the non-zero 16 x 16 blocks of H_1^+ once per engine; kh_coop_adjoint_side multiplies
// block-sparsely on the fp64 matrix cores (v_mfma_f64_16x16x4: A = operator block [row lane & 15][k lane >> 4],
// B = sixteen vectors [k lane >> 4][vector lane & 15], D = [row 4 reg + (lane >> 4)][vector lane & 15]).  A control
// that is a commutator with a diagonal operator (the transmon of BASELINE config 4) has ONE non-zero block per row
// block: the product is then a streaming pass over the store (read 16 N K nt bytes, write as many).
__global__ void kh_coop_adj_mask_kernel(const cplx *__restrict__ op /*row-major N x N*/, int N, int G,
                                        unsigned char *__restrict__ nz /*[G][G]*/)
#if KH_DEFINES(KH_TU_MAIN)
{
    const int g = blockIdx.x / G, kb = blockIdx.x % G;
    const int r = 16 * g + (threadIdx.x >> 4), col = 16 * kb + (threadIdx.x & 15);
    bool any = false;
    if (r < N && col < N) {
        const cplx v = op[(size_t)r * N + col];
        any = v.x != 0.0 || v.y != 0.0;
    }
    const int found = __syncthreads_or(any ? 1 : 0);
    if (threadIdx.x == 0) nz[blockIdx.x] = found ? 1 : 0;
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif

#define KH_COOP_ADJ_THREADS 256  // 4 waves x 16 vectors
__global__ void __launch_bounds__(KH_COOP_ADJ_THREADS)
kh_coop_adjoint_side(const cplx *__restrict__ op /*H_1^+, row-major N x N*/, const unsigned char *__restrict__ nz,
                     const cplx *__restrict__ X /*[M][N]*/, cplx *__restrict__ V /*[M][N]*/, int N, int G, long long M)
#if KH_DEFINES(KH_TU_COOP_STORE)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = lane & 15, kq = lane >> 4;
    const long long vec = ((long long)blockIdx.x * 4 + wave) * 16 + j;  // this lane's vector (B operand column)
    const bool vec_ok = vec < M;
    const cplx *x = X + (size_t)(vec_ok ? vec : 0) * N;
    cplx *v = V + (size_t)(vec_ok ? vec : 0) * N;
    for (int g = 0; g < G; ++g) {
        kh_d4 dr = {0.0, 0.0, 0.0, 0.0}, di = {0.0, 0.0, 0.0, 0.0};
        const int arow = 16 * g + j;  // A operand: row lane & 15
        for (int kb = 0; kb < G; ++kb) {
            if (!nz[g * G + kb]) continue;  // (uniform over the grid)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int kk = 16 * kb + 4 * ks + kq;
                cplx a = c_make(0.0, 0.0), b = c_make(0.0, 0.0);
                if (kk < N) {
                    if (arow < N) a = op[(size_t)arow * N + kk];
                    if (vec_ok) b = x[kk];
                }
                dr = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, dr, 0, 0, 0);
                dr = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, -b.y, dr, 0, 0, 0);
                di = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.y, di, 0, 0, 0);
                di = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.x, di, 0, 0, 0);
            }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * g + 4 * reg + kq;
            if (vec_ok && row < N) v[row] = c_make(dr[reg], di[reg]);
        }
    }
}
#else
    ;  // (defined in the translation unit that owns it: kh_common.h, KH_DEFINES)
#endif

// ---------------------------------------------------------------------------
// plain propagation with storage (backward sweep / iteration-0 forward sweep)
// ---------------------------------------------------------------------------
// SQ: the A^2 chain (one control, tables staged: c.sq != NULL) -- a template parameter, not a branch: with both forms in one
// kernel the table prefetch of the one had to stay alive across the rounds of the other
template <int MAXKS, int COLS, bool SQ>
__global__ void __launch_bounds__(KH_COOP_THREADS)
kh_coop_sweep_store(KhSweepArgs p, KhCoopArgs c_in, KhExchange ex, const double *__restrict__ pulses,
                    const cplx *__restrict__ state_in, cplx *__restrict__ store, cplx *__restrict__ state_out,
                    int direction) {
    extern __shared__ __attribute__((aligned(16))) char kh_coop_smem[];
    KhCoopLds &s = *(KhCoopLds *)kh_coop_smem;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int g, y;
    if (!kh_coop_place(c_in, g, y)) return;
    const int rowbase = g * 16;
    const int N = p.N, nt = p.nt, L = p.L;
    if (tid <= KH_MAX_DEGREE) s.deg[tid] = (SQ && c_in.ser_theta != nullptr) ? c_in.ser_theta[tid] : p.deg_theta[tid];
    if (tid == 0) s.abort = 0;
#ifdef KH_TIMING
    if (tid < 12) s.tim[tid] = 0.0;
#endif
    kh_coop_check_placement(c_in, ex, s, g, y, tid);
    KhCoopArgs c = c_in;
    c.local = __builtin_amdgcn_readfirstlane(s.local);
    c.ring_mask = c.local ? KH_COOP_RING_LOCAL - 1 : KH_COOP_RING - 1;
    kh_coop_resolve_tables(c);
    int r, col;
    kh_coop_owner_element<COLS>(tid, r, col);
    const int row = rowbase + r, k = y * COLS + col;
    const bool owner_valid = kh_coop_is_owner<COLS>(tid) && row < N;  // (columns beyond K carry zeros)
    const bool has_state = owner_valid && k < p.K;
    cplx state = has_state ? state_in[(size_t)k * N + row] : c_make(0.0, 0.0);
    unsigned int rid = 1;
    if (owner_valid) kh_coop_publish<COLS>(c, rid, y, row, col, state, c.local != 0);
    if (has_state && store != nullptr) store[((size_t)k * nt + (direction > 0 ? 0 : nt - 1)) * N + row] = state;
    __syncthreads();
    double rounds = 0.0;
    int m_hint = 12;
    const KhCoopFrag a = {(cplx *)s.frag + tid};
    cplx breg[MAXKS];
    double eps_prev = 0.0, eps_a = 0.0;  // (the pulse values B and A stand at: kh_coop_expm_action_sq_ahead)
    const KhCoopSqMasks mk = kh_coop_sq_masks(c, g, wave);
#pragma unroll
    for (int q = 0; q < MAXKS; ++q) breg[q] = c_make(0.0, 0.0);
    for (int step = 0; step < nt - 1; ++step) {
        const int n = direction > 0 ? step : nt - 2 - step;
        if (SQ && step % KH_COOP_REFRESH == 0) {
            kh_coop_sq_restart<MAXKS>(c, mk, g, wave, lane, a, breg, eps_prev);
            eps_a = 0.0;
        }
        double eps[KH_COOP_MAX_L];
        double theta = p.op_norms[0];
#pragma unroll
        for (int l = 0; l < KH_COOP_MAX_L; ++l) {
            eps[l] = 0.0;
            if (l < L) {
                eps[l] = pulses[(size_t)l * (nt - 1) + n];
                theta += fabs(eps[l]) * p.op_norms[1 + l];
            }
        }
        const double dt = p.dt[n];
        int nsub, m;
        kh_degree_lookup(theta * dt, s.deg, p.theta_max, p.inv_theta_max, m_hint, &nsub, &m);
        m_hint = m;
        if constexpr (SQ) {
            const bool have_next = step + 1 < nt - 1;
            const double eps_next = have_next ? pulses[direction > 0 ? n + 1 : n - 1] : 0.0;  // (one control)
            if (!kh_coop_expm_action_sq_ahead<MAXKS, COLS>(c, ex, mk, eps[0], eps_next, have_next, eps_prev, eps_a, a, breg, state,
                                                           rid, s, N, y, g, row, col, owner_valid, p.fre, p.fim, dt, nsub, m, tid,
                                                           wave, lane))
                return;
            rounds += (double)nsub * (((m + 1) >> 1) + 1);
        } else {
            kh_coop_build<MAXKS>(c.fops, eps, L, g, wave, lane, c.ks, a);
            if (!kh_coop_expm_action<MAXKS, COLS>(c, ex, a, state, rid, s, N, y, row, col, owner_valid, p.fre, p.fim, dt,
                                                  nsub, m, tid, wave, lane))
                return;
            rounds += (double)nsub * m;
        }
        if (has_state && store != nullptr) store[((size_t)k * nt + (direction > 0 ? n + 1 : n)) * N + row] = state;
    }
    if (has_state && state_out != nullptr) state_out[(size_t)k * N + row] = state;
    if (g == 0 && tid == 0 && p.stats != nullptr) {
        const int cols = min(c.cols, p.K - y * c.cols);
        atomicAdd(p.stats, rounds * cols);
#ifdef KH_TIMING
        if (y == 0) {
            p.stats[1] = s.tim[0] / rounds + 1e6 * (double)(long long)(s.tim[4] / rounds);
            p.stats[2] = s.tim[1] / rounds + 1e6 * (double)(long long)(100.0 * s.tim[5] / rounds);
            p.stats[3] = s.tim[2] / rounds + 1e6 * (double)(long long)(100.0 * s.tim[6] / rounds);
            p.stats[4] = 0.0;  // KH_TRACE: cumulative cycles per round ...
            p.stats[5] = s.tim[0] / rounds;
            p.stats[6] = p.stats[5] + s.tim[1] / rounds;
            p.stats[7] = p.stats[6] + s.tim[2] / rounds;
            p.stats[8] = p.stats[7] + s.tim[7] / rounds;
            p.stats[9] = p.stats[8] + s.tim[8] / rounds;
            p.stats[10] = p.stats[9] + s.tim[9] / rounds;
            p.stats[11] = s.tim[10] / (nt - 1);  // ... and the fragment updates per interval
            p.stats[12] = 1.0 + s.local;
            p.stats[13] = 0.0;
        }
#endif
    }
}

// ---------------------------------------------------------------------------
// forward sweep with sequential pulse update (optimize.py:444-508); single launch, in-kernel sums
// ---------------------------------------------------------------------------
// ADJ (first order, one control on the A^2 chain, u.adj_store = H_1^+ chi from kh_coop_adjoint_side): the update sums
// are taken on the adjoint side -- an element-wise product of the owners, no round -- and the dense P1 table of the
// fragment update is fetched into registers while the sums cross the workgroups (the registers are those the control
// operator's fragment needed before; the ~3 us L2-bound read sits in the shadow of the ~4.5 us exchange wait).
// SQ: the A^2 chain (one control, tables staged) -- a template parameter like kh_coop_sweep_store's, not a branch on c.sq
// P2P = false: launched on a single GPU only (no cross-GPU stage in the sums' exchange: kh_exchange_collect)
template <int MAXKS, int COLS, bool SO, bool ADJ = false, bool SQ = true, bool P2P = true>
__global__ void __launch_bounds__(KH_COOP_THREADS)
kh_coop_forward_update(KhSweepArgs p, KhCoopArgs c_in, KhUpdateArgs u, KhExchange ex) {
    static_assert(!(SO && ADJ), "the second-order bra depends on the new state");
    static_assert(SQ || !ADJ, "the adjoint-side form runs on the A^2 chain");
    extern __shared__ __attribute__((aligned(16))) char kh_coop_smem[];
    KhCoopLds &s = *(KhCoopLds *)kh_coop_smem;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int g, y;
    if (!kh_coop_place(c_in, g, y)) return;
    const int rowbase = g * 16;
    const int wg = y * c_in.G + g;  // linear workgroup index of the exchange
    const int N = p.N, nt = p.nt, L = p.L;
    if (tid <= KH_MAX_DEGREE) s.deg[tid] = (SQ && c_in.ser_theta != nullptr) ? c_in.ser_theta[tid] : p.deg_theta[tid];
    if (tid == 0) s.abort = 0;
#ifdef KH_TIMING
    if (tid < 12) s.tim[tid] = 0.0;
#endif
    kh_coop_check_placement(c_in, ex, s, g, y, tid);
    KhCoopArgs c = c_in;
    c.local = __builtin_amdgcn_readfirstlane(s.local);
    c.ring_mask = c.local ? KH_COOP_RING_LOCAL - 1 : KH_COOP_RING - 1;
    kh_coop_resolve_tables(c);
    int r, col;
    kh_coop_owner_element<COLS>(tid, r, col);
    const int row = rowbase + r, k = y * COLS + col;
    const bool owner_valid = kh_coop_is_owner<COLS>(tid) && row < N;
    const bool has_state = owner_valid && k < p.K;
    cplx state = has_state ? u.phi[(size_t)k * N + row] : c_make(0.0, 0.0);
    const double chi_norm = has_state ? u.chi_norms[k] : 0.0;
    unsigned int rid = 1;
    if (owner_valid) kh_coop_publish<COLS>(c, rid, y, row, col, state, c.local != 0);
    __syncthreads();
    int rounds = 0;  // (an SGPR counter)
    double g_a_loc[KH_COOP_MAX_L];
#pragma unroll
    for (int l = 0; l < KH_COOP_MAX_L; ++l) g_a_loc[l] = 0.0;
    int m_hint = 12;
    const KhCoopFrag a = {(cplx *)s.frag + tid};
    cplx breg[MAXKS];
    double eps_prev = 0.0;
    const KhCoopSqMasks mk = kh_coop_sq_masks(c, g, wave);
#pragma unroll
    for (int q = 0; q < MAXKS; ++q) breg[q] = c_make(0.0, 0.0);
    cplx bra_next = c_make(0.0, 0.0);
    if constexpr (ADJ) bra_next = has_state ? u.adj_store[((size_t)k * nt) * N + row] : c_make(0.0, 0.0);
    for (int n = 0; n < nt - 1; ++n) {
        const int par = n & 1;
        if (SQ && n % KH_COOP_REFRESH == 0) kh_coop_sq_restart<MAXKS>(c, mk, g, wave, lane, a, breg, eps_prev);
        // co-state (and, second order, previous-iteration state) element of this owner; ADJ: the element of
        // H_1^+ chi(t_n), fetched one interval ahead
        cplx bra;
        if constexpr (ADJ) {
            bra = bra_next;
            if (n + 1 < nt - 1 && has_state) bra_next = u.adj_store[((size_t)k * nt + n + 1) * N + row];
        } else {
            bra = has_state ? u.chi_store[((size_t)k * nt + n) * N + row] : c_make(0.0, 0.0);
        }
        if constexpr (SO) {
            if (has_state) {
                const cplx prev = u.fw_prev[((size_t)k * nt + n) * N + row];
                const double hs = 0.5 * u.sigma[n] / chi_norm;
                bra = c_make(fma(hs, state.x - prev.x, bra.x), fma(hs, state.y - prev.y, bra.y));
                u.fw_store[((size_t)k * nt + n) * N + row] = state;
            }
        }
        // ---- phi(t_n) of all objectives, then the update sums (optimize.py:454-470) ----
        cplx p1pre[ADJ ? MAXKS : 1];
        if constexpr (ADJ) {
            if (COLS == 2 || wave < 4) {  // <H_1^+ chi | phi>: the owners' own elements
                cplx ov = c_make(0.0, 0.0);
                c_fma_conj(ov, bra, state);
                const double piece = sum64(chi_norm * (u.mu_re * ov.y + u.mu_im * ov.x));
                if (lane == 0) s.red[par][wave][0] = piece;
            }
            // the dense table of the coming fragment update, on its way while the sums are exchanged
#ifndef KH_COOP_X_NOP1  // (traffic experiment with KH_COOP_X_NOREBUILD: wrong results) no table read per interval at all
            kh_coop_reg_load<MAXKS>(c.tab[3], g, wave, lane, c.ks, p1pre, mk.p1);
#endif
        }
#pragma unroll
        for (int l = 0; l < KH_COOP_MAX_L; ++l) {
            if (ADJ || l >= L) break;
            cplx w;
            if constexpr (SQ) {  // (the LDS fragment holds A for the whole sweep: the control operator from registers)
                cplx hreg[MAXKS];
                kh_coop_reg_load<MAXKS>(c.tab[1], g, wave, lane, c.ks, hreg, mk.h1);  // (sq: one control, l = 0)
                const KhCoopRegFrag<MAXKS, true> hf = {hreg, mk.h1};
                kh_coop_round<MAXKS, COLS>(c, ex, rid, y, N, hf, s, tid, wave, lane, w);
            } else {
                kh_coop_load_frag<MAXKS>(c.fops[1 + l], g, wave, lane, c.ks, a);
                kh_coop_round<MAXKS, COLS>(c, ex, rid, y, N, a, s, tid, wave, lane, w);
            }
            if (COLS == 2 || wave < 4) {  // (lanes that own no element contribute zeros: chi_norm, bra, w are 0 there)
                cplx ov = c_make(0.0, 0.0);
                c_fma_conj(ov, bra, w);
                const double piece = sum64(chi_norm * (u.mu_re * ov.y + u.mu_im * ov.x));
                if (lane == 0) s.red[par][wave][l] = piece;
            }
        }
        if constexpr (!ADJ) rounds += L;
        __syncthreads();
        if (s.abort) return;
        if (wave == 0) {
            double part[KH_COOP_MAX_L], D[KH_COOP_MAX_L];
#pragma unroll
            for (int l = 0; l < KH_COOP_MAX_L; ++l)
                part[l] = l < L ? (s.red[par][0][l] + s.red[par][1][l]) + (s.red[par][2][l] + s.red[par][3][l]) : 0.0;
            if constexpr (COLS == 2) {
#pragma unroll
                for (int l = 0; l < KH_COOP_MAX_L; ++l)
                    if (l < L) part[l] += (s.red[par][4][l] + s.red[par][5][l]) + (s.red[par][6][l] + s.red[par][7][l]);
            }
#ifdef KH_COOP_X_NOEXCH  // (timing experiment: wrong results) the workgroup's own sums only
            const bool ok = true;
            for (int l = 0; l < KH_COOP_MAX_L; ++l) D[l] = part[l];
#else
#ifdef KH_TIMING
            const long long tx0 = clock64();
#endif
            bool ok;
            if constexpr (ADJ) {  // (one control: half the gather registers next to the prefetched table)
                double D1[1];
                ok = kh_exchange<1, KH_GATHER_CHUNKS, P2P>(ex, n, wg, 1, lane, part, D1);
                D[0] = D1[0];
#pragma unroll
                for (int l = 1; l < KH_COOP_MAX_L; ++l) D[l] = 0.0;
            } else {
                ok = kh_exchange<KH_COOP_MAX_L>(ex, n, wg, L, lane, part, D);
            }
#ifdef KH_TIMING
            // how long each column group waits for the sums of ALL groups (the one point per interval where the
            // XCDs meet): cycles per interval, read back per group by scripts/timing_coop.py (stats[20 + y])
            if (g == 0 && lane == 0 && p.stats != nullptr && y < 16) p.stats[20 + y] += (double)(clock64() - tx0) / (nt - 1);
#endif
#endif
            if (lane == 0) {
#pragma unroll
                for (int l = 0; l < KH_COOP_MAX_L; ++l) s.D[par][l] = D[l];
                s.D[par][KH_COOP_MAX_L] = ok ? 1.0 : 0.0;
            }
        }
        __syncthreads();
        if (s.D[par][KH_COOP_MAX_L] == 0.0) return;
        // ---- pulse update (optimize.py:471-477) ----
        // (wave-uniform scalars are kept in SGPRs: the VGPR file is full of operator fragments)
        const double dt = kh_uniform(p.dt[n]);
        double eps[KH_COOP_MAX_L];
        double theta = kh_uniform(p.op_norms[0]);
#pragma unroll
        for (int l = 0; l < KH_COOP_MAX_L; ++l) {
            if (l >= L) break;
            const double d1 = kh_uniform(s.D[par][l]);
            const double stepw = kh_uniform(u.shape[(size_t)l * (nt - 1) + n] / u.lambda[l]);
            eps[l] = kh_uniform(u.guess[(size_t)l * (nt - 1) + n] + stepw * d1);
            g_a_loc[l] = kh_uniform(g_a_loc[l] + stepw * (d1 * d1) * dt);
            theta = kh_uniform(theta + fabs(eps[l]) * p.op_norms[1 + l]);
            if (wg == 0 && tid == 0) u.opt[(size_t)l * (nt - 1) + n] = eps[l];
        }
        // ---- propagate over interval n with the updated pulses (optimize.py:479-491) ----
        int nsub, m;
        kh_degree_lookup(theta * dt, s.deg, p.theta_max, p.inv_theta_max, m_hint, &nsub, &m);
        m_hint = m;
        if constexpr (ADJ) {  // (one control on the A^2 chain: the only form this instantiation contains)
            if (!kh_coop_expm_action_sq<MAXKS, COLS>(c, ex, mk, eps[0], eps_prev, a, breg, state, rid, s, N, y, g, row, col,
                                                     owner_valid, p.fre, p.fim, dt, nsub, m, tid, wave, lane, &p1pre))
                return;
            rounds += nsub * (((m + 1) >> 1) + 1);
        } else if constexpr (SQ) {
            if (!kh_coop_expm_action_sq<MAXKS, COLS>(c, ex, mk, eps[0], eps_prev, a, breg, state, rid, s, N, y, g, row, col,
                                                     owner_valid, p.fre, p.fim, dt, nsub, m, tid, wave, lane))
                return;
            rounds += nsub * (((m + 1) >> 1) + 1);
        } else {
            kh_coop_build<MAXKS>(c.fops, eps, L, g, wave, lane, c.ks, a);
            if (!kh_coop_expm_action<MAXKS, COLS>(c, ex, a, state, rid, s, N, y, row, col, owner_valid, p.fre, p.fim, dt,
                                                  nsub, m, tid, wave, lane))
                return;
            rounds += nsub * m;
        }
    }
    if (has_state) {
        u.phi[(size_t)k * N + row] = state;
        if constexpr (SO) u.fw_store[((size_t)k * nt + (nt - 1)) * N + row] = state;
    }
    if (wg == 0 && tid == 0) {
#pragma unroll
        for (int l = 0; l < KH_COOP_MAX_L; ++l)
            if (l < L) u.g_a[l] = g_a_loc[l];
    }
#ifdef KH_TIMING
    if (blockIdx.x == 0 && tid == 0 && p.stats != nullptr) {  // (raw slots of KH_TRACE: fragment updates per interval, rounds)
        p.stats[28] = s.tim[10] / (nt - 1);
        p.stats[29] = (s.tim[0] + s.tim[1] + s.tim[2] + s.tim[7] + s.tim[8]) / (nt - 1);
        p.stats[30] = s.tim[9] / (nt - 1);
    }
#endif
    if (g == 0 && tid == 0 && p.stats != nullptr) {
        const int cols = min(c.cols, p.K - y * c.cols);
        atomicAdd(p.stats, (double)rounds * cols);
    }
}
