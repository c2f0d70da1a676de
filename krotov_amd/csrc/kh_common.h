// Device-side building blocks shared by the Krotov sweep kernels (gfx950).
//
//   * complex128 helpers on double2
//   * DPP lane reductions (64-wide wavefronts; 16-lane rows)
//   * Taylor degree / sub-step selection for the exponential action
//   * the cross-workgroup exchange used once per time interval by the
//     forward-update sweep (epoch-tagged 8-byte granules, relaxed agent-scope
//     atomics, bounded spin)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double2 cplx;

#define KH_MAX_L 8          // controls per problem the kernels are compiled for
#define KH_MAX_DEGREE 64    // hard cap on the Taylor degree per sub-step

// ---------------------------------------------------------------------------
// complex arithmetic
// ---------------------------------------------------------------------------

__device__ __forceinline__ cplx c_make(double re, double im) { return make_double2(re, im); }

__device__ __forceinline__ cplx c_mul(cplx a, cplx b) {
    return make_double2(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x));
}

// acc += a * b  (4 real FMAs)
__device__ __forceinline__ void c_fma(cplx &acc, cplx a, cplx b) {
    acc.x = fma(a.x, b.x, acc.x);
    acc.x = fma(-a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y);
    acc.y = fma(a.y, b.x, acc.y);
}

// acc += conj(a) * b
__device__ __forceinline__ void c_fma_conj(cplx &acc, cplx a, cplx b) {
    acc.x = fma(a.x, b.x, acc.x);
    acc.x = fma(a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y);
    acc.y = fma(-a.y, b.x, acc.y);
}

// ---------------------------------------------------------------------------
// DPP cross-lane moves on doubles (two 32-bit DPP moves each)
// ---------------------------------------------------------------------------

template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

#define KH_DPP_XOR1 0xB1         // quad_perm [1,0,3,2]
#define KH_DPP_XOR2 0x4E         // quad_perm [2,3,0,1]
#define KH_DPP_HALF_MIRROR 0x141 // lane i <-> 7-i within each 8 lanes
#define KH_DPP_MIRROR 0x140      // lane i <-> 15-i within each 16-lane row

// all-reduce (sum) over groups of 4 / 8 / 16 adjacent lanes; every lane of the
// group ends up with the same value, summed in the same order.
__device__ __forceinline__ double sum4(double v) {
    v += dpp_move<KH_DPP_XOR1>(v);
    v += dpp_move<KH_DPP_XOR2>(v);
    return v;
}
__device__ __forceinline__ double sum8(double v) {
    v = sum4(v);
    v += dpp_move<KH_DPP_HALF_MIRROR>(v);
    return v;
}
__device__ __forceinline__ double sum16(double v) {
    v = sum8(v);
    v += dpp_move<KH_DPP_MIRROR>(v);
    return v;
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// full 64-lane sum, same value (and same summation order) in every lane
__device__ __forceinline__ double sum64(double v) {
    v = sum16(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// ---------------------------------------------------------------------------
// Taylor degree selection
// ---------------------------------------------------------------------------
// exp(B) v with theta >= ||B||: split into s sub-steps of norm th = theta/s
// <= theta_max and truncate each at the smallest degree m whose remainder
// bound th^(m+1)/(m+1)! / (1 - th/(m+2)) is <= tol.  theta=0.5, tol=2^-53
// gives m=14; theta=1.0 gives m=18.
__host__ __device__ inline void kh_choose_degree(double theta, double tol, double theta_max, int *s_out,
                                                 int *m_out) {
    int s = 1;
    if (theta > theta_max) s = (int)ceil(theta / theta_max);
    const double th = theta / s;
    double term = 1.0;
    int m = 1;
    for (; m < KH_MAX_DEGREE; ++m) {
        term *= th / m;  // th^m / m!
        const double next = term * th / (m + 1);
        if (next <= tol * (1.0 - th / (m + 2))) break;
    }
    *s_out = s;
    *m_out = m;
}

// ---------------------------------------------------------------------------
// cross-workgroup exchange of per-workgroup partial sums
// ---------------------------------------------------------------------------
// One slot per (parity, workgroup, control): two 8-byte granules
// {epoch:32 | hi32(value)} {epoch:32 | lo32(value)}.  A granule is written by
// ONE aligned 8-byte agent-scope store, so it is never torn; the reader
// accepts a value only when both tags equal the epoch it waits for.  Slots are
// double-buffered on the parity of the interval: a workgroup can be at most
// one interval ahead of the slowest reader (it needs everybody's value of
// interval n before it can publish n+1), so parity n+2 is free when written.
// The slot array is zeroed by a memset node before every launch; epoch =
// interval + 1 is never 0.

typedef unsigned long long kh_u64;

struct KhExchange {
    kh_u64 *slots;          // [2][G][L][2]
    unsigned int *abort_flag;  // set by any workgroup that gave up
    int G;                  // workgroups taking part
    long long timeout_ticks;   // wall_clock64 ticks (100 MHz)
};

__device__ __forceinline__ void kh_publish(const KhExchange &ex, int parity, int wg, int L, int l,
                                           double value, unsigned int epoch) {
    const kh_u64 bits = (kh_u64)__double_as_longlong(value);
    kh_u64 *g = ex.slots + (((size_t)parity * ex.G + wg) * L + l) * 2;
    __hip_atomic_store(g, ((kh_u64)epoch << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(g + 1, ((kh_u64)epoch << 32) | (bits & 0xffffffffull), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// Called by ONE full wave.  Returns false on timeout/abort.  On success every
// lane holds in out[l] the sum over all workgroups, accumulated in the fixed
// order (lane-strided partial sums in workgroup order, then the sum64 tree),
// identical in every workgroup.
template <int MAXL>
__device__ __forceinline__ bool kh_gather(const KhExchange &ex, int parity, int L, unsigned int epoch, int lane,
                                          double (&out)[MAXL]) {
    double acc[MAXL];
#pragma unroll
    for (int l = 0; l < MAXL; ++l) acc[l] = 0.0;
    const long long t0 = wall_clock64();
    for (int wg = lane; wg - lane < ex.G; wg += 64) {  // uniform trip count
        const bool active = wg < ex.G;
#pragma unroll
        for (int l = 0; l < MAXL; ++l) {
            if (l >= L) break;
            const kh_u64 *g = ex.slots + (((size_t)parity * ex.G + (active ? wg : 0)) * L + l) * 2;
            kh_u64 a = 0, b = 0;
            unsigned int spins = 0;
            for (;;) {
                a = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = !active || (((unsigned int)(a >> 32) == epoch) && ((unsigned int)(b >> 32) == epoch));
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 1023u) == 0) {  // wave-uniform
                    const bool gave_up =
                        (wall_clock64() - t0 > ex.timeout_ticks) ||
                        (__hip_atomic_load(ex.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u);
                    if (__any(gave_up)) {
                        if (lane == 0)
                            __hip_atomic_store(ex.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        return false;
                    }
                }
            }
            if (active) {
                const kh_u64 bits = ((a & 0xffffffffull) << 32) | (b & 0xffffffffull);
                acc[l] += __longlong_as_double((long long)bits);
            }
        }
    }
#pragma unroll
    for (int l = 0; l < MAXL; ++l) out[l] = (l < L) ? sum64(acc[l]) : 0.0;
    return true;
}
