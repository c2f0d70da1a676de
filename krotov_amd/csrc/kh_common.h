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

// Translation units.  `krotov_hip.hip` compiled by itself (no -DKH_TU) is the whole library in one unit -- what the
// timing / stress builds of scripts/ and a plain `hipcc ... krotov_hip.hip` do.  krotov_amd/build.py compiles the same
// sources as several units in parallel: KH_TU_MAIN (host code, dispatch, the small set-up kernels) sees the sweep-kernel
// templates but instantiates none of those listed in kh_instances.inc (`extern template`); every other unit
// (csrc/kh_tu.hip with -DKH_TU=<family>) holds the explicit instantiations of its family and the family's
// non-template kernels, which everywhere else are declarations only (KH_DEFINES).  A kernel launch is a host-side
// reference to the kernel's handle, so no relocatable device code is needed: every unit registers its own code object.
#define KH_TU_ALL 0
#define KH_TU_MAIN 1
#define KH_TU_GENERIC 2
#define KH_TU_MINI 3
#define KH_TU_TILE 4
#define KH_TU_Q2 5
#define KH_TU_STREAM 6
#define KH_TU_ENS 7
#define KH_TU_TILEN 8
#define KH_TU_COOP_STORE 9
#define KH_TU_COOP_UPDATE_A 10
#define KH_TU_COOP_UPDATE_B 11
#define KH_TU_ELL_STORE 12
#define KH_TU_ELL_UPDATE_A 13
#define KH_TU_ELL_UPDATE_B 14
#define KH_TU_TILEX 15
#ifndef KH_TU
#define KH_TU KH_TU_ALL
#endif
#define KH_DEFINES(owner) (KH_TU == KH_TU_ALL || KH_TU == (owner))

#define KH_MAX_L 8          // controls per problem the register-resident kernel families are compiled for
#define KH_GEN_MAX_L 32     // ... and the generic kernels (kh_generic.h), whose per-control values live in LDS
#define KH_MAX_DEGREE 64    // hard cap on the Taylor degree per sub-step

// 1/j for the Taylor coefficients: a scalar load instead of two fp64 divisions
// (v_rcp_f64 + Newton steps, ~60 cycles each) on the critical path of every term
static __constant__ double kh_inv_table[KH_MAX_DEGREE + 1] = {0.0, 1.0 / 1, 1.0 / 2, 1.0 / 3, 1.0 / 4, 1.0 / 5, 1.0 / 6, 1.0 / 7, 1.0 / 8, 1.0 / 9, 1.0 / 10, 1.0 / 11, 1.0 / 12, 1.0 / 13, 1.0 / 14, 1.0 / 15, 1.0 / 16, 1.0 / 17, 1.0 / 18, 1.0 / 19, 1.0 / 20, 1.0 / 21, 1.0 / 22, 1.0 / 23, 1.0 / 24, 1.0 / 25, 1.0 / 26, 1.0 / 27, 1.0 / 28, 1.0 / 29, 1.0 / 30, 1.0 / 31, 1.0 / 32, 1.0 / 33, 1.0 / 34, 1.0 / 35, 1.0 / 36, 1.0 / 37, 1.0 / 38, 1.0 / 39, 1.0 / 40, 1.0 / 41, 1.0 / 42, 1.0 / 43, 1.0 / 44, 1.0 / 45, 1.0 / 46, 1.0 / 47, 1.0 / 48, 1.0 / 49, 1.0 / 50, 1.0 / 51, 1.0 / 52, 1.0 / 53, 1.0 / 54, 1.0 / 55, 1.0 / 56, 1.0 / 57, 1.0 / 58, 1.0 / 59, 1.0 / 60, 1.0 / 61, 1.0 / 62, 1.0 / 63, 1.0 / 64};

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

// every pattern used below reads a valid lane for every lane, so the "old"
// operand is irrelevant: mov_dpp (undef old, bound_ctrl) compiles to a single
// v_mov_b32_dpp per dword instead of copy + dpp.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

#define KH_DPP_XOR1 0xB1         // quad_perm [1,0,3,2]
#define KH_DPP_XOR2 0x4E         // quad_perm [2,3,0,1]
#define KH_DPP_HALF_MIRROR 0x141 // lane i <-> 7-i within each 8 lanes
#define KH_DPP_MIRROR 0x140      // lane i <-> 15-i within each 16-lane row
#define KH_DPP_ROR8 0x128        // lane i <- lane (i + 8) % 16 within each 16-lane row
#define KH_DPP_ROR4 0x124        // lane i <- lane (i + 4) % 16 within each 16-lane row

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

// a wave-uniform double moved to scalar registers (frees its VGPR pair; VALU ops take it as an
// SGPR operand)
__device__ __forceinline__ double kh_uniform(double v) {
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// full 64-lane sum, same value (and same summation order) in every lane
__device__ __forceinline__ double sum64(double v) {
    v = sum16(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// The same sum on the matrix core: v_mfma_f64_4x4x4 with a constant B operand adds its A operand over the four lanes
// 16 k + i; two row rotations add the four 4-lane blocks of a 16-lane row (every lane of row r then holds the sum over
// the lanes 16 k + 4 blk + r); a second MFMA adds the four rows.  2 MFMA + 6 VALU instead of the butterfly's 12 DPP
// moves, 6 adds and 8 read-lanes; every lane ends up with the total (fixed order: deterministic).
__device__ __forceinline__ double sum64_mfma(double v) {
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
    d += dpp_move<KH_DPP_ROR8>(d);
    d += dpp_move<KH_DPP_ROR4>(d);
    return __builtin_amdgcn_mfma_f64_4x4x4f64(d, 1.0, 0.0, 0, 0, 0);
}

// ---------------------------------------------------------------------------
// Taylor degree selection
// ---------------------------------------------------------------------------
// exp(B) v with theta >= ||B||: split into s sub-steps of norm th = theta/s
// <= theta_max and truncate each at the smallest degree m whose remainder
// bound th^(m+1)/(m+1)! / (1 - th/(m+2)) is <= tol.  theta=0.5, tol=2^-53
// gives m=14; theta=1.0 gives m=18.
//
// Evaluated without divisions on the device: tab[m] (host-built, see
// kh_build_degree_table) is the largest theta for which degree m meets tol, so
// the degree is the smallest m with theta <= tab[m].  `hint` (the previous
// interval's degree) makes the search O(1) along a smooth pulse.
__host__ inline void kh_build_degree_table(double tol, double *tab /*[KH_MAX_DEGREE+1]*/) {
    tab[0] = 0.0;
    for (int m = 1; m <= KH_MAX_DEGREE; ++m) {
        // bound(th) = th^(m+1)/(m+1)! / (1 - th/(m+2)), increasing on [0, m+2)
        double lo = 0.0, hi = (double)(m + 2) * (1.0 - 1e-12);
        for (int it = 0; it < 200; ++it) {
            const double th = 0.5 * (lo + hi);
            double term = 1.0;
            for (int j = 1; j <= m + 1; ++j) term *= th / j;
            if (term <= tol * (1.0 - th / (m + 2))) lo = th; else hi = th;
        }
        tab[m] = lo;
    }
}

// ---------------------------------------------------------------------------
// Series coefficients for the two-terms-per-phase kernels (kh_tile64q2.h)
// ---------------------------------------------------------------------------
// Those kernels evaluate  sum_j c_j (f h A)^j v  with real c_j through the chain of even terms
//   T_0 = c_0 v,  T_{2p+2} = r2_p f^2 h^2 A^2 T_{2p}   and   sum_odd = f A sum_p r1_p h T_{2p},
// so all they need per degree m is c_0 and the rows {r1_p, r2_p}: r1_0 = c_1, r2_0 = c_2 (taken relative to v,
// not to T_0), r1_p = c_{2p+1}/c_{2p}, r2_p = c_{2p+2}/c_{2p} for p >= 1.
//   * Taylor (any generator): c_j = 1/j!, thresholds from kh_build_degree_table.
//   * Generators with a REAL spectrum (every operator Hermitian bit for bit, f = -+i): the truncated Chebyshev
//     series of exp(-+i theta x) on [-1, 1], rewritten in powers of (-+i theta x).  Its error 2 sum_{k>m}
//     |J_k(theta)| ~ 2 (theta/2)^(m+1)/(m+1)! is 2^m times smaller than Taylor's remainder: theta = 0.5 needs
//     degree 12 instead of 14 (6 phases instead of 7), theta = 1 needs 14 instead of 18.  The coefficients of
//     degree m are computed for theta_b = tab[m], the largest theta the degree serves, and are valid for
//     every spectrum inside [-theta_b, theta_b].  Only even degrees, theta_b <= 2 (beyond that the power
//     basis loses digits; the kernels sub-step at theta_max <= 1 anyway).
#define KH_Q2_ROWS (KH_MAX_DEGREE / 2)

// The one-term-per-phase kernels (kh_tile64.h) use the same series term by term: T_j = ratio_j (f h A) T_{j-1},
// ratios[m][0] = c_0, ratios[m][1] = c_1 (relative to v), ratios[m][j] = c_j / c_{j-1}  (Taylor: 1, 1, 1/j).
#define KH_RATIO_STRIDE (KH_MAX_DEGREE + 1)

__host__ inline void kh_build_taylor_rows(double *c0 /*[KH_MAX_DEGREE+1]*/, double *rows /*[KH_MAX_DEGREE+1][KH_Q2_ROWS][2]*/,
                                          double *ratios /*[KH_MAX_DEGREE+1][KH_RATIO_STRIDE]*/) {
    for (int m = 0; m <= KH_MAX_DEGREE; ++m) {
        c0[m] = 1.0;
        ratios[(size_t)m * KH_RATIO_STRIDE] = 1.0;
        for (int j = 1; j <= KH_MAX_DEGREE; ++j) ratios[(size_t)m * KH_RATIO_STRIDE + j] = 1.0 / j;
        for (int p = 0; p < KH_Q2_ROWS; ++p) {
            rows[((size_t)m * KH_Q2_ROWS + p) * 2 + 0] = 1.0 / (2 * p + 1);
            rows[((size_t)m * KH_Q2_ROWS + p) * 2 + 1] = 1.0 / ((2.0 * p + 1) * (2 * p + 2));
        }
    }
}

// J_k(theta), k = 0..kmax, by Miller's downward recurrence J_{k-1} = (2k/theta) J_k - J_{k+1}, normalised
// with J_0 + 2 sum_k J_{2k} = 1 (theta > 0; stable in this direction)
__host__ inline void kh_bessel_j(double theta, int kmax, long double *J) {
    const int start = kmax + 40 + (int)(2.0 * theta);
    long double jp = 0.0L, jc = 1e-300L, norm = 0.0L;
    for (int k = start; k >= 1; --k) {
        const long double jm = (2.0L * k / (long double)theta) * jc - jp;  // J_{k-1}
        jp = jc;
        jc = jm;
        if (k - 1 <= kmax) J[k - 1] = jc;
        if (((k - 1) & 1) == 0) norm += (k - 1 == 0 ? 1.0L : 2.0L) * jc;
        if (fabsl(jc) > 1e200L) {  // rescale (everything stored so far too)
            jc *= 1e-200L;
            jp *= 1e-200L;
            norm *= 1e-200L;
            for (int q = k - 1; q <= kmax; ++q) J[q] *= 1e-200L;
        }
    }
    for (int k = 0; k <= kmax; ++k) J[k] /= norm;
}

// tab[m]: largest theta <= 2 the degree-m Chebyshev truncation serves at `tol` (even m; odd m repeat m-1 so
// that the smallest-degree search never lands on them); c0, rows as above
//
// theta_cap: largest theta served by the Chebyshev form (2 for the register-tile kernels, which sub-step at theta <= 1
// anyway; 4 for the cooperative kernels, where a term costs a cross-workgroup round -- measured on the 400-dim
// Liouvillian of config 4: the power form still reaches 2e-15 at theta = 4, like Taylor's, with degree 22 instead of 30).
// delta > 0: the generator f A h is only NEARLY anti-Hermitian -- its Hermitian part is bounded by delta (a weakly
// damped Liouvillian: 3e-4 against theta = 2.6 in config 4).  Its numerical range then lies in the strip
// |Re z| <= delta intersected with the disk |z| <= theta; in the variable w = z / (i theta) of the Chebyshev series
// that is {|Im w| <= d, |w| <= 1}, d = delta / theta, whose points farthest from the segment [-1, 1] -- in the
// sense of the Bernstein ellipses with foci +-1 -- are the corners w* = sqrt(1 - d^2) +- i d (on the arc the sum of
// the distances to the foci, 2 sin(phi/2) + 2 cos(phi/2), grows with phi; on the flat edges it is convex and even).
// On the ellipse through w*,  |T_k| <= cosh(k eta)  with  eta = log |w* + sqrt(w*^2 - 1)| ~ sqrt(d)  (NOT
// asinh(d) ~ d, the semi-minor axis alone: an earlier version used that and was ~5x over the tolerance at
// delta = 0.05), and with Crouzeix's constant 1 + sqrt 2 for non-normal matrices the error is at most
// (1 + sqrt 2) 2 sum_{k > m} |J_k(theta)| cosh(k eta).
__host__ inline long double kh_bernstein_eta(long double d) {
    // w = a + i d, a = sqrt(1 - d^2);  w^2 - 1 = -2 d^2 + 2 i a d;  principal square root by hand (long double)
    const long double a = sqrtl(fmaxl(0.0L, 1.0L - d * d));
    const long double xr = -2.0L * d * d, xi = 2.0L * a * d;
    const long double r = sqrtl(xr * xr + xi * xi);
    const long double sr = sqrtl(fmaxl(0.0L, 0.5L * (r + xr))), si = sqrtl(fmaxl(0.0L, 0.5L * (r - xr)));  // (xi >= 0)
    const long double p = hypotl(a + sr, d + si), q = hypotl(a - sr, d - si);
    return logl(p > q ? p : q);
}
__host__ inline void kh_build_real_spectrum_rows(double tol, double *tab /*[KH_MAX_DEGREE+1]*/, double *c0, double *rows,
                                                 double *ratios, double theta_cap = 2.0, double delta = 0.0) {
    const int TAIL = 40;
    long double J[KH_MAX_DEGREE + TAIL + 2];
    auto err = [&](double theta, int m) {
        kh_bessel_j(theta, m + TAIL, J);
        long double e = 0.0L;
        if (delta > 0.0) {
            const long double d = (long double)delta / (long double)theta;
            const long double eta = d < 1.0L ? kh_bernstein_eta(d) : 1e3L;  // (d >= 1: no such form; the bound fails)
            for (int k = m + 1; k <= m + TAIL; ++k) e += fabsl(J[k]) * coshl(k * eta);
            return (double)(2.0L * (1.0L + sqrtl(2.0L)) * e);
        }
        for (int k = m + 1; k <= m + TAIL; ++k) e += fabsl(J[k]);
        return (double)(2.0L * e);
    };
    kh_build_taylor_rows(c0, rows, ratios);  // (rows of the degrees that stay with Taylor)
    double taylor_tab[KH_MAX_DEGREE + 1];
    kh_build_degree_table(tol, taylor_tab);
    tab[0] = 0.0;
    for (int m = 1; m <= KH_MAX_DEGREE; ++m) {
        if (m & 1) {
            tab[m] = tab[m - 1];
            continue;
        }
        if (taylor_tab[m] >= theta_cap) {  // beyond the cap of the Chebyshev form: plain Taylor (rows already there)
            tab[m] = taylor_tab[m] > tab[m - 1] ? taylor_tab[m] : tab[m - 1];
            continue;
        }
        double lo = 0.0, hi = theta_cap;
        if (err(hi, m) <= tol) {
            lo = hi;
        } else {
            for (int it = 0; it < 60; ++it) {
                const double th = 0.5 * (lo + hi);
                if (err(th, m) <= tol) lo = th; else hi = th;
            }
        }
        if (lo < tab[m - 1]) lo = tab[m - 1];
        tab[m] = lo;
        const double theta = lo;
        if (!(theta > 0.0)) continue;
        // power coefficients of T_k by the recurrence T_{k+1} = 2 x T_k - T_{k-1}
        static long double T[KH_MAX_DEGREE + 1][KH_MAX_DEGREE + 1];
        for (int k = 0; k <= m; ++k)
            for (int j = 0; j <= m; ++j) T[k][j] = 0.0L;
        T[0][0] = 1.0L;
        if (m >= 1) T[1][1] = 1.0L;
        for (int k = 1; k < m; ++k)
            for (int j = 0; j <= k + 1; ++j) T[k + 1][j] = (j > 0 ? 2.0L * T[k][j - 1] : 0.0L) - T[k - 1][j];
        kh_bessel_j(theta, m, J);
        long double c[KH_MAX_DEGREE + 1];
        for (int j = 0; j <= m; ++j) {
            long double acc = 0.0L;
            for (int k = j; k <= m; k += 2)  // (+-i)^k x^j = (+-i)^j (-1)^((k-j)/2) x^j
                acc += (k == 0 ? 1.0L : 2.0L) * J[k] * T[k][j] * (((k - j) / 2) % 2 ? -1.0L : 1.0L);
            c[j] = acc / powl((long double)theta, j);
        }
        c0[m] = (double)c[0];
        ratios[(size_t)m * KH_RATIO_STRIDE] = (double)c[0];
        for (int j = 1; j <= m; ++j) ratios[(size_t)m * KH_RATIO_STRIDE + j] = (double)(j == 1 ? c[1] : c[j] / c[j - 1]);
        for (int p = 0; 2 * p + 2 <= m && p < KH_Q2_ROWS; ++p) {
            const long double den = p == 0 ? 1.0L : c[2 * p];
            rows[((size_t)m * KH_Q2_ROWS + p) * 2 + 0] = (double)(c[2 * p + 1] / den);
            rows[((size_t)m * KH_Q2_ROWS + p) * 2 + 1] = (double)(c[2 * p + 2] / den);
        }
    }
}

// Per-thread cache of the bracket [tab[m-1], tab[m]] of the current degree: along
// a smooth pulse the degree rarely changes, and the table reads (LDS, ~100
// cycles each, dependent) would otherwise sit on every interval's critical path.
struct KhDegreeCache {
    int m;
    double lo, hi;
};

__device__ __forceinline__ void kh_degree_cached(double theta, const double *tab, double theta_max,
                                                 double inv_theta_max, KhDegreeCache &c, int *s_out, int *m_out) {
    int s = 1;
    double th = theta;
    if (theta > theta_max) {
        s = (int)ceil(theta * inv_theta_max);
        th = theta / s;
    }
    if (!(th > c.lo && th <= c.hi)) {
        int m = c.m < 1 ? 1 : c.m;
        while (m > 1 && th <= tab[m - 1]) --m;
        while (m < KH_MAX_DEGREE && th > tab[m]) ++m;
        c.m = m;
        c.lo = m > 1 ? tab[m - 1] : -1.0;
        c.hi = m < KH_MAX_DEGREE ? tab[m] : 1e300;
    }
    *s_out = s;
    *m_out = c.m;
}

__device__ __forceinline__ void kh_degree_lookup(double theta, const double *__restrict__ tab, double theta_max,
                                                 double inv_theta_max, int hint, int *s_out, int *m_out) {
    int s = 1;
    double th = theta;
    if (theta > theta_max) {  // rare: several Taylor sub-steps
        s = (int)ceil(theta * inv_theta_max);
        th = theta / s;
    }
    int m = hint < 1 ? 1 : (hint > KH_MAX_DEGREE ? KH_MAX_DEGREE : hint);
    while (m > 1 && th <= tab[m - 1]) --m;
    while (m < KH_MAX_DEGREE && th > tab[m]) ++m;
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
    int first_poll_delay;   // s_sleep units (64 cycles) between publishing and the first poll
    long long timeout_ticks;   // wall_clock64 ticks (100 MHz)
    // ---- second stage across GPUs (objectives sharded over ranks; world == 1: unused) ----
    // Every rank owns a window [2][world][L][2] of 8-byte granules in fine-grained
    // device memory; peer_windows[r] is rank r's window mapped into this process
    // (xGMI peer access; peer_windows[rank] is the local one).  After the in-GPU
    // stage the leader workgroup stores its GPU's sum into slot [parity][rank] of
    // EVERY window (system-scope atomics), and every workgroup polls its own GPU's
    // window and adds the `world` values in rank order -- all GPUs obtain the
    // bit-identical total.  Epochs are monotonic across sweeps (epoch_base), so
    // the windows never need clearing.
    kh_u64 *const *peer_windows;  // device array [world]
    kh_u64 *my_window;
    int world, rank;
    unsigned int epoch_base;
    int fail_at;  // test hook (KH_P2P_FAIL_AT): this rank withholds its GPU's sum at that interval; < 0: never
    // diagnostics of a sharded sweep (kh_p2p_stats): workgroup 0 adds the 100 MHz ticks it spent in the in-GPU gather to
    // [0] and those between publishing its GPU's sum and holding all ranks' sums to [1], per interval; NULL: off
    unsigned long long *wait_ticks;
};

// Called by (at least) the first 2*L lanes of one wave: one 8-byte store each.
__device__ __forceinline__ void kh_publish(const KhExchange &ex, int parity, int wg, int L, int lane,
                                           const double *values, unsigned int epoch) {
#ifdef KH_EXCH_STRESS  // (protocol test build: publications delayed pseudo-randomly per workgroup and interval -- results must not change)
    {
        const unsigned int hsh = ((unsigned int)wg * 2654435761u) ^ (epoch * 40503u);
        const int naps = (int)((hsh >> 9) % 19u);
        for (int d = 0; d < naps; ++d) __builtin_amdgcn_s_sleep(16);
    }
#endif
    if (lane < 2 * L) {
        const int l = lane >> 1;
        const kh_u64 bits = (kh_u64)__double_as_longlong(values[l]);
        const kh_u64 half = (lane & 1) ? (bits & 0xffffffffull) : (bits >> 32);
        kh_u64 *g = ex.slots + (((size_t)parity * ex.G + wg) * L + l) * 2 + (lane & 1);
        __hip_atomic_store(g, ((kh_u64)epoch << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#define KH_GATHER_CHUNKS 4  // workgroups per lane (default): the exchange handles up to 256 workgroups
#define KH_GATHER_CHUNKS_WIDE 8  // two 256-thread workgroups per CU: up to 512

// Called by ONE full wave.  Returns false on timeout/abort.  All granule loads
// of a polling round are issued back to back (one memory round trip per round,
// not one per producer).  On success every lane holds in out[l] the sum over
// all workgroups, accumulated in a fixed order (per lane: workgroups lane,
// lane+64, ...; then the sum64 tree) that is identical in every workgroup, so
// every workgroup derives bit-identical pulse values.
// kh_gather_range: the controls l0 ... l0 + MAXL - 1 of L (the generic kernels gather more than KH_MAX_L controls in
// groups of KH_MAX_L: a round's granules must fit the registers).
template <int MAXL, int CH>
__device__ __forceinline__ bool kh_gather_range(const KhExchange &ex, int parity, int L, int l0, unsigned int epoch, int lane,
                                                double (&out)[MAXL]);
template <int MAXL, int CH = KH_GATHER_CHUNKS>
__device__ __forceinline__ bool kh_gather(const KhExchange &ex, int parity, int L, unsigned int epoch, int lane,
                                          double (&out)[MAXL]) {
    return kh_gather_range<MAXL, CH>(ex, parity, L, 0, epoch, lane, out);
}
template <int MAXL, int CH>
__device__ __forceinline__ bool kh_gather_range(const KhExchange &ex, int parity, int L, int l0, unsigned int epoch, int lane,
                                                double (&out)[MAXL]) {
    const kh_u64 *base = ex.slots + ((size_t)parity * ex.G * L + l0) * 2;
    L -= l0;  // (controls left from l0 on; the slot stride below keeps the full count)
    const int Lfull = L + l0;
    kh_u64 a[MAXL][CH], b[MAXL][CH];
    long long t0 = 0;  // (taken when the first poll fails: s_memrealtime is an SMEM read that the next lgkmcnt wait -- the
                       // LDS write of the result -- would sit behind in every interval)
    unsigned int spins = 0;
    // a poll issued before the slowest producer's store has reached the memory
    // side costs a whole extra round trip: give the stores a head start
    for (int d = 0; d < ex.first_poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int l = 0; l < MAXL; ++l) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int wg = lane + 64 * c;
                if (l < L && wg < ex.G) {
                    const kh_u64 *g = base + ((size_t)wg * Lfull + l) * 2;
                    a[l][c] = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    b[l][c] = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    a[l][c] = b[l][c] = (kh_u64)epoch << 32;  // neutral: tag ok, value +0.0
                }
            }
        }
#pragma unroll
        for (int l = 0; l < MAXL; ++l)
#pragma unroll
            for (int c = 0; c < CH; ++c)
                ok = ok && ((unsigned int)(a[l][c] >> 32) == epoch) && ((unsigned int)(b[l][c] >> 32) == epoch);
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(1);
        if (spins == 0) t0 = wall_clock64();
        if ((++spins & 255u) == 0) {  // wave-uniform
            const bool gave_up =
                (wall_clock64() - t0 > ex.timeout_ticks) ||
                (__hip_atomic_load(ex.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u);
            if (__any(gave_up)) {
                if (lane == 0) __hip_atomic_store(ex.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
#ifdef KH_TIMING
    if (lane == 0 && blockIdx.x == 0 && blockIdx.y == 0) ex.abort_flag[1] += spins + 1u;  // polling rounds (workgroup 0)
#endif
#pragma unroll
    for (int l = 0; l < MAXL; ++l) {
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const kh_u64 bits = ((a[l][c] & 0xffffffffull) << 32) | (b[l][c] & 0xffffffffull);
            acc += __longlong_as_double((long long)bits);
        }
        out[l] = (l < L) ? sum64(acc) : 0.0;
    }
    return true;
}

// The gather of ONE control out of L (slot layout [wg][L][2]): several controls are gathered by several waves side by
// side (kh_tile64.h) -- a wave that polls 8 L granules per lane for all controls at once pays for every failed round
// with L times the loads.  Same summation order as kh_gather.
template <int CH = KH_GATHER_CHUNKS>
__device__ __forceinline__ bool kh_gather_one(const KhExchange &ex, int parity, int L, int l, unsigned int epoch, int lane,
                                              double &out) {
    const kh_u64 *base = ex.slots + (size_t)parity * ex.G * L * 2;
    kh_u64 a[CH], b[CH];
    long long t0 = 0;
    unsigned int spins = 0;
    for (int d = 0; d < ex.first_poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int wg = lane + 64 * c;
            if (wg < ex.G) {
                const kh_u64 *g = base + ((size_t)wg * L + l) * 2;
                a[c] = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b[c] = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                a[c] = b[c] = (kh_u64)epoch << 32;  // neutral: tag ok, value +0.0
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
            ok = ok && ((unsigned int)(a[c] >> 32) == epoch) && ((unsigned int)(b[c] >> 32) == epoch);
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(1);
        if (spins == 0) t0 = wall_clock64();
        if ((++spins & 255u) == 0) {  // wave-uniform
            const bool gave_up =
                (wall_clock64() - t0 > ex.timeout_ticks) ||
                (__hip_atomic_load(ex.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u);
            if (__any(gave_up)) {
                if (lane == 0) __hip_atomic_store(ex.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const kh_u64 bits = ((a[c] & 0xffffffffull) << 32) | (b[c] & 0xffffffffull);
        acc += __longlong_as_double((long long)bits);
    }
    out = sum64(acc);
    return true;
}

// ---- cross-GPU stage -------------------------------------------------------
// (kh_launder: the lane index as the optimiser cannot see through it.  The per-lane window addresses of the cross-GPU
// stage are loop-invariant; left visible they are hoisted out of the interval loop and -- in kernels that sit at the
// register limit -- carried across the whole sweep in scratch: 20 B/lane in kh_q2_forward_update<false, true, false>)
__device__ __forceinline__ int kh_launder(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ void kh_p2p_publish(const KhExchange &ex, int parity, int L, int lane_in,
                                               const double *values, unsigned int epoch) {
    const int lane = kh_launder(lane_in);
    // lane -> (peer, l, half): world * L * 2 <= 64 stores, one per lane
    const int per_peer = 2 * L;
    if (lane < ex.world * per_peer) {
        const int peer = lane / per_peer, rem = lane % per_peer, l = rem >> 1, half_sel = rem & 1;
        const kh_u64 bits = (kh_u64)__double_as_longlong(values[l]);
        const kh_u64 half = half_sel ? (bits & 0xffffffffull) : (bits >> 32);
        kh_u64 *g = ex.peer_windows[peer] + (((size_t)parity * ex.world + ex.rank) * L + l) * 2 + half_sel;
        __hip_atomic_store(g, ((kh_u64)epoch << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// one full wave; lane -> (rank r, control l) for lane < world * L; total in rank order
template <int MAXL>
__device__ __forceinline__ bool kh_p2p_gather(const KhExchange &ex, int parity, int L, unsigned int epoch, int lane_in,
                                              double (&out)[MAXL]) {
    const int lane = kh_launder(lane_in);
    const int pairs = ex.world * L;
    const bool active = lane < pairs;
    const kh_u64 *g = ex.my_window + ((size_t)parity * ex.world * L + (active ? lane : 0)) * 2;
    kh_u64 a = 0, b = 0;
    long long t0 = 0;  // (taken when the first poll fails, see kh_gather)
    unsigned int spins = 0;
    for (;;) {
        if (active) {
            a = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            b = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const bool ok = !active || (((unsigned int)(a >> 32) == epoch) && ((unsigned int)(b >> 32) == epoch));
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(1);
        if (spins == 0) t0 = wall_clock64();
        if ((++spins & 255u) == 0) {
            const bool gave_up =
                (wall_clock64() - t0 > ex.timeout_ticks) ||
                (__hip_atomic_load(ex.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u);
            if (__any(gave_up)) {
                if (lane == 0) __hip_atomic_store(ex.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
    const kh_u64 bits = ((a & 0xffffffffull) << 32) | (b & 0xffffffffull);
    const double v = active ? __longlong_as_double((long long)bits) : 0.0;
#pragma unroll
    for (int l = 0; l < MAXL; ++l) {
        double acc = 0.0;
        if (l < L)
            for (int r = 0; r < ex.world; ++r) acc += readlane_f64(v, r * L + l);
        out[l] = acc;
    }
    return true;
}

// The whole per-interval exchange, called by ONE full wave of every workgroup:
// publish this workgroup's partial sums, gather the GPU's total, and (sharded
// runs) exchange the GPU totals across ranks.  `n` = interval index.
// It comes in two halves so that a kernel can put independent work between
// the store and the first poll (the stores need ~1 us to become visible).
__device__ __forceinline__ void kh_exchange_publish(const KhExchange &ex, int n, int wg, int L, int lane,
                                                    const double *part) {
    if (ex.G == 1 && ex.world == 1) return;  // a single workgroup on a single GPU: nothing to exchange
    kh_publish(ex, n & 1, wg, L, lane, part, (unsigned)(n + 1));
}
// P2P = false: an instantiation for a single GPU (ex.world == 1 by the caller's word) without the cross-GPU stage -- the
// stage is a run-time branch otherwise, and its code, inlined into a kernel that sits at the register limit, costs the
// single-GPU path registers it does not use (cooperative update kernel: 236 -> 230 VGPRs, 20.0 -> 19.3 ms on config 4)
template <int MAXL, int CH = KH_GATHER_CHUNKS, bool P2P = true>
__device__ __forceinline__ bool kh_exchange_collect(const KhExchange &ex, int n, int wg, int L, int lane,
                                                    const double *part, double (&out)[MAXL]) {
    if (ex.G == 1 && ex.world == 1) {
#pragma unroll
        for (int l = 0; l < MAXL; ++l) out[l] = l < L ? part[l] : 0.0;
        return true;
    }
    const int parity = n & 1;
    const bool timed = P2P && ex.world > 1 && wg == 0 && ex.wait_ticks != nullptr;  // (uniform; scalar clock reads)
    long long t0 = 0;
    if (timed) t0 = wall_clock64();
    if (!kh_gather<MAXL, CH>(ex, parity, L, (unsigned)(n + 1), lane, out)) return false;
    if (P2P && ex.world > 1) {
        long long t1 = 0;
        if (timed) t1 = wall_clock64();
        const unsigned int epoch = ex.epoch_base + (unsigned)(n + 1);
        if (wg == 0 && n != ex.fail_at) kh_p2p_publish(ex, parity, L, lane, out, epoch);
        if (!kh_p2p_gather<MAXL>(ex, parity, L, epoch, lane, out)) return false;
        if (timed && lane == 0) {
            const long long t2 = wall_clock64();
            atomicAdd(ex.wait_ticks, (unsigned long long)(t1 - t0));
            atomicAdd(ex.wait_ticks + 1, (unsigned long long)(t2 - t1));
        }
    }
    return true;
}
template <int MAXL, int CH = KH_GATHER_CHUNKS, bool P2P = true>
__device__ __forceinline__ bool kh_exchange(const KhExchange &ex, int n, int wg, int L, int lane,
                                            const double *part, double (&out)[MAXL]) {
    kh_exchange_publish(ex, n, wg, L, lane, part);
    return kh_exchange_collect<MAXL, CH, P2P>(ex, n, wg, L, lane, part, out);
}
