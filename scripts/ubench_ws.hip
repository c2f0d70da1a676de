// Micro-benchmark of the wave-specialised phase (dev tool): which part is the critical path?
#include <hip/hip_runtime.h>
#include <cstdio>
#include "kh_common.h"

// MODE bits: 1 A-wave FMAs, 2 B-wave FMAs, 4 B reduce, 8 LDS write of t2, 16 LDS reads of x, 32 setprio
template <int MODE>
__global__ void __launch_bounds__(512) k(const cplx* op, cplx* out, int iters) {
    __shared__ __attribute__((aligned(16))) cplx buf[2][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, cg = lane & 7;
    const bool is_B = wave < 4;
    const int gw = wave & 3;
    cplx m0[2][8];
    for (int r = 0; r < 2; ++r)
        for (int j = 0; j < 8; ++j) m0[r][j] = op[((gw * 16 + r * 8 + (lane >> 3)) * 64 + cg + 8 * j) & 4095];
    if (tid < 64) { buf[0][tid] = c_make(1.0 / (tid + 1), 0.5); buf[1][tid] = c_make(0.25, 1.0 / (tid + 2)); }
    __syncthreads();
    if ((MODE & 32) && is_B) __builtin_amdgcn_s_setprio(3);
    cplx ev0 = c_make(0, 0), ev1 = c_make(0, 0);
    cplx xv[8];
    for (int j = 0; j < 8; ++j) xv[j] = buf[0][cg + 8 * j];
    int cur = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE & 16) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = buf[cur][cg + 8 * j];
        }
        if (is_B) {
            cplx y0 = c_make(0, 0), y1 = c_make(0, 0);
            if (MODE & 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { c_fma(y0, m0[0][j], xv[j]); c_fma(y1, m0[1][j], xv[j]); }
            } else { y0 = xv[0]; y1 = xv[1]; }
            if (MODE & 4) { y0.x = sum8(y0.x); y0.y = sum8(y0.y); y1.x = sum8(y1.x); y1.y = sum8(y1.y); }
            const double c2 = 1e-3 * kh_inv_table[(it & 15) + 1];
            const double t0x = c2 * y0.x, t0y = c2 * y0.y, t1x = c2 * y1.x, t1y = c2 * y1.y;
            ev0.x += t0x; ev0.y += t0y; ev1.x += t1x; ev1.y += t1y;
            if (MODE & 8) {
                if (cg == 0) { buf[cur ^ 1][gw * 16 + (lane >> 3)] = c_make(t0x, t0y); buf[cur ^ 1][gw * 16 + 8 + (lane >> 3)] = c_make(t1x, t1y); }
            } else { xv[0].x += t0x * 1e-9; xv[1].x += t1x * 1e-9; }
        } else {
            if (MODE & 1) {
                cplx y0 = c_make(0, 0), y1 = c_make(0, 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) { c_fma(y0, m0[0][j], xv[j]); c_fma(y1, m0[1][j], xv[j]); }
                const cplx c1 = c_make(0.0, -1e-3 * kh_inv_table[(it & 15) + 1]);
                c_fma(ev0, c1, y0); c_fma(ev1, c1, y1);
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    out[blockIdx.x * 1024 + tid * 2] = ev0; out[blockIdx.x * 1024 + tid * 2 + 1] = ev1;
}

template <int MODE>
void run(const char* name, const cplx* op, cplx* out) {
    const int iters = 20000;
    k<MODE><<<256, 512>>>(op, out, iters);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    k<MODE><<<256, 512>>>(op, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-52s %7.1f ns/phase (%6.0f cyc)\n", name, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
}

int main() {
    cplx* op; cplx* out;
    hipMalloc(&op, 64 * 64 * sizeof(cplx)); hipMalloc(&out, 256 * 1024 * sizeof(cplx));
    hipMemset(op, 0, 64 * 64 * sizeof(cplx));
    run<63>("full (prio)", op, out);
    run<31>("full (no prio)", op, out);
    run<62>("no A FMAs", op, out);
    run<61>("no B FMAs (A FMAs, B reduce/write/read)", op, out);
    run<60>("no FMAs at all: read+reduce+write+barrier", op, out);
    run<59>("no reduce", op, out);
    run<56>("read + write + barrier only", op, out);
    run<48>("read + barrier only", op, out);
    run<32>("barrier only", op, out);
    run<35>("FMAs only (A+B), barrier", op, out);
    run<34>("B FMAs only, barrier", op, out);
    run<38>("B FMAs + reduce, barrier", op, out);
    return 0;
}
