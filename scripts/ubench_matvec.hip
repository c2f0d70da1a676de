// Micro-benchmark of the pieces of one register-tile matvec phase (dev tool).
// hipcc --offload-arch=gfx950 -O3 -I krotov_amd/csrc scripts/ubench_matvec.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include "kh_common.h"

template <int RPT, int MODE>
__global__ void __launch_bounds__(512 / RPT) k(const cplx* op, cplx* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) cplx buf[2][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, cg = lane & 7;
    cplx a[RPT][8];
    for (int r = 0; r < RPT; ++r)
        for (int j = 0; j < 8; ++j) a[r][j] = op[(wave * 8 * RPT + r * 8 + (lane >> 3)) * 64 + cg + 8 * j];
    if (tid < 64) { buf[0][tid] = c_make(1.0 / (tid + 1), 0.5); buf[1][tid] = c_make(0.25, 1.0 / (tid + 2)); }
    __syncthreads();
    cplx state[RPT];
    for (int r = 0; r < RPT; ++r) state[r] = c_make(0, 0);
    cplx xv[8];
    for (int j = 0; j < 8; ++j) xv[j] = buf[0][cg + 8 * j];
    long long t0 = clock64();
    long long w0 = wall_clock64();
    cplx *xin = buf[0], *xout = buf[1];
    for (int it = 0; it < iters; ++it) {
        if (MODE & 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = xin[cg + 8 * j];
        }
        cplx y[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            cplx acc = c_make(0.0, 0.0);
            if (MODE & 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) c_fma(acc, a[r][j], xv[j]);
            } else {
                acc = xv[r];
            }
            if (MODE & 4) { acc.x = sum8(acc.x); acc.y = sum8(acc.y); }
            y[r] = acc;
        }
        const cplx coef = c_make(0.0, -1e-3 * kh_inv_table[(it & 15) + 1]);
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const cplx t = c_mul(coef, y[r]);
            state[r].x += t.x; state[r].y += t.y;
            if (MODE & 8) { if (cg == 0) xout[wave * 8 * RPT + r * 8 + (lane >> 3)] = t; }
            else { xv[r].x += t.x * 1e-9; }
        }
        if (MODE & 8) { __syncthreads(); cplx* tmp = xin; xin = xout; xout = tmp; }
    }
    long long t1 = clock64();
    long long w1 = wall_clock64();
    for (int r = 0; r < RPT; ++r) out[blockIdx.x * 512 + tid * RPT + r] = state[r];
    if (tid == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

template <int RPT, int MODE>
void run(const char* name, const cplx* op, cplx* out, long long* cyc, int grid) {
    const int iters = 20000;
    k<RPT, MODE><<<grid, 512 / RPT>>>(op, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    k<RPT, MODE><<<grid, 512 / RPT>>>(op, out, cyc, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h[2]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-34s RPT=%d grid=%3d: %7.1f ns/iter  %7.1f clk/iter (s_memtime)  %6.1f wallclk-ticks/iter  -> %.2f GHz\n",
           name, RPT, grid, ms * 1e6 / iters, (double)h[0] / iters, (double)h[1] / iters,
           (double)h[0] / iters / (ms * 1e6 / iters));
}

int main() {
    cplx* op; cplx* out; long long* cyc;
    hipMalloc(&op, 64 * 64 * sizeof(cplx)); hipMalloc(&out, 256 * 512 * sizeof(cplx)); hipMalloc(&cyc, 16);
    hipMemset(op, 0, 64 * 64 * sizeof(cplx));
    for (int grid : {1, 256}) {
        run<2, 2>("fma only", op, out, cyc, grid);
        run<2, 3>("lds read + fma", op, out, cyc, grid);
        run<2, 6>("fma + dpp reduce", op, out, cyc, grid);
        run<2, 7>("lds read + fma + reduce", op, out, cyc, grid);
        run<2, 8>("lds write + barrier only", op, out, cyc, grid);
        run<2, 9>("lds read/write + barrier", op, out, cyc, grid);
        run<2, 11>("lds rw + barrier + fma", op, out, cyc, grid);
        run<2, 15>("full phase", op, out, cyc, grid);
        run<1, 15>("full phase", op, out, cyc, grid);
        run<1, 2>("fma only", op, out, cyc, grid);
        run<1, 6>("fma + dpp reduce", op, out, cyc, grid);
    }
    return 0;
}
