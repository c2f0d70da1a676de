// Micro-benchmark: one register-tile matvec phase chain (LDS read -> FMAs -> row sums -> LDS write -> barrier) of a
// 64 x 64 complex operator held by a 512-thread workgroup, for two lane maps (dev tool, round 6):
//   M18: lane = 1 row x 8 columns  (the kernels' map: 8 broadcast ds_read_b128 per lane and phase, row sum over 8 lanes)
//   M24: lane = 2 rows x 4 columns (4 reads per lane and phase; one reduce-scatter exchange with the neighbour lane, then
//        the row sum over 8 lanes of the same parity)
// hipcc --offload-arch=gfx950 -O3 -I krotov_amd/csrc scripts/ubench_lanemap.hip -o /tmp/ubench_lanemap && /tmp/ubench_lanemap
#include <hip/hip_runtime.h>
#include <cstdio>
#include "kh_common.h"

template <int MAP, int PRODUCTS>
__global__ void __launch_bounds__(512) k(const cplx *op, cplx *out, long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) cplx buf[2][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    cplx a[PRODUCTS][8];
    int row_w = 0;       // the row this lane writes
    bool writer = false;
    if (MAP == 18) {
        const int cg = lane & 7, r = wave * 8 + (lane >> 3);
        for (int p = 0; p < PRODUCTS; ++p)
            for (int j = 0; j < 8; ++j) a[p][j] = op[(p * 64 + r) * 64 + cg + 8 * j];
        row_w = r;
        writer = cg == 0;
    } else {
        // 16 lanes share the rows (ra, ra + 1); lane holds columns c4 + 16 j; registers [0..3] = the row it KEEPS in the
        // exchange (even lanes: ra, odd lanes: ra + 1), [4..7] = the row it SENDS
        const int c4 = lane & 15, ra = wave * 8 + 2 * (lane >> 4), keep = ra + (lane & 1), send = ra + 1 - (lane & 1);
        for (int p = 0; p < PRODUCTS; ++p)
            for (int j = 0; j < 4; ++j) {
                a[p][j] = op[(p * 64 + keep) * 64 + c4 + 16 * j];
                a[p][4 + j] = op[(p * 64 + send) * 64 + c4 + 16 * j];
            }
        row_w = keep;
        writer = c4 < 2;  // lanes 0, 1 of the 16: one per row
    }
    if (tid < 64) {
        buf[0][tid] = c_make(1.0 / (tid + 1), 0.5);
        buf[1][tid] = c_make(0.25, 1.0 / (tid + 2));
    }
    __syncthreads();
    cplx state = c_make(0, 0);
    int cur = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const double cf = -1e-3 * kh_inv_table[(it & 15) + 1];
        cplx t[PRODUCTS];
        if (MAP == 18) {
            const int cg = lane & 7;
            cplx xv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = buf[cur][cg + 8 * j];
#pragma unroll
            for (int p = 0; p < PRODUCTS; ++p) {
                cplx acc = c_make(0.0, 0.0);
#pragma unroll
                for (int j = 0; j < 8; ++j) c_fma(acc, a[p][j], xv[j]);
                t[p] = c_make(cf * sum8(acc.x), cf * sum8(acc.y));
            }
        } else {
            const int c4 = lane & 15;
            cplx xv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = buf[cur][c4 + 16 * j];
#pragma unroll
            for (int p = 0; p < PRODUCTS; ++p) {
                cplx k0 = c_make(0.0, 0.0), s0 = c_make(0.0, 0.0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    c_fma(k0, a[p][j], xv[j]);
                    c_fma(s0, a[p][4 + j], xv[j]);
                }
                // reduce-scatter with the neighbour lane, then lanes of the same parity: ^2, +4, +8 within the 16
                double x = k0.x + dpp_move<KH_DPP_XOR1>(s0.x), y = k0.y + dpp_move<KH_DPP_XOR1>(s0.y);
                x += dpp_move<KH_DPP_XOR2>(x);
                y += dpp_move<KH_DPP_XOR2>(y);
                x += dpp_move<KH_DPP_ROR4>(x);
                y += dpp_move<KH_DPP_ROR4>(y);
                x += dpp_move<KH_DPP_ROR8>(x);
                y += dpp_move<KH_DPP_ROR8>(y);
                t[p] = c_make(cf * x, cf * y);
            }
        }
#pragma unroll
        for (int p = 0; p < PRODUCTS; ++p) {
            state.x += t[p].x;
            state.y += t[p].y;
        }
        if (writer) buf[cur ^ 1][row_w] = t[0];
        __syncthreads();
        cur ^= 1;
    }
    const long long t1 = clock64();
    out[blockIdx.x * 512 + tid] = state;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MAP, int PRODUCTS>
void run(const char *name, const cplx *op, cplx *out, long long *cyc, int grid) {
    const int iters = 20000;
    k<MAP, PRODUCTS><<<grid, 512>>>(op, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventRecord(a);
    k<MAP, PRODUCTS><<<grid, 512>>>(op, out, cyc, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    long long h[1];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s grid=%3d: %7.1f ns/phase  %7.1f clk/phase\n", name, grid, ms * 1e6 / iters, (double)h[0] / iters);
}

int main() {
    cplx *op, *out;
    long long *cyc;
    hipMalloc(&op, 2 * 64 * 64 * sizeof(cplx));
    hipMalloc(&out, 256 * 512 * sizeof(cplx));
    hipMalloc(&cyc, 16);
    hipMemset(op, 0, 2 * 64 * 64 * sizeof(cplx));
    for (int grid : {1, 256}) {
        run<18, 1>("M18 1 row x 8 cols, one product per phase", op, out, cyc, grid);
        run<24, 1>("M24 2 rows x 4 cols, one product per phase", op, out, cyc, grid);
        run<18, 2>("M18, two products per phase (A and B)", op, out, cyc, grid);
        run<24, 2>("M24, two products per phase (A and B)", op, out, cyc, grid);
    }
    return 0;
}
