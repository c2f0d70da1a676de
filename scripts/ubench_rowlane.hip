// Micro-benchmark (dev tool): one dependent matrix-vector phase of the register-tile kernels in two lane maps.
//   A  "tile"    : the product's map (kh_tile64q2.h): lane = 1 row x 8 columns, a wave = 8 rows x 8 column groups; the
//                  input vector is read from LDS by every lane (8 ds_read_b128: 64 KiB per workgroup and phase), row sums
//                  on the matrix core, 64 writer lanes store the new vector, one barrier.
//   B  "rowlane" : lane = row, wave w = columns 8w..8w+7.  The wave's 8 vector elements are wave-uniform: SGPR operands of
//                  the FMAs (no LDS read of the vector).  Partial row sums of the 8 waves meet in LDS (8 KiB written,
//                  8 KiB read), are added on the matrix core, and the wave's next 8 elements go to SGPRs by v_readlane.
// hipcc --offload-arch=gfx950 -O3 -I krotov_amd/csrc scripts/ubench_rowlane.hip -o /tmp/ubench_rowlane && /tmp/ubench_rowlane
#include <hip/hip_runtime.h>
#include <cstdio>
#include "kh_common.h"
#include "kh_tile64.h"

typedef KhLanes<true> L;

__global__ void __launch_bounds__(512) k_tile(const cplx *op, cplx *out, long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) cplx buf[2][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, cg = L::cg(lane);
    const int row_in = wave * 8 + L::row_in(lane), row = wave * 8 + L::row_out(lane);
    const bool writer = (lane & 7) == 0;
    cplx b[8];
    for (int j = 0; j < 8; ++j) b[j] = op[row_in * 64 + cg + 8 * j];
    if (tid < 64) buf[0][tid] = c_make(1.0 / (tid + 1), 0.5);
    __syncthreads();
    cplx state = c_make(0, 0), sacc = c_make(0, 0);
    int cur = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const double c2 = -1e-3 * kh_inv_table[(it & 7) + 1];
        cplx xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = buf[cur][cg + 8 * j];
        cplx y = c_make(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < 8; ++j) c_fma(y, b[j], xv[j]);
        const double tx = L::rowsum(y.x, c2), ty = L::rowsum(y.y, c2);
        state.x += tx;
        state.y += ty;
        sacc.x = fma(0.5, tx, sacc.x);
        sacc.y = fma(0.5, ty, sacc.y);
        if (writer) buf[cur ^ 1][row] = c_make(tx + 1e-3, ty);
        __syncthreads();
        cur ^= 1;
    }
    const long long t1 = clock64();
    out[blockIdx.x * 512 + tid] = c_make(state.x + sacc.x, state.y + sacc.y);
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

#define RL_STRIDE 72  // 16-byte slots between two waves' partial vectors: 72 = 8 mod 16 keeps the gather conflict-free

__device__ __forceinline__ double rl_uniform(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

template <bool READLANE>
__global__ void __launch_bounds__(512) k_rowlane(const cplx *op, cplx *out, long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) cplx part[2][8][RL_STRIDE];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    cplx b[8];
    for (int j = 0; j < 8; ++j) b[j] = op[lane * 64 + wave * 8 + j];
    // the wave's 8 vector elements, wave-uniform
    double xr[8], xi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        xr[j] = 1.0 / (wave * 8 + j + 1);
        xi[j] = 0.5;
    }
    const int src = L::cg(lane), rin = wave * 8 + L::row_in(lane);
    cplx state = c_make(0, 0), sacc = c_make(0, 0);
    int cur = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const double c2 = -1e-3 * kh_inv_table[(it & 7) + 1];
        cplx y = c_make(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            y.x = fma(b[j].x, xr[j], y.x);
            y.x = fma(-b[j].y, xi[j], y.x);
            y.y = fma(b[j].x, xi[j], y.y);
            y.y = fma(b[j].y, xr[j], y.y);
        }
        part[cur][wave][lane] = y;
        __syncthreads();
        const cplx v = part[cur][src][rin];
        const double tx = L::rowsum(v.x, c2), ty = L::rowsum(v.y, c2);  // rows 8 wave + row_out(lane)
        state.x += tx;
        state.y += ty;
        sacc.x = fma(0.5, tx, sacc.x);
        sacc.y = fma(0.5, ty, sacc.y);
        const double nx = tx + 1e-3, ny = ty;
        if (READLANE) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = 2 * (i & 3) + (i >> 2);  // row_out(8 m) == i
                xr[i] = rl_uniform(nx, 8 * m);
                xi[i] = rl_uniform(ny, 8 * m);
            }
        } else {  // (wrong on purpose: what the phase costs without the 32 v_readlane)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xr[i] += 1e-9;
            }
            xr[0] = rl_uniform(nx, 0);
            xi[0] = rl_uniform(ny, 0);
        }
        cur ^= 1;
    }
    const long long t1 = clock64();
    out[blockIdx.x * 512 + tid] = c_make(state.x + sacc.x, state.y + sacc.y);
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <class F>
void run(const char *name, F launch, long long *cyc, int grid) {
    const int iters = 20000;
    launch(grid, iters);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventRecord(a);
    launch(grid, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    long long h;
    hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-40s grid=%3d: %7.1f ns/phase  %7.1f s_memtime ticks/phase\n", name, grid, ms * 1e6 / iters, (double)h / iters);
}

int main() {
    cplx *op, *out;
    long long *cyc;
    hipMalloc(&op, 64 * 64 * sizeof(cplx));
    hipMalloc(&out, 256 * 512 * sizeof(cplx));
    hipMalloc(&cyc, 16);
    hipMemset(op, 0, 64 * 64 * sizeof(cplx));
    for (int grid : {1, 256}) {
        run("tile map (product)", [&](int g, int n) { k_tile<<<g, 512>>>(op, out, cyc, n); }, cyc, grid);
        run("row-lane map, SGPR operands", [&](int g, int n) { k_rowlane<true><<<g, 512>>>(op, out, cyc, n); }, cyc, grid);
        run("row-lane map without the readlanes", [&](int g, int n) { k_rowlane<false><<<g, 512>>>(op, out, cyc, n); }, cyc,
            grid);
    }
    return 0;
}
