// Micro-benchmark: can a register-tile phase of the vector-FMA kind (kh_tile64q2.h: 8 broadcast reads, 32 fp64 FMAs,
// row sums, one write, one barrier) carry a matrix-core product of the SAME tile registers for free?
//   MODE 0: the vector phase alone
//   MODE 1: + 16 v_mfma_f64_4x4x4 (4 real columns: a second complex vector pair) issued as one block
//   MODE 2: the same 16, one after every second complex FMA (interleaved in program order)
// Lane map = KhLanes<true> of kh_tile64.h: the lane's 8 tile elements are the A operands of eight 4x4 blocks.
// Build: hipcc --offload-arch=gfx950 -O3 -I krotov_amd/csrc scripts/ubench_hybrid.hip -o build/ubench_hybrid
#include <hip/hip_runtime.h>
#include <cstdio>
#include "kh_common.h"
#include "kh_tile64.h"

template <int MODE>
__global__ void __launch_bounds__(512) k(const cplx *op, cplx *out, int iters) {
    __shared__ __attribute__((aligned(16))) cplx buf[2][64];
    __shared__ __attribute__((aligned(16))) double xop[2][512];  // second vector pair in operand order (8 k-blocks)
    typedef KhLanes<true> L;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, cg = L::cg(lane);
    cplx a[8];
    for (int j = 0; j < 8; ++j) a[j] = op[(wave * 8 + L::row_in(lane)) * 64 + cg + 8 * j];
    if (tid < 64) {
        buf[0][tid] = c_make(1.0 / (tid + 1), 0.5);
        buf[1][tid] = c_make(0.25, 1.0 / (tid + 2));
    }
    xop[0][tid] = 1e-3 * tid;
    xop[1][tid] = 2e-3;
    __syncthreads();
    const int row = wave * 8 + L::row_out(lane);
    const bool writer = (lane & 7) == 0;
    cplx state = c_make(0, 0);
    double keep = 0.0;
    int cur = 0;
    for (int it = 0; it < iters; ++it) {
        cplx xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = buf[cur][cg + 8 * j];
        double xo[8];
        if (MODE != 0) {
            const double2 *xp = (const double2 *)&xop[cur][0];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double2 v = xp[q * 64 + lane];
                xo[2 * q] = v.x;
                xo[2 * q + 1] = v.y;
            }
        }
        cplx yb = c_make(0.0, 0.0);
        double d1[2] = {0.0, 0.0}, d2[2] = {0.0, 0.0};
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                d1[j & 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j].x, xo[j], d1[j & 1], 0, 0, 0);
                d2[j & 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j].y, xo[j], d2[j & 1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c_fma(yb, a[j], xv[j]);
            if (MODE == 2) {
                d1[j & 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j].x, xo[j], d1[j & 1], 0, 0, 0);
                d2[j & 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j].y, xo[j], d2[j & 1], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // 4 VALU
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMA
            }
        }
        const double c2 = -1e-3;
        const double t2x = L::rowsum(yb.x, c2), t2y = L::rowsum(yb.y, c2);
        state.x += t2x;
        state.y += t2y;
        if (writer) buf[cur ^ 1][row] = c_make(t2x, t2y);
        if (MODE != 0) {
            // (the real kernel combines re/im and the two column halves here: one DPP step and an FMA each)
            double y = (d1[0] + d1[1]) + 0.5 * (d2[0] + d2[1]);
            y += dpp_move<KH_DPP_ROR4>(y);
            keep += y;
            xop[cur ^ 1][tid] = y * 1e-3;
        }
        __syncthreads();
        cur ^= 1;
    }
    out[blockIdx.x * 512 + tid] = c_make(state.x + keep, state.y);
}

template <int MODE>
void run(const char *name, const cplx *op, cplx *out) {
    const int iters = 40000;
    k<MODE><<<256, 512>>>(op, out, iters);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventRecord(a);
    k<MODE><<<256, 512>>>(op, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-58s %7.1f ns/phase (%6.0f cycles @2.4 GHz)\n", name, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
}

int main() {
    cplx *op, *out;
    hipMalloc(&op, 64 * 64 * sizeof(cplx));
    hipMalloc(&out, 256 * 512 * sizeof(cplx));
    hipMemset(op, 0, 64 * 64 * sizeof(cplx));
    run<0>("vector phase alone", op, out);
    run<1>("+ 16 MFMA as one block", op, out);
    run<2>("+ 16 MFMA interleaved with the FMAs", op, out);
    return 0;
}
