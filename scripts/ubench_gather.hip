// LDS gather rate of one CU, the bound the sparse kernels (kh_ell.h) are priced against: every lane reads E 16-byte vector
// elements per term at the byte offsets its matrix row holds and multiplies them by its matrix entries (4 FMAs each).
// Patterns: "band" -- row r reads columns r + s_e (a Lindbladian's near-diagonal structure: neighbouring lanes read
// neighbouring elements), "random" -- arbitrary columns.  Prints GB/s per CU of gathered bytes and the clocks per
// ds_read_b128 wave-instruction.  Build + run (GPU box):
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_gather.hip -o /tmp/ubench_gather && /tmp/ubench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int E, bool FMA>
__global__ void __launch_bounds__(512) gather(const int *offs, double2 *out, int terms, int N) {
    extern __shared__ __attribute__((aligned(16))) char x[];
    const int tid = threadIdx.x;
    for (int i = tid; i < N; i += 512) ((double2 *)x)[i] = make_double2(1.0 + i, 0.5 * i);
    int off[E];
    double2 a[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        off[e] = offs[e * 512 + tid];
        a[e] = make_double2(1e-3 * (e + 1), -1e-3 * tid);
    }
    __syncthreads();
    double2 s = make_double2(0.0, 0.0);
    for (int t = 0; t < terms; ++t) {
#pragma unroll
        for (int e0 = 0; e0 < E; e0 += 4) {
            double2 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *(const double2 *)(x + off[e0 + q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (FMA) {
                    s.x = fma(a[e0 + q].x, v[q].x, fma(-a[e0 + q].y, v[q].y, s.x));
                    s.y = fma(a[e0 + q].x, v[q].y, fma(a[e0 + q].y, v[q].x, s.y));
                } else {
                    s.x += v[q].x;
                    s.y += v[q].y;
                }
            }
        }
        // (the offsets are opaque to the optimiser in every term: the loads cannot be hoisted out of the loop)
#pragma unroll
        for (int e = 0; e < E; ++e) asm volatile("" : "+v"(off[e]));
    }
    out[blockIdx.x * 512 + tid] = s;
}

template <int E, bool FMA>
static void run(const char *name, const std::vector<int> &h_off, int N, int wgs) {
    int *d_off;
    double2 *d_out;
    hipMalloc(&d_off, h_off.size() * sizeof(int));
    hipMemcpy(d_off, h_off.data(), h_off.size() * sizeof(int), hipMemcpyHostToDevice);
    hipMalloc(&d_out, (size_t)wgs * 512 * sizeof(double2));
    const int terms = 20000;
    hipEvent_t t0, t1;
    hipEventCreate(&t0);
    hipEventCreate(&t1);
    gather<E, FMA><<<wgs, 512, N * 16>>>(d_off, d_out, 100, N);
    hipEventRecord(t0);
    gather<E, FMA><<<wgs, 512, N * 16>>>(d_off, d_out, terms, N);
    hipEventRecord(t1);
    hipEventSynchronize(t1);
    float ms = 0;
    hipEventElapsedTime(&ms, t0, t1);
    const double bytes = (double)terms * 512 * E * 16;  // per workgroup = per CU
    const double gbs = bytes / (ms * 1e-3) / 1e9;
    printf("%-28s E=%2d %s  %d workgroups  %8.1f GB/s per CU  %6.1f B/clk at 2.4 GHz  %5.1f clk per ds_read_b128 wave-instruction\n", name, E,
           FMA ? "gather+fma" : "gather    ", wgs, gbs, gbs / 2.4, 2.4e9 * (ms * 1e-3) / ((double)terms * 8 * E));
    hipFree(d_off);
    hipFree(d_out);
}

int main() {
    const int N = 512;
    for (int pattern = 0; pattern < 3; ++pattern) {
        std::vector<int> off((size_t)32 * 512);
        const int shifts[8] = {0, 1, -1, 25, -25, 2, 26, -26};  // a d = 25 ladder's Liouvillian: +-1, +-d, ...
        srand(7);
        for (int e = 0; e < 32; ++e)
            for (int t = 0; t < 512; ++t) {
                int col = pattern == 0 ? (t + shifts[e % 8] + 8 * (e / 8) + N) % N : pattern == 1 ? rand() % N : t;
                off[(size_t)e * 512 + t] = col * 16;
            }
        const char *name = pattern == 0 ? "band (Lindbladian-like)" : pattern == 1 ? "random columns" : "own row (no gather)";
        run<8, true>(name, off, N, 256);
        run<8, false>(name, off, N, 256);
        run<16, true>(name, off, N, 256);
        run<8, true>(name, off, N, 1);
    }
    return 0;
}
