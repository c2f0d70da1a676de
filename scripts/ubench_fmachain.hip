// fp64 FMA issue rate vs number of independent accumulator chains and waves per SIMD (dev tool).
// hipcc --offload-arch=gfx950 -O3 scripts/ubench_fmachain.hip -o /tmp/ubench_fc && /tmp/ubench_fc
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS>
__global__ void __launch_bounds__(512) k(const double* in, double* out, long long* cyc, int iters) {
    double a[16], x[16];
    for (int j = 0; j < 16; ++j) { a[j] = in[threadIdx.x + 64 * j]; x[j] = in[threadIdx.x + 7 * j + 1]; }
    double acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = 0.0;
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        // 64 FMAs per iteration, CHAINS independent dependency chains, consecutive FMAs of a chain adjacent
        // in pairs (as c_fma generates them) when CHAINS == 2
#pragma unroll
        for (int q = 0; q < 64; ++q) {
            const int c = (CHAINS == 2) ? ((q >> 1) & 1) : (q % CHAINS);
            acc[c] = fma(a[q & 15], x[(q * 5 + 3) & 15], acc[c]);
        }
        // keep the optimiser from hoisting: perturb one operand with the result
        x[it & 15] += acc[0] * 1e-300;
    }
    long long t1 = clock64();
    double s = 0.0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CHAINS>
void run(double* in, double* out, long long* cyc, int threads) {
    const int iters = 20000;
    k<CHAINS><<<256, threads>>>(in, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    k<CHAINS><<<256, threads>>>(in, out, cyc, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const int waves_per_simd = threads / 256;
    printf("chains=%d waves/SIMD=%d: %.2f clk per FMA per SIMD (%.1f ns/iter)\n", CHAINS, waves_per_simd,
           (double)h / iters / 64.0 / waves_per_simd, ms * 1e6 / iters);
}

int main() {
    double* in; double* out; long long* cyc;
    hipMalloc(&in, 8192 * 8); hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 16);
    hipMemset(in, 0, 8192 * 8);
    for (int threads : {256, 512}) {
        run<1>(in, out, cyc, threads);
        run<2>(in, out, cyc, threads);
        run<4>(in, out, cyc, threads);
        run<8>(in, out, cyc, threads);
    }
    return 0;
}
