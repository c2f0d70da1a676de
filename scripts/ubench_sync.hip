// Micro-benchmark: cost of s_barrier, ds_write+wait, ds_read latency (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double2 cplx;

template <int THREADS, int MODE>
__global__ void __launch_bounds__(THREADS) k(cplx* out, int iters) {
    __shared__ __attribute__((aligned(16))) cplx buf[2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 64) { buf[0][tid] = make_double2(1.0 / (tid + 1), 0.5); buf[1][tid] = make_double2(0.25, 0.125); }
    __syncthreads();
    cplx acc = make_double2(0, 0);
    int cur = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE & 1) __syncthreads();                         // barrier
        if (MODE & 2) {                                        // one ds_write_b128 per 8 lanes + wait
            if ((lane & 7) == 0) buf[cur ^ 1][wave * 8 + (lane >> 3)] = make_double2(acc.x + it, acc.y);
            __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0)
        }
        if (MODE & 4) {                                        // dependent ds_read (latency)
            cplx v = buf[cur][(lane + (int)acc.x) & 63];
            acc.x += v.x * 1e-30; acc.y += v.y;
        }
        if (MODE & 8) {                                        // 8 broadcast reads
            cplx s = make_double2(0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) { cplx v = buf[cur][(lane & 7) + 8 * j]; s.x += v.x; s.y += v.y; }
            acc.x += s.x * 1e-30; acc.y += s.y * 1e-30;
        }
        cur ^= 1;
    }
    out[blockIdx.x * THREADS + tid] = acc;
}

template <int THREADS, int MODE>
void run(const char* name, cplx* out) {
    const int iters = 50000;
    k<THREADS, MODE><<<256, THREADS>>>(out, iters);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    k<THREADS, MODE><<<256, THREADS>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-44s threads=%4d: %7.1f ns/iter (%6.1f cyc @2.4GHz)\n", name, THREADS, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
}

int main() {
    cplx* out; hipMalloc(&out, 256 * 1024 * sizeof(cplx));
    run<256, 1>("barrier only", out);
    run<512, 1>("barrier only", out);
    run<1024, 1>("barrier only", out);
    run<512, 2>("ds_write + waitcnt", out);
    run<512, 3>("ds_write + waitcnt + barrier", out);
    run<512, 4>("dependent ds_read", out);
    run<512, 5>("barrier + dependent ds_read", out);
    run<512, 7>("write + barrier + dependent read", out);
    run<512, 8>("8 broadcast reads", out);
    run<512, 11>("write + barrier + 8 broadcast reads", out);
    run<256, 11>("write + barrier + 8 broadcast reads", out);
    return 0;
}
