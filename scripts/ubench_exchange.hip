// Micro-benchmark behind DESIGN.md section 3.3 ("would an XCD-local exchange help the update sweep?"): the per-interval
// sum over 512 workgroups (two per CU, 512 threads each, as kh_q2_forward_update runs), iterated with a data
// dependency from one sum to the next publication, in two forms:
//   FLAT  every workgroup publishes one epoch-tagged granule and gathers all 512 (agent scope, memory side) -- what
//         the kernels do (kh_common.h);
//   XCD   two stages: the 64 workgroups of an XCD (blockIdx % 8, checked against XCC_ID) exchange through their L2
//         (workgroup-scope stores, agent-scope loads), every member forms the XCD's sum, member 0 publishes it (a slot
//         per XCD, each in its own 128-byte line; all 64 members storing the same value to one slot costs 13 us: the
//         stores of 512 CUs to one line are serialised at the memory side) and all gather the 8 XCD sums.
// Both end with the same LDS broadcast + barrier.  Prints ns per exchange.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_exchange.hip -o build/ubench_exchange
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned long long u64;
#define WGS 512
#define THREADS 512

__device__ __forceinline__ unsigned int xcc_id() { return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 0xf; }

__device__ __forceinline__ float wave_sum(float v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
// (payload: a float in the low half of the granule; the tag in the high half)
__device__ __forceinline__ u64 pack(unsigned int epoch, float v) { return ((u64)epoch << 32) | __float_as_uint(v); }

template <int MODE, int SLEEP>
__global__ void __launch_bounds__(THREADS) k(u64 *flat, u64 *local, u64 *cross, int iters, float *out, int *misplaced,
                                              int work, int d1, int d2) {
    __shared__ float red[8];
    __shared__ float total;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wg = blockIdx.x, grp = wg & 7, member = wg >> 3;
    if (tid == 0 && xcc_id() != (unsigned)grp) atomicAdd(misplaced, 1);
    float mine = 1.0f + 1e-3f * wg;
    for (int it = 1; it <= iters; ++it) {
        const int par = it & 1;
        float sum;
        for (int w = 0; w < work; ++w) __builtin_amdgcn_s_sleep(1);  // the interval's products (no memory traffic)
        if (MODE == 3) {
            sum = mine;
        } else if (MODE == 0) {
            if (tid == 0) __hip_atomic_store(flat + par * WGS + wg, pack(it, mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            u64 g;
            for (int w = 0; w < d1; ++w) __builtin_amdgcn_s_sleep(1);
            do {
                g = __hip_atomic_load(flat + par * WGS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (SLEEP && (unsigned int)(g >> 32) != (unsigned int)it) __builtin_amdgcn_s_sleep(1);
            } while ((unsigned int)(g >> 32) != (unsigned int)it);
            const float s = wave_sum(__uint_as_float((unsigned int)g));
            if (lane == 0) red[wave] = s;
            __syncthreads();
            if (tid == 0) total = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
            __syncthreads();
            sum = total;
        } else {
            if (wave == 0) {
                if (lane == 0) {
                    if (MODE == 1)
                        __hip_atomic_store(local + (par * 8 + grp) * 64 + member, pack(it, mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else
                        __hip_atomic_store(local + (par * 8 + grp) * 64 + member, pack(it, mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                u64 g;
                for (int w = 0; w < d1; ++w) __builtin_amdgcn_s_sleep(1);
                do {
                    g = __hip_atomic_load(local + (par * 8 + grp) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (SLEEP && (unsigned int)(g >> 32) != (unsigned int)it) __builtin_amdgcn_s_sleep(1);
                } while ((unsigned int)(g >> 32) != (unsigned int)it);
                const float sg = wave_sum(__uint_as_float((unsigned int)g));
                if (lane == 0 && member == 0) __hip_atomic_store(cross + (par * 8 + grp) * 16, pack(it, sg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                float v = 0.0f;
                for (int w = 0; w < d2; ++w) __builtin_amdgcn_s_sleep(1);
                if (lane < 8) {
                    do {
                        g = __hip_atomic_load(cross + (par * 8 + lane) * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (SLEEP && (unsigned int)(g >> 32) != (unsigned int)it) __builtin_amdgcn_s_sleep(1);
                    } while ((unsigned int)(g >> 32) != (unsigned int)it);
                    v = __uint_as_float((unsigned int)g);
                }
                const float s = wave_sum(v);
                if (lane == 0) total = s;
            }
            __syncthreads();
            sum = total;
            __syncthreads();
        }
        mine = 1.0f + 1e-9f * sum + 1e-3f * wg;  // (the next publication depends on the sum)
    }
    if (tid == 0) out[wg] = mine;
}

template <int MODE, int SLEEP>
static float run(u64 *flat, u64 *local, u64 *cross, float *out, int *misplaced, int work, int d1, int d2, int *mis) {
    const int iters = 20000;
    float ms = 0.0f;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(flat, 0, 2 * WGS * 8);
        hipMemset(local, 0, 2 * 8 * 64 * 8);
        hipMemset(cross, 0, 2 * 8 * 16 * 8);
        hipMemset(misplaced, 0, 4);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipEventRecord(a);
        void *args[] = {&flat, &local, &cross, (void *)&iters, &out, &misplaced, &work, &d1, &d2};
        const hipError_t err = hipLaunchCooperativeKernel((const void *)k<MODE, SLEEP>, dim3(WGS), dim3(THREADS), args, 0, 0);
        if (err != hipSuccess) {
            printf("launch failed: %s\n", hipGetErrorString(err));
            return -1.0f;
        }
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    hipMemcpy(mis, misplaced, 4, hipMemcpyDeviceToHost);
    return ms * 1e6f / iters;
}

int main() {
    u64 *flat, *local, *cross;
    float *out;
    int *misplaced, mis = 0;
    hipMalloc(&flat, 2 * WGS * 8);
    hipMalloc(&local, 2 * 8 * 64 * 8);
    hipMalloc(&cross, 2 * 8 * 16 * 8);
    hipMalloc(&out, WGS * 4);
    hipMalloc(&misplaced, 4);
    const int work = 100;  // s_sleep units (64 clocks) of "products" between two exchanges: about 3 us
    const float base = run<3, 0>(flat, local, cross, out, misplaced, work, 0, 0, &mis);
    printf("interval without an exchange (%d sleep units): %.0f ns; workgroups not on XCD blockIdx %% 8: %d of %d\n", work, base, mis, WGS);
    printf("exchange cost = interval - that; delays in sleep units before the first poll of a stage\n");
    for (int d1 : {0, 4, 8, 12, 16, 24})
        printf("flat (512 granules, agent scope), delay %2d:                     %6.0f ns\n", d1,
               run<0, 1>(flat, local, cross, out, misplaced, work, d1, 0, &mis) - base);
    for (int d1 : {0, 4, 8, 12})
        for (int d2 : {0, 4, 8, 12})
            printf("two stages (64 in the XCD's L2, then 8 across), delays %2d, %2d:  %6.0f ns   agent-scope stores in stage 1: %6.0f ns\n",
                   d1, d2, run<1, 1>(flat, local, cross, out, misplaced, work, d1, d2, &mis) - base,
                   run<2, 1>(flat, local, cross, out, misplaced, work, d1, d2, &mis) - base);
    return 0;
}
