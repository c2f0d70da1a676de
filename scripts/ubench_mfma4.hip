// Layout probe for v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4): which output lanes see the product of
// A-lane la and B-lane lb.  Build: hipcc --offload-arch=gfx950 -O2 scripts/ubench_mfma4.hip -o /tmp/ubench_mfma4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(unsigned long long *hits) {  // hits[la * 64 + lb] = ballot of output lanes != 0
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) hits[la * 64 + lb] = m;
        }
}

int main() {
    unsigned long long *d;
    hipMalloc(&d, 4096 * 8);
    probe<<<1, 64>>>(d);
    std::vector<unsigned long long> h(4096);
    hipMemcpy(h.data(), d, 4096 * 8, hipMemcpyDeviceToHost);
    // for every A lane: the B lanes it pairs with and the output lanes
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d pairs with B lanes:", la);
        for (int lb = 0; lb < 64; ++lb)
            if (h[la * 64 + lb]) {
                printf(" %d->", lb);
                for (int o = 0; o < 64; ++o)
                    if (h[la * 64 + lb] >> o & 1) printf("%d,", o);
            }
        printf("\n");
    }
    return 0;
}
