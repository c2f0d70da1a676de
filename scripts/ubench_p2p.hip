// Micro-benchmark behind the multi-GPU predictions of DESIGN.md section 4: one store -> poll hop between two
// workgroups through (a) ordinary device memory with agent-scope atomics (the in-GPU exchange of kh_common.h) and
// (b) a FINE-GRAINED window with system-scope atomics -- exactly the allocation (hipExtMallocWithFlags,
// hipDeviceMallocFinegrained) and the instructions (kh_p2p_publish / kh_p2p_gather) the cross-GPU stage uses,
// here with both ends on ONE GPU (the only set-up a 1-GPU box offers): what is left out is the xGMI link itself.
// Two workgroups ping-pong an epoch-tagged 8-byte granule; a round trip is two hops.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_p2p.hip -o build/ubench_p2p
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned long long u64;

template <int SCOPE>
__global__ void pingpong(u64 *win, int rounds, long long *ticks, int other_block) {
    // block 0 and block `other_block` take part (other_block % 8 != 0: a different XCD); the rest exit
    const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == other_block ? 1 : -1);
    if (me < 0 || threadIdx.x != 0) return;
    u64 *mine = win + 16 * me, *theirs = win + 16 * (1 - me);  // (separate 128-byte lines)
    const long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        if (me == 0) {
            __hip_atomic_store(theirs, (u64)r, __ATOMIC_RELAXED, SCOPE);
            while (__hip_atomic_load(mine, __ATOMIC_RELAXED, SCOPE) != (u64)r) {
            }
        } else {
            while (__hip_atomic_load(mine, __ATOMIC_RELAXED, SCOPE) != (u64)r) {
            }
            __hip_atomic_store(theirs, (u64)r, __ATOMIC_RELAXED, SCOPE);
        }
    }
    if (me == 0) ticks[0] = wall_clock64() - t0;
}

template <int SCOPE>
static void run(const char *name, u64 *win, long long *d_ticks, int other) {
    const int rounds = 20000;
    hipMemset(win, 0, 4096);
    pingpong<SCOPE><<<256, 64>>>(win, rounds, d_ticks, other);
    hipDeviceSynchronize();
    hipMemset(win, 0, 4096);
    pingpong<SCOPE><<<256, 64>>>(win, rounds, d_ticks, other);
    long long t = 0;
    hipMemcpy(&t, d_ticks, 8, hipMemcpyDeviceToHost);
    // wall_clock64: 100 MHz
    printf("%-64s partner block %3d: %6.0f ns per hop\n", name, other, t * 10.0 / rounds / 2.0);
}

int main() {
    u64 *coarse = nullptr, *fine = nullptr;
    long long *d_ticks;
    hipMalloc(&coarse, 4096);
    hipMalloc(&d_ticks, 8);
    if (hipExtMallocWithFlags((void **)&fine, 4096, hipDeviceMallocFinegrained) != hipSuccess) {
        printf("fine-grained allocation failed\n");
        return 1;
    }
    for (int other : {8, 1, 5}) {  // 8: same XCD as block 0 (blockIdx %% 8 placement, observed), 1 and 5: other XCDs
        run<__HIP_MEMORY_SCOPE_AGENT>("device memory, agent-scope atomics (in-GPU exchange)", coarse, d_ticks, other);
        run<__HIP_MEMORY_SCOPE_SYSTEM>("fine-grained window, system-scope atomics (cross-GPU stage)", fine, d_ticks, other);
        run<__HIP_MEMORY_SCOPE_SYSTEM>("device memory, system-scope atomics", coarse, d_ticks, other);
    }
    return 0;
}
