#include <hip/hip_runtime.h>
__global__ void k(int *out) {
    int v = threadIdx.x;
    int r = __builtin_amdgcn_mov_dpp(v, 0x13C, 0xF, 0xF, true);   // wave_ror:1
    int s = __builtin_amdgcn_mov_dpp(v, 0x138, 0xF, 0xF, true);   // wave_shr:1
    int l = __builtin_amdgcn_mov_dpp(v, 0x134, 0xF, 0xF, true);   // wave_rol:1
    out[threadIdx.x] = r; out[64+threadIdx.x] = s; out[128+threadIdx.x]=l;
}
int main(){ int *d; hipMalloc(&d, 192*4); k<<<1,64>>>(d); int h[192]; hipMemcpy(h,d,192*4,hipMemcpyDeviceToHost);
 printf("ror: %d %d %d ... %d | shr: %d %d ... %d | rol: %d %d ... %d\n", h[0],h[1],h[2],h[63],h[64],h[65],h[127],h[128],h[129],h[191]); return 0; }
