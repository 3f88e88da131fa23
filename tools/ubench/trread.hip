// Probe: what does ds_read_b64_tr_b16 return?  LDS is filled with its own index; every lane passes the address of 4
// contiguous bf16 (8 bytes); the 4 returned 16-bit values per lane are printed.
//   hipcc --offload-arch=gfx950 -O3 -w trread.hip -o trread && ./trread
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* o, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // mode 0: lane l points at row l of a [64][4] matrix;  mode 1: 16-lane group g points at [4 x 16] block g, lane i at
    // 4 contiguous elements starting at (i>>2)*16 + (i&3)*4 of that block
    const int off = mode == 0 ? l * 4 : (l >> 4) * 64 + ((l & 15) >> 2) * 16 + (l & 3) * 4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(sm + off));
    for (int j = 0; j < 4; ++j) o[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : "   ");
    }
    return 0;
}
