// Probe: operand / result layout of v_mfma_f32_32x32x16_bf16 and v_mfma_f32_32x32x2_f32 on gfx950.
//   hipcc --offload-arch=gfx950 -O2 mfma32.hip -o mfma32 && ./mfma32
// A[m][k] = 1 at a single (m0, k0), B[k][n] = 1 at a single (k0, n0) -> D has a single 1 at (m0, n0); the lane / register that
// holds it, and the lane / slot that had to supply the operands, are printed for a few positions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe_bf16(int la, int ja, int lb, int jb, float* out) {
    const int lane = threadIdx.x;
    unsigned short av[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane == la) av[ja] = 0x3f80;        // bf16 1.0
    if (lane == lb) bv[jb] = 0x3f80;
    u32x4 a, b;
    for (int e = 0; e < 4; ++e) {
        a[e] = (unsigned)av[2 * e] | ((unsigned)av[2 * e + 1] << 16);
        b[e] = (unsigned)bv[2 * e] | ((unsigned)bv[2 * e + 1] << 16);
    }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}
__global__ void probe_f32(int la, int lb, float* out) {
    const int lane = threadIdx.x;
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(lane == la ? 1.f : 0.f, lane == lb ? 1.f : 0.f, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}
int main() {
    float* d; hipMalloc(&d, 64 * 16 * 4);
    float h[64 * 16];
    printf("bf16 32x32x16: A lane la slot ja, B lane lb slot jb -> nonzero D entries (lane, reg)\n");
    int cases[][4] = {{0, 0, 0, 0}, {5, 0, 7, 0}, {5, 3, 7, 3}, {5, 3, 7, 2}, {37, 1, 7, 1}, {37, 1, 39, 1}, {5, 0, 39, 0}, {31, 7, 63, 7}, {63, 7, 63, 7}};
    for (auto& cs : cases) {
        probe_bf16<<<1, 64>>>(cs[0], cs[1], cs[2], cs[3], d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("A(l=%d,j=%d) B(l=%d,j=%d):", cs[0], cs[1], cs[2], cs[3]);
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) if (h[l * 16 + i] != 0.f) printf(" (lane %d, reg %d)=%g", l, i, h[l * 16 + i]);
        printf("\n");
    }
    printf("f32 32x32x2: A lane la, B lane lb\n");
    int c2[][2] = {{0, 0}, {5, 7}, {37, 39}, {5, 39}, {37, 7}, {13, 40}, {45, 40}};
    for (auto& cs : c2) {
        probe_f32<<<1, 64>>>(cs[0], cs[1], d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("A(l=%d) B(l=%d):", cs[0], cs[1]);
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) if (h[l * 16 + i] != 0.f) printf(" (lane %d, reg %d)=%g", l, i, h[l * 16 + i]);
        printf("\n");
    }
    return 0;
}
