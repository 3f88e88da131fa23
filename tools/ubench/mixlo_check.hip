// v_fma_mixlo_f16 / v_fma_mixhi_f16 as the second fp16 piece of a pair (cc_fwd_bf16_kernel.h pc_lo_pair) against the reference
// formulation cvt_pk(x - float(hi)): bit-for-bit over random values of many binades, zeros, negative zero, subnormal remainders.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ unsigned cvt_pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, h2)); }
__global__ void k(const float* x, unsigned* ref, unsigned* got, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x0 = x[2 * i], x1 = x[2 * i + 1];
    unsigned hi = cvt_pk(x0, x1);
    h2 h = __builtin_bit_cast(h2, hi);
    ref[i] = cvt_pk(x0 - (float)h[0], x1 - (float)h[1]);
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(x1));
    got[i] = lo;
}
int main() {
    const int n = 1 << 20;
    float* hx = (float*)malloc(8 * n);
    srand(1);
    for (int i = 0; i < 2 * n; ++i) {
        float m = (float)rand() / RAND_MAX * 2 - 1;
        int e = rand() % 40 - 28;
        hx[i] = ldexpf(m, e);
        if (i % 97 == 0) hx[i] = 0.f;
        if (i % 101 == 0) hx[i] = -0.f;
    }
    float* dx; unsigned *dr, *dg;
    hipMalloc(&dx, 8 * n); hipMalloc(&dr, 4 * n); hipMalloc(&dg, 4 * n);
    hipMemcpy(dx, hx, 8 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dr, dg, n);
    unsigned *hr = (unsigned*)malloc(4 * n), *hg = (unsigned*)malloc(4 * n);
    hipMemcpy(hr, dr, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(hg, dg, 4 * n, hipMemcpyDeviceToHost);
    int bad = 0, signz = 0;
    for (int i = 0; i < n; ++i) if (hr[i] != hg[i]) {
        // a +-0 difference in a half is harmless for a sum of pieces; count it separately
        unsigned d = hr[i] ^ hg[i];
        if ((d & 0x7fff7fffu) == 0 && ((hr[i] & 0x7fffu) == 0 || (d & 0xffffu) == 0) && ((hr[i] >> 16 & 0x7fffu) == 0 || (d >> 16) == 0)) { ++signz; continue; }
        if (bad++ < 10) printf("x = (%g, %g): ref %08x got %08x\n", hx[2 * i], hx[2 * i + 1], hr[i], hg[i]);
    }
    printf("pairs %d  mismatches %d  (+-0 only: %d)\n", n, bad, signz);
    return bad != 0;
}
