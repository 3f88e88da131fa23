// Microbenchmark: how many independent VALU instructions hide in the shadow of one bf16 MFMA, per MFMA shape.
//   hipcc --offload-arch=gfx950 -O3 -w fill.hip -o fill && ./fill
// One wave issues NIT x 4 groups of { one MFMA (rotating over 4 accumulators), F filler VALU instructions on registers
// the MFMA does not touch }, program order forced with volatile inline asm.  Reported: cycles per group, calibrated on
// the F = 0 stream of v_mfma_f32_16x16x32_bf16 = 16 cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define NIT 10000

template <int BIG, int F, int KIND>
__global__ __launch_bounds__(512) void k(float* out) {
    f32x4 a4[4]; f32x16 a16[4];
    for (int j = 0; j < 4; ++j) { a4[j] = f32x4{0, 0, 0, 0}; for (int i = 0; i < 16; ++i) a16[j][i] = 0.f; }
    u32x4 a = {threadIdx.x, 2, 3, 4}, b = {5, 6, 7, threadIdx.x};
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    float c1 = 1.0001f, c2 = 0.5f;
    for (int it = 0; it < NIT; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (BIG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a16[j]) : "v"(a), "v"(b));
            else     asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a4[j]) : "v"(a), "v"(b));
#pragma unroll
            for (int q = 0; q < F; ++q) {
                float& x = v[(j * F + q) & 7];
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
                if (KIND == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(c2));
                if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(c1));
                if (KIND == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c1));
            }
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) { s += a4[j][0]; for (int i = 0; i < 16; ++i) s += a16[j][i]; }
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double g_ns_per_cycle = 0;
template <int BIG, int F, int KIND>
void run(float* out) {
    static const char* kinds[] = {"v_fma_f32", "v_max_f32", "v_cvt_pk_bf16_f32", "v_mul_f32"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double res[2];
    for (int wv = 0; wv < 2; ++wv) {
        const int threads = wv ? 512 : 256;
        hipLaunchKernelGGL((k<BIG, F, KIND>), dim3(256), dim3(threads), 0, 0, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<BIG, F, KIND>), dim3(256), dim3(threads), 0, 0, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        res[wv] = ms * 1e6 / (NIT * 4.0) / (wv ? 2 : 1);       // ns per group per SIMD
    }
    if (!BIG && F == 0 && g_ns_per_cycle == 0) g_ns_per_cycle = res[0] / 16.0;
    printf("%s + %d x %-18s : %6.1f cycles/group (1 wave/SIMD)  %6.1f (2 waves/SIMD)\n", BIG ? "32x32x16" : "16x16x32", F, kinds[KIND],
           res[0] / g_ns_per_cycle, res[1] / g_ns_per_cycle);
}
template <int BIG, int KIND> void sweep(float* out) {
    run<BIG, 0, KIND>(out); run<BIG, 1, KIND>(out); run<BIG, 2, KIND>(out); run<BIG, 3, KIND>(out); run<BIG, 4, KIND>(out);
    run<BIG, 5, KIND>(out); run<BIG, 6, KIND>(out); run<BIG, 8, KIND>(out); run<BIG, 12, KIND>(out);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    sweep<0, 0>(out); sweep<1, 0>(out); sweep<0, 1>(out); sweep<1, 1>(out); sweep<0, 2>(out); sweep<1, 2>(out); sweep<0, 3>(out); sweep<1, 3>(out);
    return 0;
}
