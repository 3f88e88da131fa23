// Microbenchmark: can bf16 / fp32 MFMA overlap with VALU work (a) from another wave on the same SIMD,
// (b) interleaved inside one wave?   hipcc --offload-arch=gfx950 -O3 overlap.hip -o overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define NIT 4000
// mode: 0 = MFMA only, 1 = VALU only, 2 = even waves MFMA / odd waves VALU, 3 = interleaved in one wave (1 MFMA : K VALU)
template <int BF16, int KV>
__global__ __launch_bounds__(512) void k(int mode, float* out, int nw) {
    const int wid = threadIdx.x >> 6;
    f32x4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    u32x4 a = {threadIdx.x, 2, 3, 4}, b = {5, 6, 7, threadIdx.x};
    float fa = threadIdx.x * 1e-3f, fb = 1.0001f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    const bool do_m = mode == 0 || mode == 3 || (mode == 2 && (wid & 4) == 0);   // waves 0-3 vs 4-7 (same SIMDs)
    const bool do_v = mode == 1 || mode == 3 || (mode == 2 && (wid & 4) != 0);
    if (mode == 3) {
        for (int it = 0; it < NIT; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (BF16) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[j], 0, 0, 0);
                else acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[j], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < KV; ++q) v[(j * KV + q) & 7] = fmaf(v[(j * KV + q) & 7], 1.0001f, 0.5f);
            }
        }
    } else {
        if (do_m)
            for (int it = 0; it < NIT; ++it) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (BF16) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[j], 0, 0, 0);
                    else acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[j], 0, 0, 0);
                }
            }
        if (do_v)
            for (int it = 0; it < NIT; ++it) {
#pragma unroll
                for (int j = 0; j < 4 * KV; ++j) v[j & 7] = fmaf(v[j & 7], 1.0001f, 0.5f);
            }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int BF16, int KV>
void run(const char* name, int threads) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        if (mode == 2 && threads != 512) continue;
        hipLaunchKernelGGL((k<BF16, KV>), dim3(256), dim3(threads), 0, 0, mode, out, threads / 64);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<BF16, KV>), dim3(256), dim3(threads), 0, 0, mode, out, threads / 64);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * 2.4e9;   // nominal
        printf("%s KV=%d threads=%d mode=%d: %.3f ms  = %.1f cycles per (MFMA + %d VALU) group @2.4GHz\n", name, KV, threads, mode, ms, cyc / (NIT * 4.0), KV);
    }
    hipFree(out);
}

int main() {
    run<1, 3>("bf16", 256); run<1, 3>("bf16", 512);
    run<1, 6>("bf16", 256); run<1, 6>("bf16", 512);
    run<0, 3>("fp32", 256); run<0, 3>("fp32", 512);
    run<0, 6>("fp32", 256); run<0, 6>("fp32", 512);
    return 0;
}
