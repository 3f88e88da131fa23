// Microbenchmark: fill.hip with a REALISTIC filler -- the activation/split chain of the quadrature kernels
// (mul, max, mul, max, cvt_pk, lshl, and, sub, sub, cvt_pk: dependent, on rotating registers) spread behind MFMAs that
// read rotating A/B operand registers.  MODE 0: MFMAs only; 1: chains only; 2: one chain per G MFMAs, interleaved
// MFMA / chain slice; 3: same instructions, all MFMAs of a group first, then the chain.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define NIT 4000

template <int MODE, int G>
__global__ __launch_bounds__(512) void k(float* out) {
    f32x4 acc[4];
    u32x4 A[8], B[4];
    float z[16];
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0, 0, 0, 0};
    for (int j = 0; j < 8; ++j) A[j] = u32x4{threadIdx.x + j, 2, 3, 4};
    for (int j = 0; j < 4; ++j) B[j] = u32x4{5, 6, 7, threadIdx.x + j};
    for (int i = 0; i < 16; ++i) z[i] = threadIdx.x * 1e-3f + i;
    float slope = 0.01f;
    unsigned msk = 0xffff0000u;
    for (int it = 0; it < NIT; ++it) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {           // 8 chains per iteration, each behind G MFMAs
            float &x0 = z[2 * c], &x1 = z[2 * c + 1];
            float t0, t1, h, r0, r1;
            if (MODE == 3 || MODE == 0) {
#pragma unroll
                for (int q = 0; q < G; ++q)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(c * G + q) & 3]) : "v"(A[(c * G + q) & 7]), "v"(B[(c + q) & 3]));
            }
            // chain slices: 10 VALU
#define MF(q) if (MODE == 2 && (q) < G) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(c * G + (q)) & 3]) : "v"(A[(c * G + (q)) & 7]), "v"(B[(c + (q)) & 3]));
            if (MODE != 0) {
                MF(0)
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(slope), "v"(x0));
                asm volatile("v_max_f32 %0, %1, %0" : "+v"(t0) : "v"(x0));
                if (G >= 4) { MF(1) }
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(slope), "v"(x1));
                asm volatile("v_max_f32 %0, %1, %0" : "+v"(t1) : "v"(x1));
                if (G >= 4) { MF(2) } else { MF(1) }
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(t0), "v"(t1));
                asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(r0) : "v"(h));
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(r1) : "v"(msk), "v"(h));
                if (G >= 4) { MF(3) } else { MF(2) }
                asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r0) : "v"(t0));
                asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r1) : "v"(t1));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x0) : "v"(r0), "v"(r1));
                if (G >= 5) { MF(4) }
                x1 = h;
            } else if (MODE == 2) {}
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j][0];
    for (int i = 0; i < 16; ++i) s += z[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static double g_cyc = 0;
template <int MODE, int G>
void run(float* out) {
    static const char* names[] = {"MFMA only", "chains only", "interleaved", "MFMAs then chain"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double res[2];
    for (int wv = 0; wv < 2; ++wv) {
        const int threads = wv ? 512 : 256;
        hipLaunchKernelGGL((k<MODE, G>), dim3(256), dim3(threads), 0, 0, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, G>), dim3(256), dim3(threads), 0, 0, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        res[wv] = ms * 1e6 / (NIT * 8.0) / (wv ? 2 : 1);     // ns per (G MFMAs + 1 chain) per SIMD
    }
    if (MODE == 0 && g_cyc == 0) g_cyc = res[0] / (16.0 * G);
    printf("G=%d %-18s: %7.1f cycles per group (1 wave/SIMD) %7.1f (2 waves/SIMD)   [MFMA pipe %d]\n", G, names[MODE], res[0] / g_cyc, res[1] / g_cyc, MODE == 1 ? 0 : 16 * G);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    run<0, 4>(out); run<1, 4>(out); run<2, 4>(out); run<3, 4>(out);
    run<0, 3>(out); run<2, 3>(out); run<3, 3>(out);
    run<0, 5>(out); run<2, 5>(out); run<3, 5>(out);
    return 0;
}
