// Microbenchmark: issue cost of the VALU instructions the quadrature kernels are made of, one wave per SIMD and two.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
// Each wave runs NIT iterations of 16 independent instructions of one kind (inline asm, 16 disjoint destination
// registers / register pairs).  Cycles are calibrated on v_mfma_f32_16x16x32_bf16 = 16 cycles (4 passes).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define NIT 20000

template <int OP>
__global__ __launch_bounds__(512) void k(float* out) {
    float v[16];
    f32x2 w[16];
    f32x4 acc[16];
    u32x4 a = {threadIdx.x, 2, 3, 4}, b = {5, 6, 7, threadIdx.x};
    for (int i = 0; i < 16; ++i) { v[i] = threadIdx.x * 1e-3f + i; w[i] = f32x2{v[i], v[i] + 0.5f}; acc[i] = f32x4{0, 0, 0, 0}; }
    float c1 = 1.0001f, c2 = 0.5f;
    f32x2 p1 = {1.0001f, 0.9999f};
    unsigned sel = 0x07060302u;
    unsigned long long msk = 0x5555555555555555ull ^ blockIdx.x;
    float sc = 1.0001f + blockIdx.x * 0.f;
    for (int it = 0; it < NIT; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c1), "v"(c2));
            if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c1));
            if (OP == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c2));
            if (OP == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[j]) : "v"(p1));
            if (OP == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w[j]) : "v"(p1));
            if (OP == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[j]) : "v"(p1));
            if (OP == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c1));
            if (OP == 7) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(v[j]));
            if (OP == 8) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[j]) : "v"(sel));
            if (OP == 9) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v[j]) : "v"(c1), "v"(c2));
            if (OP == 10) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c1), "v"(sel));
            if (OP == 11) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
            if (OP == 12) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[j], 0, 0, 0);
            if (OP == 13) asm volatile("v_mov_b32 %0, %1" : "+v"(v[j]) : "v"(c1));
            if (OP == 14) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[j]) : "v"(c1), "v"(c2));
            if (OP == 15) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(w[j]) : "v"(p1));
            if (OP == 16) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(c1));
            if (OP == 17) asm volatile("v_max_f32 %0, %0, %0" : "+v"(v[j]));
            if (OP == 18) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c1));
            if (OP == 19) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c1), "s"(msk));
            if (OP == 20) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[j]) : "s"(sc));
            if (OP == 21) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c1));
            if (OP == 22) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(v[j]));
            if (OP == 23) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[j]) : "v"(c1));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i] + w[i][0] + w[i][1] + acc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static const char* kNames[] = {"v_fma_f32", "v_mul_f32", "v_max_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_cvt_pk_bf16_f32",
                               "v_lshlrev_b32", "v_and_b32", "v_dot2c_f32_bf16", "v_perm_b32", "v_exp_f32", "v_mfma_f32_16x16x32_bf16",
                               "v_mov_b32", "v_fmac_f32", "v_pk_mul_f32 op_sel", "v_cndmask_b32", "v_max_f32 x,x", "v_add_f32", "v_cndmask_b32 sgpr", "v_mul_f32 sgpr", "v_sub_f32", "v_and_b32 literal", "v_fma_f32 2src"};
static double g_ref[2];
template <int OP>
void run(float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wv = 0; wv < 2; ++wv) {
        const int threads = wv ? 512 : 256;
        hipLaunchKernelGGL((k<OP>), dim3(256), dim3(threads), 0, 0, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<OP>), dim3(256), dim3(threads), 0, 0, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ns = ms * 1e6 / (NIT * 16.0) / (wv ? 2 : 1);     // per instruction per SIMD
        if (OP == 12) g_ref[wv] = ns / 16.0;                          // ns per cycle
        printf("%-28s %d wave/SIMD: %7.3f ns/instr = %5.2f cycles\n", kNames[OP], wv + 1, ns, g_ref[wv] > 0 ? ns / g_ref[wv] : 0.0);
    }
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    run<12>(out); run<0>(out); run<1>(out); run<2>(out); run<3>(out); run<4>(out); run<5>(out); run<6>(out); run<7>(out); run<8>(out);
    run<9>(out); run<10>(out); run<11>(out); run<13>(out); run<14>(out); run<15>(out); run<16>(out); run<17>(out); run<18>(out); run<19>(out); run<20>(out); run<21>(out); run<22>(out); run<23>(out);
    printf("--- second pass ---\n"); run<12>(out); run<0>(out); run<1>(out); run<2>(out); run<8>(out); run<13>(out); run<18>(out); run<3>(out); run<4>(out);
    return 0;
}
