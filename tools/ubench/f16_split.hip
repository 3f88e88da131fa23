// f16_split.hip -- facts the fp16-piece recompute of the backward kernels rests on (gfx950):
//   1. does v_mfma_f32_16x16x32_f16 keep SUBNORMAL f16 inputs (the low piece of a value below 2^-3 is subnormal)?
//   2. does the two-piece split  hi = f16(a), lo = f16(a - hi)  (v_cvt_pk_f16_f32 + v_fma_mix_f32) round to nearest and
//      produce subnormal low pieces, i.e. |a - hi - lo| <= max(2^-22 |a|, 2^-25)?
//   3. what the three-term product  hi*hi + hi*lo + lo*hi  loses against the exact fp32 product.
// hipcc --offload-arch=gfx950 -O3 f16_split.hip -o f16_split && ./f16_split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifdef PLAIN_SPLIT
__device__ __forceinline__ void split2(float a0, float a1, h16x2& hi, h16x2& lo) {
    hi = __builtin_convertvector(f32x2{a0, a1}, h16x2);
    const float r0 = a0 - (float)hi[0];
    const float r1 = a1 - (float)hi[1];
    lo = __builtin_convertvector(f32x2{r0, r1}, h16x2);
}
#else
// four instructions per pair: the remainder a - hi straight off the packed halves (v_fma_mix_f32 reads an f16 half as an operand)
__device__ __forceinline__ void split2(float a0, float a1, h16x2& hi, h16x2& lo) {
    hi = __builtin_convertvector(f32x2{a0, a1}, h16x2);
    const unsigned hiu = __builtin_bit_cast(unsigned, hi);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hiu), "v"(a0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hiu), "v"(a1));
    lo = __builtin_convertvector(f32x2{r0, r1}, h16x2);
}
#endif

// D[i][j] = sum_k A[i][k] B[k][j]; lane (g = lane >> 4, p = lane & 15): A row p, k = 8 g .. 8 g + 7; B column p, same k
__global__ void mfma_subnormal(float* out, float a_val, float b_val) {
    const int lane = threadIdx.x;
    h16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)a_val; b[j] = (_Float16)b_val; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    out[lane] = c[0];
}

__global__ void split_check(const float* in, float* res, float* hi_out, float* lo_out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    h16x2 hi, lo;
    split2(in[2 * i], in[2 * i + 1], hi, lo);
    for (int j = 0; j < 2; ++j) {
        res[2 * i + j] = (in[2 * i + j] - (float)hi[j]) - (float)lo[j];
        hi_out[2 * i + j] = (float)hi[j];
        lo_out[2 * i + j] = (float)lo[j];
    }
}

// one 16 x 16 x 32 product with three terms on random data against the fp64 product of the fp32 inputs
__global__ void three_term(const float* A, const float* B, float* D) {      // A [16][32], B [32][16] row-major
    const int lane = threadIdx.x, g = lane >> 4, p = lane & 15;
    h16x8 ah, al, bh, bl;
    for (int j = 0; j < 8; j += 2) {
        h16x2 h, l;
        split2(A[p * 32 + 8 * g + j], A[p * 32 + 8 * g + j + 1], h, l);
        ah[j] = h[0]; ah[j + 1] = h[1]; al[j] = l[0]; al[j + 1] = l[1];
        split2(B[(8 * g + j) * 16 + p], B[(8 * g + j + 1) * 16 + p], h, l);
        bh[j] = h[0]; bh[j + 1] = h[1]; bl[j] = l[0]; bl[j + 1] = l[1];
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + p] = c[r];           // D row 4 g + r, column p
}

int main() {
    float* d;
    hipMalloc(&d, 64 * sizeof(float));
    float h[64];
    struct { float a, b; const char* what; } cases[] = {
        {5.9604645e-8f, 1024.f, "a = 2^-24 (smallest f16 subnormal), b = 2^10: exact sum 32 * 2^-14 = 0.001953125"},
        {3.0517578e-5f, 1.f, "a = 2^-15 (subnormal), b = 1: exact 32 * 2^-15 = 0.0009765625"},
        {6.1035156e-5f, 1.f, "a = 2^-14 (smallest normal), b = 1: exact 0.001953125"},
        {1.f, 5.9604645e-8f, "a = 1, b = 2^-24 (subnormal B operand): exact 1.9073486e-06"},
    };
    for (auto& cs : cases) {
        mfma_subnormal<<<1, 64>>>(d, cs.a, cs.b);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mfma f16 subnormal input: %-90s -> %.9g\n", cs.what, h[0]);
    }
    // split accuracy over magnitudes
    const int n = 1 << 16;
    std::vector<float> in(n), res(n), hi(n), lo(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const float mag = ldexpf(1.f, (rand() % 36) - 24);                // 2^-24 .. 2^11
        in[i] = mag * (1.f + (float)rand() / RAND_MAX) * ((rand() & 1) ? 1.f : -1.f);
    }
    float *din, *dres, *dhi, *dlo;
    hipMalloc(&din, n * 4); hipMalloc(&dres, n * 4); hipMalloc(&dhi, n * 4); hipMalloc(&dlo, n * 4);
    hipMemcpy(din, in.data(), n * 4, hipMemcpyHostToDevice);
    split_check<<<n / 2 / 256, 256>>>(din, dres, dhi, dlo, n);
    hipMemcpy(res.data(), dres, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hi.data(), dhi, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(lo.data(), dlo, n * 4, hipMemcpyDeviceToHost);
    double worst_rel = 0, worst_abs = 0, worst_viol = 0;
    int nsub = 0;
    for (int i = 0; i < n; ++i) {
        const double a = fabs(in[i]), e = fabs(res[i]);
        const double bound = fmax(ldexp(a, -22), ldexp(1.0, -25));
        worst_viol = fmax(worst_viol, e / bound);
        if (a >= 0.125) worst_rel = fmax(worst_rel, e / a); else worst_abs = fmax(worst_abs, e);
        if (lo[i] != 0.f && fabs(lo[i]) < 6.1035156e-5) ++nsub;
    }
    printf("split: worst |a - hi - lo| / |a| for |a| >= 2^-3: %.3g (2^-22 = %.3g); worst abs error below: %.3g (2^-25 = %.3g); "
           "worst error / bound %.3f; subnormal low pieces seen: %d\n", worst_rel, ldexp(1.0, -22), worst_abs, ldexp(1.0, -25), worst_viol, nsub);
    // three-term product
    std::vector<float> A(512), B(512), D(256);
    for (auto& v : A) v = 2.f * rand() / RAND_MAX - 1.f;
    for (auto& v : B) v = 2.f * rand() / RAND_MAX - 1.f;
    float *dA, *dB, *dD;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    three_term<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    double worst = 0, worst32 = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double ref = 0, mag = 0;
            float f32 = 0.f;
            for (int k = 0; k < 32; ++k) { ref += (double)A[i * 32 + k] * B[k * 16 + j]; mag += fabs((double)A[i * 32 + k] * B[k * 16 + j]); f32 = fmaf(A[i * 32 + k], B[k * 16 + j], f32); }
            worst = fmax(worst, fabs(D[i * 16 + j] - ref) / mag);
            worst32 = fmax(worst32, fabs(f32 - ref) / mag);
        }
    printf("three-term f16 product, K = 32, uniform(-1, 1): worst |err| / sum|terms| = %.3g (an fp32 fma chain: %.3g)\n", worst, worst32);
    return 0;
}
