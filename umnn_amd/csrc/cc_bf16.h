// bf16-split helpers shared by the forward and backward bf16 kernels: operand types, the 16x16x32 MFMA wrapper,
// round-to-nearest splitting of fp32 values into bf16 pieces.
#pragma once
#include "cc_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short bf16_rn_bits(float x) { return f32_to_bf16_rn(x); }
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// K = 16 variant (4 bf16 per lane): same 16 cycles as the K = 32 instruction, for products whose contraction is only
// 16 long (the backward's dW = delta x a over the 16 points of a tile) -- no zero-padded operand halves to build.
__device__ __forceinline__ f32x4 mfma_bf16_k16(u32x2 a, u32x2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}

// (x0,x1) -> packed bf16 pairs of the NPARTS pieces (piece k of x0 in the low half of out[k], of x1 in the high half)
template <int NPARTS>
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned (&out)[NPARTS]) {
#pragma unroll
    for (int k = 0; k < NPARTS; ++k) {
        const bf16x2 h = __builtin_convertvector(f32x2{x0, x1}, bf16x2);      // v_cvt_pk_bf16_f32 (round to nearest even)
        const unsigned bits = __builtin_bit_cast(unsigned, h);
        out[k] = bits;
        if (k + 1 < NPARTS) {
            x0 -= __uint_as_float(bits << 16);
            x1 -= __uint_as_float(bits & 0xffff0000u);
        }
    }
}

