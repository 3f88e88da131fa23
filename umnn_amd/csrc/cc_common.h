// Shared device/host definitions for the gfx950 Clenshaw-Curtis kernels.
//
// Register / LDS conventions used by every kernel in this directory
// ------------------------------------------------------------------
// All hidden-layer GEMMs run on v_mfma_f32_16x16x4_f32 (exact fp32, 32 cycles per SIMD) in the
// orientation  D[feature][point] += A[feature][k] * B[k][point]:
//   * a wave owns P "point tiles" of 16 integrals each; lane = 16*g + p  (g = lane>>4, p = lane&15)
//     holds point p of the tile in every MFMA B operand and in every accumulator;
//   * an activation vector of one hidden layer lives in registers act[tile t][r] (f32x4 per tile):
//     lane (g,p), tile t, component r  <->  feature f = 16*t + 4*r + g  of point p.
//     With that numbering the accumulator of output tile t *is already* the B operand of K-step
//     s = 4*t + r of the next layer (features 4*s .. 4*s+3 on lane groups g = 0..3): layers chain
//     through registers with no cross-lane movement and no LDS round trip;
//   * the A operand of (output tile t, K-step s) is 64 floats, lane l -> W[f_out][f_in] with
//     f_out = 16*t + 4*(rho&3) + (rho>>2), rho = l&15  and  f_in = 4*s + (l>>4); it is staged once
//     per workgroup in LDS as img[(t*KS + s)*64 + l] and read with a conflict-free ds_read_b32;
//   * every hidden layer carries one extra constant-one feature (index H_l): the bias of the next
//     layer is the weight column of that feature, and the row H_{l+1} of the next image is
//     e_{H_l}, so the constant propagates.  Accumulators therefore start at zero and no bias
//     registers are needed.  Padding features beyond H_l+1 are exact zeros.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/umnn_cc.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define UMNN_WAVES_PER_BLOCK 4
#define UMNN_BLOCK (64 * UMNN_WAVES_PER_BLOCK)

struct MlpDev {
    const float* W[UMNN_MAX_LINEAR];
    const float* b[UMNN_MAX_LINEAR];
    int width[UMNN_MAX_LINEAR + 1];   // [1+E, H1..HL, 1]
    int n_linear;
    int t_out[UMNN_MAX_LINEAR];       // t_out[l]  = ceil((H_l+1)/16): tiles of hidden layer l (l = 1..L)
    int ks_in[UMNN_MAX_LINEAR];       // ks_in[l]  = ceil((H_l+1)/4):  K-steps when layer l is the input
    int lds_off[UMNN_MAX_LINEAR];     // float offset of image l (hidden l -> hidden l+1), l = 1..L-1
    int t_mfma[UMNN_MAX_LINEAR];      // tiles of hidden layer l produced by MFMA (= t_out[l], or t_out[l]-1 in TAIL mode)
    int tail_off;                     // TAIL mode: float offset of the tail-row weights [l-1][j][g][16] in LDS
    int n_tail;                       // TAIL mode: real features in the last tile (computed on the VALU), 0..3
    int hidden_act;
    int out_act;
};

// feature index held by lane group g in component r of tile t
__device__ __forceinline__ int feat_of(int t, int r, int g) { return 16 * t + 4 * r + g; }
// output feature of accumulator row rho (= lane&15 in an A operand) of tile t
__device__ __forceinline__ int fout_of(int t, int rho) { return 16 * t + 4 * (rho & 3) + (rho >> 2); }

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- fp32 / bf16 tensor I/O (configuration C4: bf16 activations around fp32 arithmetic) -----------------------------
// The kernels compute in fp32 whatever the storage type; `bf16` flags say how x-class tensors (x, x0, g, F, f, z, log_jac,
// dx) and h-class tensors (h, dh) are stored.  bf16 loads are exact widenings; stores round to nearest even.
__device__ __forceinline__ unsigned short f32_to_bf16_rn(float x) {
    const unsigned u = __float_as_uint(x);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float io_ld(const float* p, long long i, int bf16) {
    return bf16 ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(p)[i] << 16) : p[i];
}
__device__ __forceinline__ void io_st(float* p, long long i, float v, int bf16) {
    if (bf16) reinterpret_cast<unsigned short*>(p)[i] = f32_to_bf16_rn(v);
    else p[i] = v;
}
struct IoView {          // read-only view of an fp32 or bf16 array: view[i], view + offset
    const float* p;
    int bf16;
    __device__ __forceinline__ float operator[](long long i) const { return io_ld(p, i, bf16); }
    __device__ __forceinline__ IoView operator+(long long o) const {
        return IoView{bf16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(p) + o) : p + o, bf16};
    }
};

// c += W_1[:, 1:] h for one tile of 16 integrals (c = the per-tile constants of layer 1, feature-major accumulators of the fp32
// MFMA; lane (g, p): K = embedding columns 4 se + g, output rows fout_of(t, p)).  CH K-steps per trip, ALL their loads (h from HBM,
// W_1 from L2) issued before the first product: opening a tile costs one memory latency per trip, not one per K-step -- a wave
// that opens a tile has nothing else to issue meanwhile (and in the workgroup pipeline seven other waves wait for it).
template <int NT, int CH>
__device__ __forceinline__ void item_embedding_gemm(const IoView& hb, const float* __restrict__ W0, int H1, int E, int d, int g, int p,
                                                    f32x4 (&c)[NT]) {
    for (int se0 = 0; se0 < (E + 3) / 4; se0 += CH) {
        float hv[CH], A[CH][NT];
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int e = 4 * (se0 + q) + g;
            hv[q] = e < E ? hb[(long long)e * d] : 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int fo = fout_of(t, p);
                A[q][t] = (fo < H1 && e < E) ? W0[fo * (1 + E) + 1 + e] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < CH; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t) c[t] = mfma16(A[q][t], hv[q], c[t]);
    }
}

// v_max_f32 without the canonicalising v_max(v,v) hipcc puts in front of fmaxf on MFMA results (fmaxf must quiet
// signalling NaNs; the hardware instruction on already-finite data does not need it).  One VALU op instead of two.
// Inline assembly and MFMA hazards: the compiler's hazard recogniser does not look inside `asm`, and a free-standing "=v" output may be
// given a register that an in-flight MFMA still reads as its C operand when the accumulators live in VGPRs and the compiler has renamed
// one (dst != srcC) -- round 5 met exactly that with another instruction (EXPERIMENTS.md).  A hazard of this kind is a property of
// the BINARY (straight-line distance between two instructions), so the bit-exactness and parity tests of a build either see it or it is
// not there.  UMNN_ASM_TIED (the forward translation units: built with -amdgpu-mfma-vgpr-form, re-scheduled by hand every round): the
// output is tied to the first input, whose last writer is a compiler-visible instruction -- no such register can be handed out.  The
// backward translation units keep the untied form: tied, every activation that stays live costs a v_mov (measured: C3 backward
// 10.30 -> 10.42 ms, MNIST-shaped stages 1.77 -> 1.81 ms).
__device__ __forceinline__ float vmax_f32(float a, float b) {
#ifdef UMNN_ASM_TIED
    asm("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    return a;
#else
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}
// hidden activation: LeakyReLU(0.01) (slope = 0.01) or ReLU (slope = 0) as max(v, slope*v)
__device__ __forceinline__ float hidden_act_f(float v, float slope) { return vmax_f32(v, slope * v); }
__device__ __forceinline__ float hidden_grad_f(float v, float slope) { return v > 0.f ? 1.f : slope; }

__device__ __forceinline__ float out_act_f(float v, int kind) {
    if (kind == UMNN_OUT_ELU_PLUS_ONE) return v > 0.f ? v + 1.f : __expf(v);   // ELU(v)+1 = exp(v) for v<=0
    return 1.f / (1.f + __expf(-v));
}
__device__ __forceinline__ float out_grad_f(float v, int kind) {
    if (kind == UMNN_OUT_ELU_PLUS_ONE) return v > 0.f ? 1.f : __expf(v);
    const float s = 1.f / (1.f + __expf(-v));
    return s * (1.f - s);
}

// f or 1/f (inv_f, ParallelNeuralIntegral.py:58-59).  inv_f is a launch argument and almost never set, but as a select both
// sides are evaluated for every node: the hardware reciprocal (v_rcp_f32, 1 ulp) costs one instruction there, the IEEE
// division sequence eleven -- 22 of the ~450 vector instructions of a two-tile node of the forward kernel.  (A real branch
// would split the hand-scheduled node body into several scheduling regions.)
__device__ __forceinline__ float maybe_inverse(float f, int inv_f) { return inv_f ? __builtin_amdgcn_rcpf(f) : f; }

// sum over the four lane groups (lanes p, p+16, p+32, p+48); every lane gets the total.
// gfx950 row/half swaps keep this on the VALU (no LDS-crossbar bpermute): permlane16_swap(v,v) returns
// {rows (0,0,2,2), rows (1,1,3,3)} of v, permlane32_swap(v,v) returns {(lo,lo), (hi,hi)}.
__device__ __forceinline__ float group_allreduce(float v) {
    const unsigned u = __float_as_uint(v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const unsigned w = __float_as_uint(s);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Stage the hidden->hidden weight images (see header comment) into LDS.  Called by all threads
// of the workgroup; caller issues the __syncthreads().
__device__ __forceinline__ void stage_hidden_images(const MlpDev& m, float* lds, int tid, int nthreads) {
    const int L = m.n_linear - 1;
    for (int l = 1; l < L; ++l) {
        const int Hin = m.width[l], Hout = m.width[l + 1];
        const int ks = m.ks_in[l], to = m.t_mfma[l + 1];
        const float* __restrict__ W = m.W[l];
        const float* __restrict__ b = m.b[l];
        float* img = lds + m.lds_off[l];
        const int total = to * ks * 64;
        for (int idx = tid; idx < total; idx += nthreads) {
            const int ln = idx & 63, ts = idx >> 6;
            const int s = ts % ks, t = ts / ks;
            const int fo = fout_of(t, ln & 15), fi = 4 * s + (ln >> 4);
            float v = 0.f;
            if (fo < Hout) {
                if (fi < Hin) v = W[fo * Hin + fi];
                else if (fi == Hin) v = b[fo];
            } else if (fo == Hout && fi == Hin) {
                v = 1.f;
            }
            img[idx] = v;
        }
    }
    // TAIL mode: rows of the last tile's real features, stored [l-1][j][g][16] so a lane reads its 4t..4t+3 as b128
    if (m.n_tail > 0) {
        for (int l = 1; l < L; ++l) {
            const int Hin = m.width[l], Hout = m.width[l + 1];
            const float* __restrict__ W = m.W[l];
            const float* __restrict__ b = m.b[l];
            float* tw = lds + m.tail_off + (l - 1) * (3 * 4 * 16);
            for (int idx = tid; idx < m.n_tail * 64; idx += nthreads) {
                const int s = idx & 15, gg = (idx >> 4) & 3, j = idx >> 6;
                const int fo = Hout - m.n_tail + j, fi = 4 * s + gg;
                tw[idx] = fi < Hin ? W[fo * Hin + fi] : (fi == Hin ? b[fo] : 0.f);
            }
        }
    }
}

// Bijective XCD-aware remap: hardware places workgroup b on XCD b % 8; give each XCD a
// contiguous chunk of the tile space so neighbouring tiles (which share h cache lines) share an L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk / 8, r = nblk % 8, xcd = bid % 8, j = bid / 8;
    const unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + j;
}
