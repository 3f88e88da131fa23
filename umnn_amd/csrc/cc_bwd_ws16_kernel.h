// Weight-stationary workgroup-pipeline backward on fp16 PIECES (round 4).  Same pipeline, roles, LDS tiles and timetable as
// cc_bwd_ws_kernel.h (read its header first); what changes is the number format of everything that crosses the matrix cores
// (reference lines: ParallelNeuralIntegral.py:66-94,110-123).
//
// Why.  The backward needs the SIGN of every hidden pre-activation (LeakyReLU / ReLU kinks), so its forward recompute has to be
// fp32-accurate.  On bf16 pieces (8 significand bits) that takes THREE pieces and six cross terms: 48 of the 72 GEMM MFMAs of a
// tile-node and layer.  An fp16 piece carries 11 bits: two pieces hold 22 bits of a value, and the three products
// hi*hi + hi*lo + lo*hi are exact fp16 x fp16 -> fp32 products that miss only lo*lo (2^-22 of the term) -- measured on the matrix
// core (tools/ubench/f16_split.hip): 4.8e-8 of the sum of |terms| for K = 32, BETTER than an fp32 fma chain (7.8e-8); subnormal
// low pieces are kept by v_mfma_f32_16x16x32_f16 (probed there), so the absolute error of a piece pair is <= max(2^-22 |a|, 2^-25).
// The recompute therefore drops from 48 to 24 MFMAs per layer, the forward weights from 96 to 64 registers, the activation tiles
// lose their third-piece companions (13.5 KB of LDS), and the split of an activation pair is FOUR instructions
// (v_cvt_pk_f16_f32, two v_fma_mix_f32 that subtract the packed halves straight from the fp32 values, v_cvt_pk_f16_f32) instead
// of eleven.  The delta chain and the dW products run on the same two fp16 pieces (three terms each, as before on bf16).
//
// Range.  fp16 has five exponent bits.  Activations and weights of these nets are O(1e-3 .. 1e2): inside it, with subnormal low
// pieces below 2^-3 (absolute error 2^-25, see above).  Cotangents are not: delta = g (x - x0)/2 w_k f'(..) w_out .. is ~1e-8
// for a mean log-likelihood over 8192 rows.  They are therefore carried SCALED by a power of two sigma chosen per launch from
// max |g (x - x0) / 2| max_k w_k + max |g_fx| (cc_bwd_cotmax_kernel, one pass over g, x, x0 ahead of this kernel; sigma puts that
// maximum at 2^WS16_T): wave F3 multiplies the per-integral cotangent base by sigma once per tile, everything downstream is linear
// in it, and dc, d_theta leave multiplied by 1 / sigma (exact: powers of two).  Elements more than 2^-(WS16_T + 3) below the
// launch's largest root cotangent lose relative precision gracefully (one bit per binade; the absolute error stays 2^-25 sigma^-1,
// i.e. 2^-(25 + WS16_T) of the largest cotangent) -- they are the ones that do not matter to d_theta, and for d_h / d_x of their
// own rows the error is still below 2^-16 relative down to 2^-(WS16_T + 9) of the maximum.
// Overflow (|a_l| or |sigma delta_l| >= 65520) cannot be excluded for arbitrary weights, so it is DETECTED, not assumed away: an
// overflowing piece is +-inf, which reaches the output-layer sum of wave F3 (activations) or the dc sum of wave B1 (cotangents)
// as inf / NaN; both are checked (one compare per step / per tile) and raise a device flag, and the launcher queues the bf16
// kernel of cc_bwd_ws_kernel.h right behind this one with "run only if the flag is set": same outputs, rewritten.  inv_f launches
// (cotangent x -1/f^2, unbounded) stay on the bf16 kernel.
// FRONT = the middle stage of the three-stage backward (cc_backward_front.hip; MNISTExperiment's 31-100-50^4-1): wave Ca takes
// z_2 from HBM instead of computing layer 1, wave B1 writes delta_2 (un-scaled) back instead of dc / dW_1; single-chunk calls
// only, because d_theta slices are written here and the fallback must be able to rewrite them (1.26 -> 1.15 ms per MNIST block).
// Round 6: the three B waves are software-pipelined (tail of element u - 1 behind the GEMM of element u: ws16_role_Bp, ws16_role_B1p),
// which moves this kernel's timetable and rings off cc_bwd_ws_kernel.h's -- see "Rings of this kernel" below.
#pragma once
#include "cc_bwd_ws_kernel.h"

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

#ifndef WS16_T
#define WS16_T 8                                  // the launch's largest root cotangent lands in [2^(T-1), 2^T)
#endif
constexpr int W16_NP = 2;                         // fp16 pieces of every matrix operand (three cross terms)
// Rings of this kernel (round 6): waves B3 and B2 are software-pipelined like B1 (their tails one step behind their GEMMs: ws16_role_Bp), so
// delta_3 / delta_2 and everything behind them run one / two steps later than in cc_bwd_ws_kernel.h's timetable: a_2 lives nine steps, a_1
// thirteen (a_3 still five: a pipelined B wave fetches its signs with the GEMM and holds them a step), fourteen steps from Ca to B1's tail.
//   step u      Ca: a_1[u]          u+1/2  F1 -> a_2[u]        u+3/4  F2 -> a_3[u]        u+5/6  F3 -> S4[u]        u+7  Cb: delta_4[u]
//   step u+8    B3: GEMM;  Ca: dW_3          u+9   B3: tail -> delta_3[u]
//   step u+10   B2: GEMM;  dW_2 (Cb, F2)     u+11  B2: tail -> delta_2[u]
//   step u+12   B1: GEMM;  dW_1 (Cb, F1)     u+13  B1: tail (dc, dW_1[:,0])
// LDS: 33 activation / cotangent tiles of 4.5 KB + the S4 pair + the quadrature tables (W16_MAX_NODES = 256 nodes: longer rules run the bf16
// pipeline) = 155 KB.
constexpr int W16_NS1 = 13, W16_NS2 = 9, W16_NS3 = 5;
constexpr int W16_OFF_A1 = 0;
constexpr int W16_OFF_A2 = W16_OFF_A1 + W16_NS1 * WS_TILE;
constexpr int W16_OFF_A3 = W16_OFF_A2 + W16_NS2 * WS_TILE;
constexpr int W16_OFF_D = W16_OFF_A3 + W16_NS3 * WS_TILE;      // delta_l, l = 2..4: tile (l - 2) * 2 + (u & 1)
constexpr int W16_OFF_S4 = W16_OFF_D + 6 * WS_TILE;            // (no third-piece tiles: the S4 tiles follow the cotangent tiles)
constexpr int W16_DEPTH = 13;                                  // steps between an element entering (Ca) and leaving (B1's tail)
__host__ __device__ constexpr int w16_a_off(int l) { return l == 1 ? W16_OFF_A1 : (l == 2 ? W16_OFF_A2 : W16_OFF_A3); }
__host__ __device__ constexpr int w16_a_ns(int l) { return l == 1 ? W16_NS1 : (l == 2 ? W16_NS2 : W16_NS3); }
constexpr int W16_TILE_USHORTS = W16_OFF_S4 + 2 * WS_P3;          // everything ws16_clear_tiles zeroes
constexpr int W16_MAX_NODES = 256;                // quadrature tables in LDS: w_k at float k, s_k at float W16_MAX_NODES + k (n + 1 above it: the bf16 pipeline)
constexpr int W16_OFF_TAB = W16_TILE_USHORTS;
constexpr int W16_LDS_USHORTS = W16_OFF_TAB + 2 * W16_MAX_NODES * 2;
constexpr int W16_IMG = BT * BKS * W16_NP * FRAG; // staging image of one weight matrix (start of the launch only)
static_assert(6 * W16_IMG <= W16_TILE_USHORTS, "the staging images must fit the tile area");
static_assert(W16_LDS_USHORTS * 2 <= 160 * 1024, "LDS");

// device scalars of a launch (workspace): filled by cc_bwd_cotmax_kernel, read by every role that scales
struct Ws16Scal { unsigned cotmax, gfxmax, wmax, flag; };

__device__ __forceinline__ f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ ws_f32x16 ws16_mfma32(u32x4 a, u32x4 b, ws_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}
// first stage of the two-piece split of a pair: returns the packed leading pieces (round to nearest even), leaves the EXACT
// remainders in x0 / x1 -- v_fma_mix_f32 reads an f16 half of a register as a source operand, so no unpacking instruction
__device__ __forceinline__ unsigned h16_split_stage(float& x0, float& x1) {
    const unsigned bits = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, h16x2));
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(x0) : "v"(bits));
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x1) : "v"(bits));
    return bits;
}
__device__ __forceinline__ unsigned h16_split_last(float x0, float x1) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, h16x2));
}

template <int NRL>
__device__ __forceinline__ void h16_split_regs(const f32x4 (&act)[BT], BFrag<W16_NP>& bf) {
    constexpr int NLIVE = NRL > 0 ? NRL : 4 * BT;
#pragma unroll
    for (int s = 0; s < BKS; ++s) {
        unsigned q[4][W16_NP];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            q[j][0] = q[j][1] = 0u;
            if (8 * s + 2 * j < NLIVE) {
                float x0 = act[2 * s + (j >> 1)][2 * (j & 1)], x1 = act[2 * s + (j >> 1)][2 * (j & 1) + 1];
                q[j][0] = h16_split_stage(x0, x1);
                q[j][1] = h16_split_last(x0, x1);
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < W16_NP; ++k2) bf.v[s][k2] = u32x4{q[0][k2], q[1][k2], q[2][k2], q[3][k2]};
    }
}

// fragment image of W_l (rows = out features, K = in features incl. the constant-one feature / bias column) or of W_l^T
// (rows = in features, K = out features, nothing through the constant feature) as two fp16 pieces: the index arithmetic of
// stage_frag_image (cc_bwd_bf16_kernel.h), the pieces of this file
// LOSCALE (forward images): the low piece is stored as (W - hi) * 2^11.  Weights of these nets are below 2^-3, so the plain
// remainder would be an fp16 SUBNORMAL -- absolute error 2^-25 instead of 2^-23 |W|, which was the largest single term of the
// recompute's noise (measured as LeakyReLU kink decisions that differ from the six-term bf16 recompute at the benchmarked
// size: 624 rows of d_h against 191 for the exact-fp32 kernels); scaled, the low piece has its full 11 bits whatever |W| is.  The
// products with it accumulate in their own accumulators, added back times 2^-11 (exact) in front of the activation.
constexpr float W16_LOSCALE = 2048.f, W16_LOUNSCALE = 1.f / 2048.f;
template <bool TRANSPOSED>
__device__ __forceinline__ void ws16_stage_image(const MlpDev& m, int l, unsigned short* img, int tid, int nthreads) {
    const int Hin = m.width[l], Hout = m.width[l + 1];
    const float* __restrict__ W = m.W[l];
    const float* __restrict__ b = m.b[l];
    for (int idx = tid; idx < BT * BKS * FRAG; idx += nthreads) {
        const int j = idx & 7, ln = (idx >> 3) & 63, ts = idx >> 9;
        const int s = ts % BKS, t = ts / BKS;
        const int frow = fout_of(t, ln & 15);
        const int fk = feat_of(2 * s + (j >> 2), j & 3, ln >> 4);
        float v = 0.f;
        if (!TRANSPOSED) {
            const int fo = frow, fi = fk;
            if (fo < Hout) {
                if (fi < Hin) v = W[fo * Hin + fi];
                else if (fi == Hin) v = b[fo];
            } else if (fo == Hout && fi == Hin) {
                v = 1.f;
            }
        } else {
            const int fi = frow, fo = fk;
            if (fo < Hout && fi < Hin) v = W[fo * Hin + fi];
        }
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)((v - (float)hi) * (TRANSPOSED ? 1.f : W16_LOSCALE));
        img[(ts * W16_NP + 0) * FRAG + ln * 8 + j] = __builtin_bit_cast(unsigned short, hi);
        img[(ts * W16_NP + 1) * FRAG + ln * 8 + j] = __builtin_bit_cast(unsigned short, lo);
    }
}

// sigma = 2^(WS16_T - ceil(log2 R)), R = the launch's largest root cotangent; 1 when there is nothing to scale
__device__ __forceinline__ float ws16_sigma(const Ws16Scal* sc, float& inv_sigma) {
    const float R = fmaf(__uint_as_float(sc->cotmax), __uint_as_float(sc->wmax), __uint_as_float(sc->gfxmax));
    int se = 127;
    if (R > 0.f && R < __builtin_inff()) {
        const int e = (int)((__float_as_uint(R) >> 23) & 0xffu);        // R in [2^(e-127), 2^(e-126))
        se = 127 + WS16_T - (e - 126);
        se = se < 8 ? 8 : (se > 246 ? 246 : se);                        // (sigma and 1 / sigma both normal)
    }
    inv_sigma = __uint_as_float((unsigned)(254 - se) << 23);
    return __uint_as_float((unsigned)se << 23);
}
__device__ __forceinline__ bool ws16_finite(float v) { return __builtin_fabsf(v) < __builtin_inff(); }

// max |g (x - x0) / 2|, max |g_fx| over the integrals and max |w_k| over the nodes, as the bit patterns of non-negative floats
// (which order like unsigned integers): order-independent, so the result -- and with it sigma -- is deterministic
template <int = 0>           // (a template so that every translation unit that includes this header may hold a copy)
__global__ __launch_bounds__(256) void cc_bwd_cotmax_kernel(const BwdArgs a, Ws16Scal* sc) {
    __shared__ float red[3][4];
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < a.NI; q += (long long)gridDim.x * blockDim.x) {
        const float xv = io_ld(a.x, q, a.x_bf16), x0v = a.x0 ? io_ld(a.x0, q, a.x_bf16) : 0.f;
        m0 = fmaxf(m0, fabsf(io_ld(a.g, q, a.x_bf16) * (xv - x0v) * 0.5f));
        if (a.gfx) m1 = fmaxf(m1, fabsf(io_ld(a.gfx, q, a.x_bf16)));
    }
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k <= a.n; k += blockDim.x) m2 = fmaxf(m2, fabsf(a.ccw[k]));
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        m0 = fmaxf(m0, __shfl_xor(m0, o));
        m1 = fmaxf(m1, __shfl_xor(m1, o));
        m2 = fmaxf(m2, __shfl_xor(m2, o));
    }
    // one atomic per workgroup and scalar (the first version issued one per WAVE: 12 k same-address atomics = 96 us per call)
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = m0; red[1][threadIdx.x >> 6] = m1; red[2][threadIdx.x >> 6] = m2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        m0 = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
        m1 = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
        m2 = fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3]));
        // (an INFINITE cotangent has the largest bit pattern: sigma falls back to 1 and the run ends in the flag.  A NaN one is dropped
        // by fmaxf -- sigma then comes from the finite values -- and reaches the dc sum of its tile as NaN, which raises the flag too:
        // either way the bf16 pipeline rewrites the launch)
        atomicMax(&sc->cotmax, __float_as_uint(m0));
        if (a.gfx) atomicMax(&sc->gfxmax, __float_as_uint(m1));
        if (blockIdx.x == 0) atomicMax(&sc->wmax, __float_as_uint(m2));
    }
}

// the 12 matrix instructions of one dW layer on fp16 operands (cross terms outermost, as ws_dw_mfma)
template <int IDX>
__device__ __forceinline__ void ws16_dw_mfma(ws_f32x16 (&dW)[2][2], const WsOps& o) {
    constexpr int term = IDX / 4, to = (IDX % 4) / 2, ti = IDX % 2;
    constexpr int pa = term == 2 ? 1 : 0, pb = term == 1 ? 1 : 0;
    dW[to][ti] = ws16_mfma32(o.A[to][pa], o.B[ti][pb], dW[to][ti]);
}
// this workgroup's d_theta slice (layout: ws_write_dw), the accumulators un-scaled on the way out
__device__ __forceinline__ void ws16_write_dw(const BwdArgs& a, float* part, int l, const ws_f32x16 (&dW)[2][2], int lane, float inv_sigma) {
    const MlpDev& m = a.m;
    const int Hin = m.width[l], Hout = m.width[l + 1];
#pragma unroll
    for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int fo = slot_feature(32 * to + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3));
                const int fi = slot_feature(32 * ti + (lane & 31));
                if (fo < Hout) {
                    const int idx = fi < Hin ? a.poffW[l] + fo * Hin + fi : (fi == Hin ? a.poffb[l] + fo : -1);
                    if (idx >= 0) part[idx] = dW[to][ti][v] * inv_sigma;
                }
            }
}
__device__ __forceinline__ void ws16_zero_dw(ws_f32x16 (&dW)[2][2]) {
#pragma unroll
    for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int v = 0; v < 16; ++v) dW[to][ti][v] = 0.f;
}
__device__ __forceinline__ void ws16_clear_tiles(unsigned short* lds16) {
    __syncthreads();
    for (int i = threadIdx.x; i < W16_TILE_USHORTS / 8; i += blockDim.x) reinterpret_cast<u32x4*>(lds16)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
}
#define W16_LOAD_B(ops, At) do { ws_load_op<4>(ops, At, At); ws_load_op<6>(ops, At, At); ws_load_op<5>(ops, At, At); ws_load_op<7>(ops, At, At); } while (0)
#define W16_LOAD_A(ops, Dt) do { ws_load_op<0>(ops, Dt, Dt); ws_load_op<2>(ops, Dt, Dt); ws_load_op<1>(ops, Dt, Dt); ws_load_op<3>(ops, Dt, Dt); } while (0)

// half of a dW product: the 32-row block HALF of delta_{l+1}^T (A) against both 32-column blocks of a_l (B); six matrix
// instructions per tile-node.  The dW work of a tile-node is spread over FOUR waves this way (Ca: all of dW_3; Cb: the first half
// of dW_1 and the second of dW_2; F1: the other half of dW_1; F2: the other half of dW_2) -- in round 3's layout wave Cb carried
// two whole layers and paced the pipeline together with F3.
struct WsOpsH { u32x4 A[W16_NP], B[2][W16_NP]; };
template <int HALF, int PIECE>
__device__ __forceinline__ void ws16_load_hA(WsOpsH& o, const unsigned short* Dt) {
    const unsigned short* src = Dt + PIECE * 16 * TRS + 32 * HALF;
    const u32x2 x = ws_tr_read(src);
    const u32x2 y = ws_tr_read(src + 4 * TRS);
    o.A[PIECE] = u32x4{x[0], x[1], y[0], y[1]};
}
template <int TAU, int PIECE>
__device__ __forceinline__ void ws16_load_hB(WsOpsH& o, const unsigned short* At) {
    const unsigned short* src = At + PIECE * 16 * TRS + 32 * TAU;
    const u32x2 x = ws_tr_read(src);
    const u32x2 y = ws_tr_read(src + 4 * TRS);
    o.B[TAU][PIECE] = u32x4{x[0], x[1], y[0], y[1]};
}
__device__ __forceinline__ void ws16_load_hB_all(WsOpsH& o, const unsigned short* At) {
    ws16_load_hB<0, 0>(o, At); ws16_load_hB<1, 0>(o, At); ws16_load_hB<0, 1>(o, At); ws16_load_hB<1, 1>(o, At);
}
template <int IDX>
__device__ __forceinline__ void ws16_dwh_mfma(ws_f32x16 (&dW)[2], const WsOpsH& o) {
    constexpr int term = IDX / 2, ti = IDX % 2;
    constexpr int pa = term == 2 ? 1 : 0, pb = term == 1 ? 1 : 0;
    dW[ti] = ws16_mfma32(o.A[pa], o.B[ti][pb], dW[ti]);
}
template <int HALF>
__device__ __forceinline__ void ws16_write_dwh(const BwdArgs& a, float* part, int l, const ws_f32x16 (&dW)[2], int lane, float inv_sigma) {
    const MlpDev& m = a.m;
    const int Hin = m.width[l], Hout = m.width[l + 1];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int fo = slot_feature(32 * HALF + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3));
            const int fi = slot_feature(32 * ti + (lane & 31));
            if (fo < Hout) {
                const int idx = fi < Hin ? a.poffW[l] + fo * Hin + fi : (fi == Hin ? a.poffb[l] + fo : -1);
                if (idx >= 0) part[idx] = dW[ti][v] * inv_sigma;
            }
        }
}
// quadrature tables staged in LDS by the prologue (a scalar load per step would share the LDS counter and stall every step's
// operand wait by an L2 latency: 250 cycles per step in round 3's F3 / B1)
__device__ __forceinline__ float ws16_ccw(const unsigned short* lds16, int k) { return reinterpret_cast<const float*>(lds16 + W16_OFF_TAB)[k]; }
__device__ __forceinline__ float ws16_ccs(const unsigned short* lds16, int k) { return reinterpret_cast<const float*>(lds16 + W16_OFF_TAB)[W16_MAX_NODES + k]; }

// ============================================================================================================ wave Ca
// layer 1 (a_1 of a new tile-node per step, two fp16 pieces -> tile A1) and dW_3
template <int NRL, bool FRONT>
__device__ __forceinline__ void ws16_role_Ca(const BwdBf16Args& args, unsigned short* lds16, int S, const WsShape sh, float* part) {
    constexpr int NLIVE = NRL > 0 ? NRL : 4 * BT;
    constexpr int NPAIR = (NLIVE + 1) / 2;
    const BwdArgs& a = args.b;
    const MlpDev& m = a.m;
    const int lane = threadIdx.x & 63, g = lane >> 4, p = lane & 15;
    const int H1 = m.width[1], E = a.E, d = a.d;
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    const int own = p * TRS + g * 16;
    const int trb = (8 * (g >> 1) + (p >> 2)) * TRS + 16 * (g & 1) + 4 * (p & 3);
    const int nit = sh.nit;
    ws16_clear_tiles(lds16);
    float inv_sigma;
    (void)ws16_sigma(reinterpret_cast<const Ws16Scal*>(args.scal), inv_sigma);

    float w1x[BT][4];
    {
        const float* __restrict__ W0 = m.W[0];
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = feat_of(t, r, g);
                w1x[t][r] = (!FRONT && f < H1) ? W0[f * (1 + E)] : 0.f;
            }
    }
    ws_f32x16 dW[2][2];
    ws16_zero_dw(dW);

    WsCursor cu{0, 0};
    float xv = 0.f, x0v = 0.f, dxv = 0.f;
    f32x4 c[BT];
#pragma unroll
    for (int t = 0; t < BT; ++t) c[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Opening a tile: x, x0, the tile's 16 x E embedding values from HBM, c = b_1 + W_1[:, 1:] h on the fp32 matrix pipe; all loads
    // of eight K-steps issued before the first product (item_embedding_gemm)
    // FRONT (middle stage of the three-stage backward, cc_backward_front.hip): "layer 1" is the front kernel's z_2 out of HBM,
    // [tile][node][register][lane].  z0 = z_2 of node 0 of the current tile (the tangent element -- d z_2 / d t at node 0 -- needs
    // its signs).  A step is ~0.8 us and an HBM load ~1 us: fetched ONE element ahead into a register set that is copied at the end
    // of the step, the load never landed in time and the whole workgroup waited for this wave at every barrier (0.34 of the 1.16
    // ms of an MNIST block -- measured by building the kernel without the fetch).  So the elements alternate between two register
    // sets, zs[0] and zs[1], the step loop is unrolled by two (every register of a set keeps its name: no copies of registers with a
    // load in flight), and right after element s has been read out of its set the same set becomes the target of element s + 2:
    // two steps of distance, loads in flight across the step barrier (which waits for LDS traffic only).  The loads are
    // unconditional -- past the last element the address falls back to the start of the buffer -- so that a set is always
    // re-defined as a whole.
    const int nl2 = NRL > 0 ? NRL : args.nl2;
    float zs[2][BT][4], z0[BT][4];
#pragma unroll
    for (int t = 0; t < BT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { zs[0][t][r] = 0.f; zs[1][t][r] = 0.f; z0[t][r] = 0.f; }
    auto fetch_z = [&](const WsCursor& c2, float (&dst)[BT][4]) __attribute__((always_inline)) {
        const bool there = c2.j < nit;
        const size_t tile0 = (size_t)ws_grp(c2) * (size_t)(a.n + 1) * nl2 * 64 + lane;
        const bool tan = there && ws_is_tan(sh, c2);
        const size_t base = !there ? (size_t)lane
                                   : (tan ? (size_t)ws_grp(c2) * nl2 * 64 + lane : tile0 + (size_t)ws_node(sh, c2) * nl2 * 64);
        const float* __restrict__ src = tan ? args.tz2 : args.z2;
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 4 * t + r, jj = j < nl2 ? j : nl2 - 1;       // (unconditional loads: index clamped, value masked)
                if (j < NLIVE) { const float v = src[base + (size_t)jj * 64]; dst[t][r] = j < nl2 ? v : 0.f; }
            }
    };
    auto new_item = [&]() __attribute__((always_inline)) {
        if constexpr (FRONT) return;
        const long long q0 = (long long)(args.grp0 + ws_grp(cu)) * 16 + p;
        const long long qq = q0 < a.NI ? q0 : a.NI - 1;
        xv = io_ld(a.x, qq, a.x_bf16);
        x0v = a.x0 ? io_ld(a.x0, qq, a.x_bf16) : 0.f;
        dxv = xv - x0v;
        const long long bi = qq / d;
        const IoView hb = IoView{a.h, a.h_bf16} + (bi * ((long long)E * d) + (qq - bi * d));
        const float* __restrict__ b0 = m.b[0];
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = feat_of(t, r, g);
                c[t][r] = f < H1 ? b0[f] : (f == H1 ? 1.f : 0.f);
            }
        item_embedding_gemm<BT, 8>(hb, m.W[0], H1, E, d, g, p, c);
    };
    if (nit > 0) new_item();
    WsCursor cf = cu;                                   // (FRONT) the element the next fetch brings in
    if constexpr (FRONT) {
        fetch_z(cf, zs[0]); cf = ws_next(sh, cf);
        fetch_z(cf, zs[1]); cf = ws_next(sh, cf);
    }

    float actF[BT][4];
    unsigned qF[8][W16_NP];
    float rf0[8], rf1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { rf0[j] = rf1[j] = 0.f; qF[j][0] = qF[j][1] = 0u; }
#pragma unroll
    for (int t = 0; t < BT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) actF[t][r] = 0.f;

    WS_TIMING_DECL;
    int rA3 = ws_ring0<W16_NS3, WS_TILE>(8), rD4 = ws_ring0<2, WS_TILE>(8);
    int rO1 = ws_ring0<W16_NS1, WS_TILE>(0);
    bool live = cu.j < nit;
    bool is_tan = live && ws_is_tan(sh, cu);
    float tk = 0.f;
    if constexpr (!FRONT) {
        const int k = ws_node(sh, cu);
        const float uu = ws16_ccs(lds16, k) + 1.f;
        tk = (k == 0) ? xv : __fadd_rn(x0v, __fmul_rn(dxv, uu) * 0.5f);
    }
    WsOps ops;
    {
        const unsigned short* A3n = lds16 + W16_OFF_A3 + rA3 + trb;
        W16_LOAD_B(ops, A3n);
    }
    auto step = [&](auto parc) __attribute__((always_inline)) {
        constexpr int PAR = decltype(parc)::value;      // (FRONT) the register set of this step's element
        float (&zc)[BT][4] = zs[PAR];
        WS_T(t0);
        const WsCursor nx = live ? ws_next(sh, cu) : cu;
        const int kn = ws_node(sh, nx);
        const float ccs_n = ws16_ccs(lds16, kn);
        if constexpr (FRONT) {
            if (live && cu.e == 0) {
#pragma unroll
                for (int t = 0; t < BT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) z0[t][r] = zc[t][r];
            }
        }
        const unsigned short* D4 = lds16 + W16_OFF_D + 4 * WS_TILE + rD4 + trb;
        unsigned short* const O1 = lds16 + W16_OFF_A1 + rO1 + own;
        // operands of dW_3: the a_3 half came in before the barrier, the delta_4 half (written last step) here
        W16_LOAD_A(ops, D4);
        // layer 1 of element s (tangent element: w1 . act'(z_1) of node 0)
        auto layer1_reg = [&](auto ec, auto tanc) __attribute__((always_inline)) {
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4;
            constexpr bool TAN = decltype(tanc)::value;
            if constexpr (e < NLIVE) {
                if constexpr (FRONT) {
                    if constexpr (TAN) actF[t][r] = zc[t][r] * (z0[t][r] > 0.f ? 1.f : slope);
                    else actF[t][r] = hidden_act_f(zc[t][r], slope);
                } else {
                    const float z = fmaf(w1x[t][r], tk, c[t][r]);
                    if constexpr (TAN) actF[t][r] = w1x[t][r] * (z > 0.f ? 1.f : slope);
                    else actF[t][r] = hidden_act_f(z, slope);
                }
            }
        };
        auto pairF = [&](auto jc, auto stc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, st = decltype(stc)::value, t = j / 2, r = 2 * (j & 1);
            if constexpr (j < NPAIR) {
                if constexpr (st == 0) { rf0[j] = actF[t][r]; rf1[j] = actF[t][r + 1]; qF[j][0] = h16_split_stage(rf0[j], rf1[j]); }
                if constexpr (st == 1) qF[j][1] = h16_split_last(rf0[j], rf1[j]);
            }
        };
        auto store_a = [&](auto sc, auto kc) __attribute__((always_inline)) {
            constexpr int ks = decltype(sc)::value, k2 = decltype(kc)::value;
            *reinterpret_cast<u32x4*>(O1 + k2 * 16 * TRS + ks * 8) = u32x4{qF[4 * ks][k2], qF[4 * ks + 1][k2], qF[4 * ks + 2][k2], qF[4 * ks + 3][k2]};
        };
        WS_T(t1);
        if (is_tan) swp_static_for<16>([&](auto ec) { layer1_reg(ec, std::true_type{}); });
        else swp_static_for<16>([&](auto ec) { layer1_reg(ec, std::false_type{}); });
        if constexpr (FRONT) {
            __builtin_amdgcn_sched_barrier(0);
            fetch_z(cf, zc);                            // this set has been read: element s + 2 into it
            cf = ws_next(sh, cf);
        }
        __builtin_amdgcn_sched_barrier(0);
        swp_static_for<12>([&](auto nc) {
            constexpr int nn = decltype(nc)::value;
            WS_MARK(nn, 12);
            ws16_dw_mfma<nn>(dW, ops);
            // the split of a_1: two pairs per slot, stage by stage; then the four stores
            if constexpr (nn < 4) { pairF(std::integral_constant<int, 2 * nn>{}, std::integral_constant<int, 0>{}); pairF(std::integral_constant<int, 2 * nn + 1>{}, std::integral_constant<int, 0>{}); }
            if constexpr (nn >= 1 && nn < 5) { pairF(std::integral_constant<int, 2 * (nn - 1)>{}, std::integral_constant<int, 1>{}); pairF(std::integral_constant<int, 2 * (nn - 1) + 1>{}, std::integral_constant<int, 1>{}); }
            if constexpr (nn == 5 || nn == 6) store_a(std::integral_constant<int, 0>{}, std::integral_constant<int, nn - 5>{});
            if constexpr (nn == 7 || nn == 8) store_a(std::integral_constant<int, 1>{}, std::integral_constant<int, nn - 7>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        // next element: its item data if it opens a new tile, its node position from the table value fetched above
        if (live) {
            const bool crossed = nx.j != cu.j;
            cu = nx;
            live = cu.j < nit;
            if (crossed && live) new_item();
            is_tan = live && ws_is_tan(sh, cu);
            if constexpr (!FRONT) {
                const float uu = ccs_n + 1.f;
                tk = (kn == 0) ? xv : __fadd_rn(x0v, __fmul_rn(dxv, uu) * 0.5f);
            }
        }
        ws_adv<W16_NS3, WS_TILE>(rA3); ws_adv<2, WS_TILE>(rD4);
        ws_adv<W16_NS1, WS_TILE>(rO1);
        {   // next step's a_3 operand of dW_3 (a tile written four steps ago)
            const unsigned short* A3n = lds16 + W16_OFF_A3 + rA3 + trb;
            W16_LOAD_B(ops, A3n);
        }
        WS_T(t2);
        __syncthreads();
        WS_T(t3);
        WS_TIMING_ACC(t0, t1, t2, t3);
    };
    if constexpr (FRONT) {
        for (int s = 0; s < S; s += 2) {
            step(std::integral_constant<int, 0>{});
            if (s + 1 < S) step(std::integral_constant<int, 1>{});
        }
    } else {
        for (int s = 0; s < S; ++s) step(std::integral_constant<int, 0>{});
    }
    WS_TIMING_OUT(S);
    ws16_write_dw(a, part, 3, dW, lane, inv_sigma);
}

// ============================================================================================================ wave Cb
// delta_4 = dout w_out act'(a_4) (dout arrives scaled by sigma), the first half of dW_1 and the second half of dW_2
template <int NRL>
__device__ __forceinline__ void ws16_role_Cb(const BwdBf16Args& args, unsigned short* lds16, int S, const WsShape sh, float* part) {
    constexpr int NLIVE = NRL > 0 ? NRL : 4 * BT;
    constexpr int L = 4;
    const BwdArgs& a = args.b;
    const MlpDev& m = a.m;
    const int lane = threadIdx.x & 63, g = lane >> 4, p = lane & 15;
    const int HL = m.width[L];
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    const int own = p * TRS + g * 16;
    const int trb = (8 * (g >> 1) + (p >> 2)) * TRS + 16 * (g & 1) + 4 * (p & 3);
    ws16_clear_tiles(lds16);
    float inv_sigma;
    (void)ws16_sigma(reinterpret_cast<const Ws16Scal*>(args.scal), inv_sigma);
    float wout[BT][4];
#pragma unroll
    for (int t = 0; t < BT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = feat_of(t, r, g);
            wout[t][r] = f < HL ? m.W[L][f] : 0.f;       // (no cotangent through the constant feature's slot)
        }
    ws_f32x16 dW2[2], dW1[2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int v = 0; v < 16; ++v) { dW2[ti][v] = 0.f; dW1[ti][v] = 0.f; }
    unsigned q4[8][W16_NP];
    float rb0[8], rb1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { rb0[j] = rb1[j] = 0.f; q4[j][0] = q4[j][1] = 0u; }
    WS_TIMING_DECL;
    int rA2 = ws_ring0<W16_NS2, WS_TILE>(10), rA1 = ws_ring0<W16_NS1, WS_TILE>(12);
    int rD3 = ws_ring0<2, WS_TILE>(10), rD2 = ws_ring0<2, WS_TILE>(12);
    int rS4 = ws_ring0<2, WS_P3>(7), rD4 = ws_ring0<2, WS_TILE>(7);
    WsOpsH o2, o1;
    ws16_load_hB_all(o2, lds16 + W16_OFF_A2 + rA2 + trb);          // (first step: the tiles are still zero)
    ws16_load_hB_all(o1, lds16 + W16_OFF_A1 + rA1 + trb);
    for (int s = 0; s < S; ++s) {
        WS_T(t0);
        // delta_4 of element s - 7 from what F3 left a step ago, as micro-operations behind the matrix instructions below
        const unsigned short* S4 = lds16 + W16_OFF_S4 + rS4;
        unsigned short* const D4o = lds16 + W16_OFF_D + 4 * WS_TILE + rD4 + own;
        u32x4 sg4[BKS];
#pragma unroll
        for (int s2 = 0; s2 < BKS; ++s2) sg4[s2] = *reinterpret_cast<const u32x4*>(S4 + own + s2 * 8);
        const float dout = *reinterpret_cast<const float*>(S4 + p * TRS + 64);
        const unsigned short* D3 = lds16 + W16_OFF_D + 2 * WS_TILE + rD3 + trb;
        const unsigned short* D2 = lds16 + W16_OFF_D + 0 * WS_TILE + rD2 + trb;
        // (the a_2 / a_1 halves of the operands came in before the barrier: only the cotangent halves, written last step, are fetched here)
        ws16_load_hA<1, 0>(o2, D3); ws16_load_hA<1, 1>(o2, D3);
        float d4[BT][4];
        auto d4_reg = [&](auto ec) __attribute__((always_inline)) {
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4;
            if constexpr (e < NLIVE) d4[t][r] = (dout * wout[t][r]) * act_grad_q(sg4, t, r, slope);
            else d4[t][r] = 0.f;
        };
        auto pair4 = [&](auto jc, auto stc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, st = decltype(stc)::value, t = j / 2, r = 2 * (j & 1);
            if constexpr (2 * j < NLIVE) {
                if constexpr (st == 0) { rb0[j] = d4[t][r]; rb1[j] = d4[t][r + 1]; q4[j][0] = h16_split_stage(rb0[j], rb1[j]); }
                if constexpr (st == 1) q4[j][1] = h16_split_last(rb0[j], rb1[j]);
            }
        };
        auto store_d4 = [&](auto sc, auto kc) __attribute__((always_inline)) {
            constexpr int ks = decltype(sc)::value, k2 = decltype(kc)::value;
            *reinterpret_cast<u32x4*>(D4o + k2 * 16 * TRS + ks * 8) = u32x4{q4[4 * ks][k2], q4[4 * ks + 1][k2], q4[4 * ks + 2][k2], q4[4 * ks + 3][k2]};
        };
        WS_T(t1);
        swp_static_for<24>([&](auto nc) {
            constexpr int nn = decltype(nc)::value;
            WS_MARK(nn, 24);
            // twelve matrix instructions over the 24 slots: the dW_2 half on the even slots of the first half, the dW_1 half after it
            if constexpr (nn < 12 && nn % 2 == 0) ws16_dwh_mfma<nn / 2>(dW2, o2);
            if constexpr (nn >= 12 && nn % 2 == 0) ws16_dwh_mfma<(nn - 12) / 2>(dW1, o1);
            if constexpr (nn == 0) ws16_load_hA<0, 0>(o1, D2);
            if constexpr (nn == 1) ws16_load_hA<0, 1>(o1, D2);
            // one register per slot, pair j split at slots 2j + 2 / 2j + 3, K-steps stored at 10, 11 / 18, 19
            if constexpr (nn < 16) d4_reg(std::integral_constant<int, nn>{});
            if constexpr (nn >= 2 && nn < 18 && (nn % 2) == 0) pair4(std::integral_constant<int, (nn - 2) / 2>{}, std::integral_constant<int, 0>{});
            if constexpr (nn >= 3 && nn < 19 && (nn % 2) == 1) pair4(std::integral_constant<int, (nn - 3) / 2>{}, std::integral_constant<int, 1>{});
            if constexpr (nn == 10 || nn == 11) store_d4(std::integral_constant<int, 0>{}, std::integral_constant<int, nn - 10>{});
            if constexpr (nn == 18 || nn == 19) store_d4(std::integral_constant<int, 1>{}, std::integral_constant<int, nn - 18>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        ws_adv<W16_NS2, WS_TILE>(rA2); ws_adv<W16_NS1, WS_TILE>(rA1);
        ws_adv<2, WS_TILE>(rD3); ws_adv<2, WS_TILE>(rD2);
        ws_adv<2, WS_P3>(rS4); ws_adv<2, WS_TILE>(rD4);
        // next step's a_2 / a_1 operands (tiles written eight and twelve steps ago)
        ws16_load_hB_all(o2, lds16 + W16_OFF_A2 + rA2 + trb);
        ws16_load_hB_all(o1, lds16 + W16_OFF_A1 + rA1 + trb);
        WS_T(t2);
        __syncthreads();
        WS_T(t3);
        WS_TIMING_ACC(t0, t1, t2, t3);
    }
    WS_TIMING_OUT(S);
    ws16_write_dwh<1>(a, part, 2, dW2, lane, inv_sigma);
    ws16_write_dwh<0>(a, part, 1, dW1, lane, inv_sigma);
}

// ============================================================================================================ waves F1..F3
// forward GEMM of hidden layer LAYER -> LAYER + 1: W_l as two fp16 pieces (64 registers) for the whole launch; per step 24 MFMAs
// on one tile-node with the activation / split of the tile-node before behind them; F1, F2 also carry one half of a dW product
// (F1: the second half of dW_1, F2: the first half of dW_2); F3 ends in the output layer
template <int NRL, int LAYER>
__device__ __forceinline__ void ws16_role_F(const BwdBf16Args& args, unsigned short* lds16, int S, const WsShape sh, float* part) {
    constexpr int NLIVE = NRL > 0 ? NRL : 4 * BT;
    constexpr int NPAIR = (NLIVE + 1) / 2;
    constexpr int L = 4;
    constexpr int DF = 2 * LAYER - 1;                  // the GEMM works on element s - DF, its vector work runs one step later
    constexpr int DP = 2 * LAYER;
    constexpr bool IS_OUT = LAYER == 3;
    constexpr int LO = LAYER < 3 ? LAYER + 1 : 3;      // layer of the activation tile this wave writes (F3 writes the S4 tile instead)
    constexpr bool HAS_DW = LAYER < 3;                 // half a dW product: layer DWL, 32-row block DWH, element s - DD
    constexpr int DWL = LAYER == 1 ? 1 : 2, DWH = LAYER == 1 ? 1 : 0, DD = DWL == 1 ? 12 : 10;
    const BwdArgs& a = args.b;
    const MlpDev& m = a.m;
    const int lane = threadIdx.x & 63, g = lane >> 4, p = lane & 15;
    const int HL = m.width[L], n = a.n;
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    const int own = p * TRS + g * 16;
    const int trb = (8 * (g >> 1) + (p >> 2)) * TRS + 16 * (g & 1) + 4 * (p & 3);
    const int nit = sh.nit;
    u32x4 Wf[BT][BKS][W16_NP];
    {
        const unsigned short* imf = lds16 + (LAYER - 1) * W16_IMG + lane * 8;
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int s2 = 0; s2 < BKS; ++s2)
#pragma unroll
                for (int k2 = 0; k2 < W16_NP; ++k2) Wf[t][s2][k2] = *reinterpret_cast<const u32x4*>(imf + ((t * BKS + s2) * W16_NP + k2) * FRAG);
    }
    ws16_clear_tiles(lds16);
    float inv_sigma;
    const float sigma = ws16_sigma(reinterpret_cast<const Ws16Scal*>(args.scal), inv_sigma);

    float wout[BT][4];
#pragma unroll
    for (int t = 0; t < BT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            wout[t][r] = 0.f;
            if constexpr (IS_OUT) {
                const int f = feat_of(t, r, g);
                wout[t][r] = f < HL ? m.W[L][f] : (f == HL ? m.b[L][0] : 0.f);
            }
        }
    f32x4 dwo[BT];
#pragma unroll
    for (int t = 0; t < BT; ++t) dwo[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    WsCursor cp{0, 0};                                  // element of the vector work (s - DP)
    float xvP = 0.f, x0vP = 0.f, gvP = 0.f, gfxvP = 0.f, gfxS = 0.f, cotbase = 0.f;
    float fxv = 0.f, fx0v = 0.f, dfdt = 0.f, fp0 = 0.f;
    bool bad = false;                                   // an inf / NaN reached the output-layer sum: some piece overflowed
    auto new_item_P = [&]() __attribute__((always_inline)) {
        if constexpr (IS_OUT) {
            const long long q = (long long)(args.grp0 + ws_grp(cp)) * 16 + p;
            const bool ok = q < a.NI;
            const long long qq = ok ? q : a.NI - 1;
            xvP = io_ld(a.x, qq, a.x_bf16);
            x0vP = a.x0 ? io_ld(a.x0, qq, a.x_bf16) : 0.f;
            gvP = ok ? io_ld(a.g, qq, a.x_bf16) : 0.f;
            gfxvP = (ok && a.gfx) ? io_ld(a.gfx, qq, a.x_bf16) : 0.f;
            cotbase = (gvP * (xvP - x0vP) * 0.5f) * sigma;             // (everything downstream of here carries sigma)
            gfxS = gfxvP * sigma;
            fxv = 0.f; fx0v = 0.f; dfdt = 0.f;
        }
    };
    if (nit > 0) new_item_P();

    f32x4 acc[BT], acc2[BT];                           // acc2: the products with the scaled low pieces of W_l
    float actF[BT][4];
#pragma unroll
    for (int t = 0; t < BT; ++t) {
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) actF[t][r] = 0.f;
    }
    unsigned qF[8][W16_NP], q4[8];
    float rf0[8], rf1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { rf0[j] = rf1[j] = 0.f; qF[j][0] = qF[j][1] = 0u; q4[j] = 0u; }
    float sd4[4] = {0.f, 0.f, 0.f, 0.f}, doutN = 0.f;
    float sc_a = 0.f, sc_sd = 0.f, sc_ex = 0.f, sc_f = 0.f, sc_fp = 0.f;      // (ELU + 1 outputs only: the launcher keeps sigmoid nets on the bf16 pipeline)

    WS_TIMING_DECL;
    int rAin = ws_ring0<w16_a_ns(LAYER), WS_TILE>(DF);
    int rAout = ws_ring0<w16_a_ns(LO), WS_TILE>(DP), rD4 = ws_ring0<2, WS_P3>(DP);
    float ccwP = 0.f;
    if constexpr (IS_OUT) ccwP = ws16_ccw(lds16, ws_node(sh, cp));
    ws_f32x16 dWh[2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int v = 0; v < 16; ++v) dWh[ti][v] = 0.f;
    int rAdw = ws_ring0<w16_a_ns(DWL), WS_TILE>(DD), rDdw = ws_ring0<2, WS_TILE>(DD);
    WsOpsH oh;
    if constexpr (HAS_DW) ws16_load_hB_all(oh, lds16 + w16_a_off(DWL) + rAdw + trb);
    for (int s = 0; s < S; ++s) {
        WS_T(t0);
        const bool liveP = s >= DP && cp.j < nit;
        const bool tanP = liveP && ws_is_tan(sh, cp);
        const int kP = ws_node(sh, cp);
        const WsCursor nxP = liveP ? ws_next(sh, cp) : cp;
        float ccw_n = 0.f;
        if constexpr (IS_OUT) {
            ccw_n = ws16_ccw(lds16, ws_node(sh, nxP));
            if (liveP && cp.e == 0) new_item_P();
        }
        const float nodef = (liveP && !tanP) ? 1.f : 0.f, k0f = kP == 0 ? 1.f : 0.f;
        const unsigned short* Ain = lds16 + w16_a_off(LAYER) + rAin + own;                  // a_l[s - DF]
        unsigned short* const Aout = lds16 + w16_a_off(LO) + rAout + own;                   // a_{l+1}[s - DP]
        unsigned short* const S4out = lds16 + W16_OFF_S4 + rD4;                            // F3: leading piece of a_4[s - 6], dout
        BFrag<W16_NP> bf;
#pragma unroll
        for (int s2 = 0; s2 < BKS; ++s2)
#pragma unroll
            for (int k2 = 0; k2 < W16_NP; ++k2) bf.v[s2][k2] = *reinterpret_cast<const u32x4*>(Ain + k2 * 16 * TRS + s2 * 8);
        if constexpr (HAS_DW) {
            const unsigned short* Ddw = lds16 + W16_OFF_D + (DWL + 1 - 2) * 2 * WS_TILE + rDdw + trb;      // delta_{DWL+1}[s - DD]
            ws16_load_hA<DWH, 0>(oh, Ddw); ws16_load_hA<DWH, 1>(oh, Ddw);
        }

        // ---- micro-operations of the vector work of element s - DP
        auto act_reg = [&](auto ec, auto tanc) __attribute__((always_inline)) {
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4, j = e / 2;
            constexpr bool TAN = decltype(tanc)::value;
            if constexpr (e < NLIVE) {
                acc[t][r] = fmaf(acc2[t][r], W16_LOUNSCALE, acc[t][r]);
                if constexpr (!IS_OUT) {
                    if constexpr (TAN) {
                        // tangent element: times act'(a_{l+1}) of the element before (node 0), read off its leading piece
                        const unsigned u = qF[j][0];
                        const bool pos = (e & 1) ? ((int)u > 0xffff) : ((short)(u & 0xffffu) > 0);
                        actF[t][r] = acc[t][r] * (pos ? 1.f : slope);
                    } else {
                        actF[t][r] = hidden_act_f(acc[t][r], slope);
                    }
                } else {
                    actF[t][r] = hidden_act_f(acc[t][r], slope);
                }
            }
        };
        auto pairF = [&](auto jc, auto stc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, st = decltype(stc)::value, t = j / 2, r = 2 * (j & 1);
            if constexpr (j < NPAIR) {
                if constexpr (st == 0) { rf0[j] = actF[t][r]; rf1[j] = actF[t][r + 1]; qF[j][0] = h16_split_stage(rf0[j], rf1[j]); }
                if constexpr (st == 1) qF[j][1] = h16_split_last(rf0[j], rf1[j]);
            }
        };
        auto store_a = [&](auto sc, auto kc) __attribute__((always_inline)) {
            constexpr int ks = decltype(sc)::value, k2 = decltype(kc)::value;
            *reinterpret_cast<u32x4*>(Aout + k2 * 16 * TRS + ks * 8) = u32x4{qF[4 * ks][k2], qF[4 * ks + 1][k2], qF[4 * ks + 2][k2], qF[4 * ks + 3][k2]};
        };
        // F3: output layer of element s - 6 (four partial sums: no thirteen-deep dependent chain in front of the matrix loop)
        auto out_dot = [&](auto ec) __attribute__((always_inline)) {
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4;
            if constexpr (e < 4) sd4[e] = 0.f;
            if constexpr (e < NLIVE) sd4[r] = fmaf(wout[t][r], actF[t][r], sd4[r]);
        };
        // tangent element (once per tile): the dot product of w_out with d a_L / d t = acc . act'(a_L of node 0) instead; the sign of
        // a_L(node 0) is read back from the S4 tile this wave wrote a step ago, so that no activation value is carried from step
        // to step (the carried copy cost ~40 register moves per step on BOTH paths)
        auto out_dot_tangent = [&]() __attribute__((always_inline)) {
            const unsigned short* S4prev = lds16 + W16_OFF_S4 + (rD4 ^ WS_P3) + own;
            u32x4 sgp[BKS];
#pragma unroll
            for (int s2 = 0; s2 < BKS; ++s2) sgp[s2] = *reinterpret_cast<const u32x4*>(S4prev + s2 * 8);
#pragma unroll
            for (int r = 0; r < 4; ++r) sd4[r] = 0.f;
#pragma unroll
            for (int t = 0; t < BT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * t + r < NLIVE) sd4[r] = fmaf(wout[t][r], acc[t][r] * act_grad_q(sgp, t, r, slope), sd4[r]);
        };
        // the scalar part in stages, one per matrix-instruction slot (a serial chain: cross-lane sum, exp, reciprocal, selects)
        auto out_scalar = [&](auto stc) __attribute__((always_inline)) {
            constexpr int st = decltype(stc)::value;
            if constexpr (st == 0) sc_a = (sd4[0] + sd4[1]) + (sd4[2] + sd4[3]);
            if constexpr (st == 1) {
                const unsigned u = __float_as_uint(sc_a);
                auto q = __builtin_amdgcn_permlane16_swap(u, u, false, false);
                sc_a = __uint_as_float(q[0]) + __uint_as_float(q[1]);
            }
            if constexpr (st == 2) {
                const unsigned w = __float_as_uint(sc_a);
                auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
                sc_sd = __uint_as_float(q[0]) + __uint_as_float(q[1]);
            }
            if constexpr (st == 3) sc_ex = __expf(sc_sd);
            if constexpr (st == 4) {
                sc_f = sc_sd > 0.f ? sc_sd + 1.f : sc_ex;
                sc_fp = sc_sd > 0.f ? 1.f : sc_ex;
            }
            if constexpr (st == 5) bad = bad || !ws16_finite(sc_sd);
            if constexpr (st == 6) {
                // (no cotangent flows back from a tangent element or a drained slot: their factor is 0; g_fx enters at node 0)
                const float cot = fmaf(cotbase, ccwP, gfxS * k0f) * nodef;
                doutN = cot * sc_fp;
            }
        };
        auto out_reg = [&](auto ec) __attribute__((always_inline)) {       // d w_out += dout a_L  (delta_L is formed by wave Cb)
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4;
            if constexpr (e < NLIVE) dwo[t][r] = fmaf(doutN, actF[t][r], dwo[t][r]);
        };
        auto pair4 = [&](auto jc) __attribute__((always_inline)) {          // leading piece of a_L: carries its sign
            constexpr int j = decltype(jc)::value, t = j / 2, r = 2 * (j & 1);
            if constexpr (j < NPAIR) q4[j] = h16_split_last(actF[t][r], actF[t][r + 1]);
        };
        auto store_s4 = [&](auto sc) __attribute__((always_inline)) {
            constexpr int ks = decltype(sc)::value;
            *reinterpret_cast<u32x4*>(S4out + own + ks * 8) = u32x4{q4[4 * ks], q4[4 * ks + 1], q4[4 * ks + 2], q4[4 * ks + 3]};
        };
        WS_T(t1);
        // ---- the activations (and F3's dot product) of element s - DP first, as one burst of vector instructions while the operand
        // fetches above are in flight; then the GEMM of element s - DF (24 MFMAs, A operands = this wave's registers) with the rest of
        // the vector work behind its matrix instructions.  Measured alternatives (EXPERIMENTS.md): the burst spread behind the matrix
        // instructions too makes every F role 5-20 % longer (a vector instruction directly behind an MFMA waits out that
        // instruction's issue passes); a second copy of the matrix loop for tangent elements costs 8 % of the kernel (the eight role
        // loops together sit at the instruction cache's capacity).  Only these activations differ for a tangent element, so the
        // uniform branch covers the burst alone.
        if constexpr (IS_OUT) {
            swp_static_for<16>([&](auto ec) { act_reg(ec, std::false_type{}); });
            swp_static_for<16>([&](auto ec) { out_dot(ec); });
            if (tanP) out_dot_tangent();
        } else {
            if (tanP) swp_static_for<16>([&](auto ec) { act_reg(ec, std::true_type{}); });
            else swp_static_for<16>([&](auto ec) { act_reg(ec, std::false_type{}); });
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            swp_static_for<24>([&](auto nc) {
                constexpr int nn = decltype(nc)::value;
                WS_MARK(nn, 24);
                constexpr int s2 = nn / 12, term = (nn % 12) / 4, t = nn % 4;
                constexpr int wa = term == 2 ? 1 : 0, ba = term == 1 ? 1 : 0;
                // (the products with the scaled low weight piece go to acc2.  Scaling the low ACTIVATION piece the same way was
                // measured too: 467 -> 449 differing rows for +5 % kernel time -- not kept)
                if constexpr (wa == 0) acc[t] = mfma_f16(Wf[t][s2][0], bf.v[s2][ba], nn < 4 ? zero : acc[t]);
                else acc2[t] = mfma_f16(Wf[t][s2][1], bf.v[s2][0], s2 == 0 ? zero : acc2[t]);
                if constexpr (HAS_DW && nn % 4 == 3) ws16_dwh_mfma<nn / 4>(dWh, oh);
                if constexpr (!IS_OUT) {
                    // the split of a_{l+1}: per K-step the four pairs stage by stage, then its two stores (20 micro-operations)
                    if constexpr (nn < 20) {
                        constexpr int ks = nn / 10, w = nn % 10;
                        if constexpr (w < 8) pairF(std::integral_constant<int, 4 * ks + w % 4>{}, std::integral_constant<int, w / 4>{});
                        else store_a(std::integral_constant<int, ks>{}, std::integral_constant<int, w - 8>{});
                    }
                } else {
                    // the leading piece of a_L does not wait for the scalar chain; dout and d w_out follow it
                    if constexpr (nn < 7) out_scalar(std::integral_constant<int, nn>{});
                    if constexpr (nn < 8) pair4(std::integral_constant<int, nn>{});
                    if constexpr (nn == 8 || nn == 9) store_s4(std::integral_constant<int, nn - 8>{});
                    if constexpr (nn == 9) *reinterpret_cast<float*>(S4out + p * TRS + 64) = doutN;
                    if constexpr (nn >= 10 && nn < 18) { out_reg(std::integral_constant<int, 2 * (nn - 10)>{}); out_reg(std::integral_constant<int, 2 * (nn - 10) + 1>{}); }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // ---- what only some elements of a tile do (uniform branches outside the scheduled region)
        if constexpr (IS_OUT) {
            if (liveP) {
                if (tanP) dfdt = fp0 * sc_sd;            // the sum was w_out . d a_L / d t at node 0 -> d f / d t
                else if (kP == 0) { fxv = sc_f; fp0 = sc_fp; }
                else if (kP == n) fx0v = sc_f;
            }
            if (liveP && cp.e == sh.ne - 1) {
                const long long q = (long long)(args.grp0 + ws_grp(cp)) * 16 + p;
                if (q < a.NI && g == 0) {
                    if (a.dx) io_st(a.dx, q, fmaf(gfxvP, dfdt, fxv * gvP), a.x_bf16);
                    if (a.dx0) io_st(a.dx0, q, -fx0v * gvP, a.x_bf16);
                }
            }
        }
        cp = nxP;
        ccwP = ccw_n;
        ws_adv<w16_a_ns(LAYER), WS_TILE>(rAin);
        ws_adv<w16_a_ns(LO), WS_TILE>(rAout); ws_adv<2, WS_P3>(rD4);
        if constexpr (HAS_DW) {
            ws_adv<w16_a_ns(DWL), WS_TILE>(rAdw); ws_adv<2, WS_TILE>(rDdw);
            ws16_load_hB_all(oh, lds16 + w16_a_off(DWL) + rAdw + trb);      // next step's a_l operand of the dW half (an old tile)
        }
        WS_T(t2);
        __syncthreads();
        WS_T(t3);
        WS_TIMING_ACC(t0, t1, t2, t3);
    }
    WS_TIMING_OUT(S);
    if constexpr (HAS_DW) ws16_write_dwh<DWH>(a, part, DWL, dWh, lane, inv_sigma);
    if constexpr (IS_OUT) {
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = dwo[t][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
                const int f = feat_of(t, r, g);
                if (p == 0) {
                    const int idx = f < HL ? a.poffW[L] + f : (f == HL ? a.poffb[L] : -1);
                    if (idx >= 0) part[idx] = v * inv_sigma;
                }
            }
        if (__any(bad) && lane == 0) atomicOr(&reinterpret_cast<Ws16Scal*>(args.scal)->flag, 1u);
    }
}

// ============================================================================================================ wave B3, software-pipelined (round 6)
// After B1 (below) had been pipelined, F3 alone paced the step, and F3 shares its SIMD with B3 -- whose 72-instruction tail ran un-shadowed
// behind its GEMM.  B3's result IS waited for (by B2 and the dW_2 products), so lagging its tail by a step moves everything behind it one step
// later: two more ring slots (a_2: 9, a_1: 12), nothing else -- the signs of a_3 are fetched with the GEMM and held, so a_3's ring stays.
// Step s: W_3^T GEMM of element s - 8 into one of two alternating accumulator sets; behind its matrix instructions the tail of element
// s - 9 from the other set: act'(a_3), the two-piece split, the four stores of delta_3.  Same arithmetic in the same order: bit-identical.
// LAYER = 2 (wave B2) the same one step on: GEMM of element s - 10, tail -> delta_2 of element s - 11 (one more slot for a_1).
template <int NRL, int LAYER>
__device__ __forceinline__ void ws16_role_Bp(const BwdBf16Args& args, unsigned short* lds16, int S) {
    constexpr int NLIVE = NRL > 0 ? NRL : 4 * BT;
    constexpr int NPAIR = (NLIVE + 1) / 2;
    static_assert(LAYER == 3 || LAYER == 2, "the pipelined middle B waves");
    constexpr int DB = LAYER == 3 ? 8 : 10, DT = DB + 1;          // GEMM on element s - DB, tail (-> delta_LAYER) on element s - DT
    const MlpDev& m = args.b.m;
    const int lane = threadIdx.x & 63, g = lane >> 4, p = lane & 15;
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    const int own = p * TRS + g * 16;
    u32x4 WT[BT][BKS][W16_NP];
    {
        const unsigned short* imt = lds16 + (3 + LAYER - 1) * W16_IMG + lane * 8;
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int s2 = 0; s2 < BKS; ++s2)
#pragma unroll
                for (int k2 = 0; k2 < W16_NP; ++k2) WT[t][s2][k2] = *reinterpret_cast<const u32x4*>(imt + ((t * BKS + s2) * W16_NP + k2) * FRAG);
    }
    ws16_clear_tiles(lds16);
    WS_TIMING_DECL;
    int rAsg = ws_ring0<w16_a_ns(LAYER), WS_TILE>(DB), rDin = ws_ring0<2, WS_TILE>(DB), rDout = ws_ring0<2, WS_TILE>(DT);
    f32x4 nds[2][BT];
    u32x4 sgs[2][BKS];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int t = 0; t < BT; ++t) nds[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < BKS; ++s2) sgs[q][s2] = u32x4{0u, 0u, 0u, 0u};
    }
    float dl[BT][4];
    unsigned q3[8][W16_NP];
    float rb0[8], rb1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { rb0[j] = rb1[j] = 0.f; q3[j][0] = q3[j][1] = 0u; }
#pragma unroll
    for (int t = 0; t < BT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dl[t][r] = 0.f;
    auto step = [&](auto parc) __attribute__((always_inline)) {
        constexpr int PAR = decltype(parc)::value;
        f32x4 (&nd)[BT] = nds[PAR];
        const f32x4 (&ndp)[BT] = nds[PAR ^ 1];
        const u32x4 (&sgp)[BKS] = sgs[PAR ^ 1];
        WS_T(t0);
        const unsigned short* Asg = lds16 + w16_a_off(LAYER) + rAsg + own;                                 // a_l[s - DB]
        const unsigned short* Din = lds16 + W16_OFF_D + (LAYER + 1 - 2) * 2 * WS_TILE + rDin + own;        // delta_{l+1}[s - DB]
        unsigned short* const Dout = lds16 + W16_OFF_D + (LAYER - 2) * 2 * WS_TILE + rDout + own;          // delta_l[s - DT]
        BFrag<W16_NP> bd;
#pragma unroll
        for (int s2 = 0; s2 < BKS; ++s2)
#pragma unroll
            for (int k2 = 0; k2 < W16_NP; ++k2) bd.v[s2][k2] = *reinterpret_cast<const u32x4*>(Din + k2 * 16 * TRS + s2 * 8);
#pragma unroll
        for (int s2 = 0; s2 < BKS; ++s2) sgs[PAR][s2] = *reinterpret_cast<const u32x4*>(Asg + s2 * 8);
        WS_T(t1);
        auto dl_reg = [&](auto ec) __attribute__((always_inline)) {
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4;
            if constexpr (e < NLIVE) dl[t][r] = ndp[t][r] * act_grad_q(sgp, t, r, slope);
            else dl[t][r] = 0.f;
        };
        auto pair3 = [&](auto jc, auto stc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, st = decltype(stc)::value, t = j / 2, r = 2 * (j & 1);
            if constexpr (j < NPAIR) {
                if constexpr (st == 0) { rb0[j] = dl[t][r]; rb1[j] = dl[t][r + 1]; q3[j][0] = h16_split_stage(rb0[j], rb1[j]); }
                if constexpr (st == 1) q3[j][1] = h16_split_last(rb0[j], rb1[j]);
            }
        };
        auto store_d3 = [&](auto sc, auto kc) __attribute__((always_inline)) {
            constexpr int ks = decltype(sc)::value, k2 = decltype(kc)::value;
            *reinterpret_cast<u32x4*>(Dout + k2 * 16 * TRS + ks * 8) = u32x4{q3[4 * ks][k2], q3[4 * ks + 1][k2], q3[4 * ks + 2][k2], q3[4 * ks + 3][k2]};
        };
        {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            swp_static_for<24>([&](auto nc) {
                constexpr int nn = decltype(nc)::value;
                WS_MARK(nn, 24);
                constexpr int s2 = nn / 12, term = (nn % 12) / 4, t = nn % 4;
                if constexpr (term < 2) nd[t] = mfma_f16(WT[t][s2][0], bd.v[s2][term], (s2 == 0 && term == 0) ? zero : nd[t]);
                else nd[t] = mfma_f16(WT[t][s2][1], bd.v[s2][0], nd[t]);
                // the tail of element s - DT (wave Cb's slot plan): one register per slot, pair j split at slots 2j + 2 / 2j + 3, its K-steps
                // stored at 10, 11 / 18, 19
                if constexpr (nn < 16) dl_reg(std::integral_constant<int, nn>{});
                if constexpr (nn >= 2 && nn < 18 && (nn % 2) == 0) pair3(std::integral_constant<int, (nn - 2) / 2>{}, std::integral_constant<int, 0>{});
                if constexpr (nn >= 3 && nn < 19 && (nn % 2) == 1) pair3(std::integral_constant<int, (nn - 3) / 2>{}, std::integral_constant<int, 1>{});
                if constexpr (nn == 10 || nn == 11) store_d3(std::integral_constant<int, 0>{}, std::integral_constant<int, nn - 10>{});
                if constexpr (nn == 18 || nn == 19) store_d3(std::integral_constant<int, 1>{}, std::integral_constant<int, nn - 18>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        ws_adv<w16_a_ns(LAYER), WS_TILE>(rAsg); ws_adv<2, WS_TILE>(rDin); ws_adv<2, WS_TILE>(rDout);
        WS_T(t2);
        __syncthreads();
        WS_T(t3);
        WS_TIMING_ACC(t0, t1, t2, t3);
    };
    for (int s = 0; s < S; s += 2) {
        step(std::integral_constant<int, 0>{});
        if (s + 1 < S) step(std::integral_constant<int, 1>{});
    }
    WS_TIMING_OUT(S);
}

// ============================================================================================================ wave B1, software-pipelined (round 6)
// B1 is the LAST role of the pipeline: nothing downstream waits for its result, so its vector tail -- act'(a_1), the dc sums, dW_1[:, 0]:
// 65 instructions per step that ran un-shadowed behind the 24 matrix instructions of its GEMM -- can lag the GEMM by one step at no cost
// in rings or latency.  Step s: W_1^T GEMM of element s - 12 into one of two alternating accumulator sets, with the tail of element
// s - 13 (the OTHER set, and the a_1 signs fetched a step ago) as fillers behind its matrix instructions.  Timing-only ablations had
// put B1 (with F3) on the critical path: its two accumulations alone were 2.9 % of the kernel (EXPERIMENTS.md).  Same arithmetic, same
// summation order per integral: results are bit-identical to the un-pipelined role.
// FRONT (middle stage of the three-stage backward): the tail is act'(a_2) and the un-scaling of delta_2 as fillers; its 13 stores to HBM
// and the finiteness check stay one uniform block behind the schedule.
template <int NRL, bool FRONT>
__device__ __forceinline__ void ws16_role_B1p(const BwdBf16Args& args, unsigned short* lds16, int S, const WsShape sh, float* part) {
    constexpr int NLIVE = NRL > 0 ? NRL : 4 * BT;
    constexpr int LAYER = 1, DB = 12, DT = DB + 1;                // GEMM on element s - DB, tail on element s - DT
    const BwdArgs& a = args.b;
    const MlpDev& m = a.m;
    const int lane = threadIdx.x & 63, g = lane >> 4, p = lane & 15;
    const int H1 = m.width[1], E = a.E;
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    const int own = p * TRS + g * 16;
    const int nit = sh.nit;
    u32x4 WT[BT][BKS][W16_NP];
    {
        const unsigned short* imt = lds16 + (3 + LAYER - 1) * W16_IMG + lane * 8;
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int s2 = 0; s2 < BKS; ++s2)
#pragma unroll
                for (int k2 = 0; k2 < W16_NP; ++k2) WT[t][s2][k2] = *reinterpret_cast<const u32x4*>(imt + ((t * BKS + s2) * W16_NP + k2) * FRAG);
    }
    ws16_clear_tiles(lds16);
    float inv_sigma;
    (void)ws16_sigma(reinterpret_cast<const Ws16Scal*>(args.scal), inv_sigma);

    f32x4 dW1x[BT], dcs[BT];
#pragma unroll
    for (int t = 0; t < BT; ++t) { dW1x[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dcs[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    bool bad = false;
    WsCursor cb{0, 0};                                  // the element of the TAIL (s - DT)
    float xvB = 0.f, x0vB = 0.f, dxvB = 0.f;
    auto new_item_B = [&]() __attribute__((always_inline)) {
        if constexpr (FRONT) return;
        const long long q = (long long)(args.grp0 + ws_grp(cb)) * 16 + p;
        const long long qq = q < a.NI ? q : a.NI - 1;
        xvB = io_ld(a.x, qq, a.x_bf16);
        x0vB = a.x0 ? io_ld(a.x0, qq, a.x_bf16) : 0.f;
        dxvB = xvB - x0vB;
    };
    if (nit > 0) new_item_B();

    WS_TIMING_DECL;
    int rAsg = ws_ring0<w16_a_ns(LAYER), WS_TILE>(DB), rDin = ws_ring0<2, WS_TILE>(DB);
    float tkB = 0.f;
    float d2v[BT][4];                                   // (FRONT) delta_2 of the tail's element, un-scaled, on its way to HBM
#pragma unroll
    for (int t = 0; t < BT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) d2v[t][r] = 0.f;
    if constexpr (!FRONT) {
        const int kB = ws_node(sh, cb);
        const float uu = ws16_ccs(lds16, kB) + 1.f;
        tkB = (kB == 0) ? xvB : __fadd_rn(x0vB, __fmul_rn(dxvB, uu) * 0.5f);
    }
    f32x4 nds[2][BT];
    u32x4 sgs[2][BKS];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int t = 0; t < BT; ++t) nds[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < BKS; ++s2) sgs[q][s2] = u32x4{0u, 0u, 0u, 0u};
    }
    int sstep = 0;
    auto step = [&](auto parc) __attribute__((always_inline)) {
        constexpr int PAR = decltype(parc)::value;
        f32x4 (&nd)[BT] = nds[PAR];                     // this step's GEMM output (element s - DB)
        const f32x4 (&ndp)[BT] = nds[PAR ^ 1];          // last step's: the tail's input (element s - DT)
        const u32x4 (&sgp)[BKS] = sgs[PAR ^ 1];
        const int s = sstep;
        WS_T(t0);
        const bool liveB = s >= DT && cb.j < nit;
        WsCursor nxB = cb;
        if (liveB) nxB = ws_next(sh, cb);
        const int kBn = ws_node(sh, nxB);
        float ccs_n = 0.f;
        if constexpr (!FRONT) ccs_n = ws16_ccs(lds16, kBn);
        const unsigned short* Asg = lds16 + w16_a_off(LAYER) + rAsg + own;                                  // a_1[s - DB]
        const unsigned short* Din = lds16 + W16_OFF_D + (LAYER + 1 - 2) * 2 * WS_TILE + rDin + own;         // delta_2[s - DB]
        BFrag<W16_NP> bd;
#pragma unroll
        for (int s2 = 0; s2 < BKS; ++s2)
#pragma unroll
            for (int k2 = 0; k2 < W16_NP; ++k2) bd.v[s2][k2] = *reinterpret_cast<const u32x4*>(Din + k2 * 16 * TRS + s2 * 8);
#pragma unroll
        for (int s2 = 0; s2 < BKS; ++s2) sgs[PAR][s2] = *reinterpret_cast<const u32x4*>(Asg + s2 * 8);
        WS_T(t1);
        // the tail of element s - DT, one register per micro-operation (5 vector instructions), behind the matrix instructions below
        auto tail_reg = [&](auto ec) __attribute__((always_inline)) {
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4;
            if constexpr (e < NLIVE) {
                const float dl = ndp[t][r] * act_grad_q(sgp, t, r, slope);
                if constexpr (FRONT) {
                    d2v[t][r] = dl * inv_sigma;
                } else {
                    dcs[t][r] += dl;
                    dW1x[t][r] = fmaf(dl, tkB, dW1x[t][r]);
                }
            }
        };
        {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            swp_static_for<24>([&](auto nc) {
                constexpr int nn = decltype(nc)::value;
                WS_MARK(nn, 24);
                // the matrix instructions in the order of the un-pipelined role (K-step, then cross term, tiles innermost)
                constexpr int s2 = nn / 12, term = (nn % 12) / 4, t = nn % 4;
                if constexpr (term < 2) nd[t] = mfma_f16(WT[t][s2][0], bd.v[s2][term], (s2 == 0 && term == 0) ? zero : nd[t]);
                else nd[t] = mfma_f16(WT[t][s2][1], bd.v[s2][0], nd[t]);
                // (the first operand fetch of the step is still in flight behind slot 0: the tail starts at slot 2, a register every
                // slot and a half)
                if constexpr (nn >= 2 && nn < 22 && ws_op_of_slot(nn - 2) >= 0) tail_reg(std::integral_constant<int, ws_op_of_slot(nn - 2)>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // ---- what only some elements of a tile do (uniform branches outside the scheduled region): the tail's element is the one in `cb`
        if constexpr (FRONT) {
            // delta_2 = dL/dz_2 of this node goes back to HBM for the front-backward kernel (not for the tangent element); a non-finite
            // value (an overflowed cotangent piece) raises the flag
            if (liveB && !ws_is_tan(sh, cb)) {
                const int nl2 = NRL > 0 ? NRL : args.nl2;
                const size_t base = ((size_t)ws_grp(cb) * (size_t)(a.n + 1) + (size_t)ws_node(sh, cb)) * nl2 * 64 + lane;
                float chk = 0.f;
#pragma unroll
                for (int t = 0; t < BT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * t + r < NLIVE && 4 * t + r < nl2) { args.d2[base + (size_t)(4 * t + r) * 64] = d2v[t][r]; chk = fmaf(d2v[t][r], 0.f, chk); }
                bad = bad || !(chk == 0.f);
            }
            if (liveB) cb = nxB;
        } else {
        if (liveB && cb.e == sh.ne - 1) {
            const long long q = (long long)(args.grp0 + ws_grp(cb)) * 16 + p;
            float chk = 0.f;
#pragma unroll
            for (int t = 0; t < BT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) chk = fmaf(dcs[t][r], 0.f, chk);         // (NaN iff some entry is inf / NaN)
            bad = bad || !(chk == 0.f);
            if (q < a.NI) {
#pragma unroll
                for (int t = 0; t < BT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = feat_of(t, r, g);
                        if (f < H1) a.dc[q * H1 + f] = dcs[t][r] * inv_sigma;
                    }
            }
#pragma unroll
            for (int t = 0; t < BT; ++t) dcs[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (liveB) {
            const bool crossed = nxB.j != cb.j;
            cb = nxB;
            if (crossed && cb.j < nit) new_item_B();
            const float uu = ccs_n + 1.f;
            tkB = (kBn == 0) ? xvB : __fadd_rn(x0vB, __fmul_rn(dxvB, uu) * 0.5f);
        }
        }
        ws_adv<w16_a_ns(LAYER), WS_TILE>(rAsg); ws_adv<2, WS_TILE>(rDin);
        WS_T(t2);
        __syncthreads();
        WS_T(t3);
        WS_TIMING_ACC(t0, t1, t2, t3);
        ++sstep;
    };
    // (S + 1 steps of tail work in S + 1 iterations would need one more barrier than the other roles run: the tail of the LAST element
    // that can be live -- GEMM'd at step S - 2, element nit * ne - 1 = S - 2 - DB -- runs at step S - 1: inside the S steps)
    for (int s = 0; s < S; s += 2) {
        step(std::integral_constant<int, 0>{});
        if (s + 1 < S) step(std::integral_constant<int, 1>{});
    }
    WS_TIMING_OUT(S);
    if constexpr (!FRONT) {
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = dW1x[t][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
                const int f = feat_of(t, r, g);
                if (p == 0 && f < H1) part[a.poffW[0] + f * (1 + E)] = v * inv_sigma;
            }
    }
    if (__any(bad) && lane == 0) atomicOr(&reinterpret_cast<Ws16Scal*>(args.scal)->flag, 1u);
}

// wave -> role.  Waves w and w + 4 share a SIMD: Ca + Cb, F1 + B1, F2 + B2, F3 + B3.
template <int NRL, bool FRONT = false>
__global__ __launch_bounds__(64 * WS_WAVES, 1) void cc_bwd_ws16_kernel(const BwdBf16Args args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned short* lds16 = reinterpret_cast<unsigned short*>(lds);
    const BwdArgs& a = args.b;
    const MlpDev& m = a.m;
    const int tid = threadIdx.x;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- the weights: staged once as fp16 fragment images, then read into the owners' registers; the quadrature tables
    for (int l = 1; l <= 3; ++l) {
        ws16_stage_image<false>(m, l, lds16 + (l - 1) * W16_IMG, tid, blockDim.x);
        ws16_stage_image<true>(m, l, lds16 + (3 + l - 1) * W16_IMG, tid, blockDim.x);
    }
    {
        float* tab = reinterpret_cast<float*>(lds16 + W16_OFF_TAB);
        for (int k = tid; k < W16_MAX_NODES; k += blockDim.x) {
            tab[k] = k <= a.n ? a.ccw[k] : 0.f;
            tab[W16_MAX_NODES + k] = k <= a.n ? a.ccs[k] : 0.f;
        }
    }
    __syncthreads();
    const int role = wid & 3, upper = wid >> 2;
    WsShape sh;
    sh.tan0 = a.gfx != nullptr ? 1 : 0;
    sh.ne = a.n + 1 + sh.tan0;
    sh.nit = blockIdx.x < a.ngroups ? (int)((a.ngroups - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
    const int S = sh.nit * sh.ne + W16_DEPTH;
    // one d_theta slice per workgroup (as the middle stage of the three-stage backward: the first of the workgroup's four -- the
    // front kernels use all four; single-chunk launches only, so every slice entry is written, never accumulated)
    float* part = a.partials + (size_t)blockIdx.x * (FRONT ? UMNN_WAVES_PER_BLOCK : 1) * a.n_params;
    if (!upper) {
        if (role == 0) ws16_role_Ca<NRL, FRONT>(args, lds16, S, sh, part);
        else if (role == 1) ws16_role_F<NRL, 1>(args, lds16, S, sh, part);
        else if (role == 2) ws16_role_F<NRL, 2>(args, lds16, S, sh, part);
        else ws16_role_F<NRL, 3>(args, lds16, S, sh, part);
    } else {
        if (role == 0) ws16_role_Cb<NRL>(args, lds16, S, sh, part);
        else if (role == 1) ws16_role_B1p<NRL, FRONT>(args, lds16, S, sh, part);
        else if (role == 2) ws16_role_Bp<NRL, 2>(args, lds16, S);
        else ws16_role_Bp<NRL, 3>(args, lds16, S);
    }
}
