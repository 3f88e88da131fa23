// Forward quadrature with the hidden-layer GEMMs on the bf16 matrix cores, fp32 accuracy recovered by splitting.
//
// Why: on gfx950 the fp32-input MFMA runs at the fp32 VECTOR rate and (measured) time-shares the
// SIMD's fp32 lanes with ordinary VALU work, so the exact-fp32 kernel is bounded by 32 cycles per 16x16x4 MFMA
// PLUS ~3 cycles per VALU instruction.  v_mfma_f32_16x16x32_bf16 has 16x the rate on a separate pipe.  Every fp32
// operand is split into bf16 pieces x = hi + lo (+ lo2), each the round-to-nearest bf16 of the running remainder,
// and the product W*a is formed from the significant cross terms, accumulated in fp32 inside the MFMA:
//     NPARTS = 2 (3 terms):  Whi*ahi + Whi*alo + Wlo*ahi                 error ~2^-16 per product (F to ~5e-6)
//     NPARTS = 3 (6 terms):  + Whi*alo2 + Wlo2*ahi + Wlo*alo             error ~2^-24: fp32-level (F to ~4e-7)
// Layer 1 (one FMA per feature), the hoisted first-layer term, the output dot product, ELU and the quadrature sum
// stay in fp32.  Same reference lines as cc_forward.hip.
//
// Layout: the same feature numbering as the fp32 kernels, lane (g,p) / tile t / component r <-> feature 16t + 4r + g
// (accumulator row rho = 4g + r of tile t is given output feature 16t + 4(rho&3) + (rho>>2) when the weight image
// is staged).  Features are therefore dense in the register index 4t + r: a layer of width H has only
// ceil((H+1)/4) live registers per lane (13 of 16 for H = 50) and the activation / split VALU work of the dead
// ones is skipped (NRL template parameter).  One K-step of 16x16x32 consumes 32 features = two tiles (2s, 2s+1);
// lane group g supplies k-slots 8g..8g+7 = its own 4 components of tile 2s followed by its 4 components of tile
// 2s+1 -- again the accumulators of one layer ARE (after activation, splitting and packing) the B operands of the
// next, with no cross-lane movement.  Weight fragments are pre-split and pre-permuted into LDS: fragment (tile t', K-step s, part)
// is 64 lanes x 8 bf16 = 1 KiB, read with one ds_read_b128 per lane.
#pragma once
#include <type_traits>
#include <utility>
#include "cc_bf16.h"
#include "cc_fwd_shared.h"
#include "cc_host.h"

// ---- the 16-bit PIECE type of this translation unit.  The kernels below are written once; cc_forward_bf16.hip / cc_invert.hip /
// cc_forward_p32.hip compile them with bf16 pieces (namespace fwd_bf16: 8 significand bits per piece, fp32's exponent range),
// cc_forward_f16.hip with fp16 pieces (-DUMNN_FWD_PIECE_F16, namespace fwd_f16: 11 bits per piece -- two pieces / three cross
// terms are then fp32-level, ~4e-7 on F instead of ~6e-6 -- but fp16's exponent range: a hidden activation beyond +-65504 overflows
// its leading piece; that is detected in the output-layer sum and turns the integral into NaN, never into a wrong finite number).
#ifdef UMNN_FWD_PIECE_F16
#define UMNN_FWD_NS fwd_f16
#else
#define UMNN_FWD_NS fwd_bf16
#endif
namespace UMNN_FWD_NS {
#ifdef UMNN_FWD_PIECE_F16
typedef _Float16 pc_x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pc_x4 __attribute__((ext_vector_type(4)));
typedef _Float16 pc_x2 __attribute__((ext_vector_type(2)));
constexpr unsigned PC_MINUS_ONE = 0xBC00u;
constexpr bool PC_F16 = true;
__device__ __forceinline__ f32x4 pc_mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pc_x8, a), __builtin_bit_cast(pc_x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 pc_mfma_k16(u32x2 a, u32x2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(pc_x4, a), __builtin_bit_cast(pc_x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pc_cvt_pk(float x0, float x1) {      // v_cvt_pk_f16_f32: round to nearest even
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, pc_x2));
}
__device__ __forceinline__ float pc_lo_f32(unsigned bits) { return (float)__builtin_bit_cast(pc_x2, bits)[0]; }
__device__ __forceinline__ float pc_hi_f32(unsigned bits) { return (float)__builtin_bit_cast(pc_x2, bits)[1]; }
#else
constexpr unsigned PC_MINUS_ONE = 0xBF80u;
constexpr bool PC_F16 = false;
__device__ __forceinline__ f32x4 pc_mfma(u32x4 a, u32x4 b, f32x4 c) { return mfma_bf16(a, b, c); }
__device__ __forceinline__ f32x4 pc_mfma_k16(u32x2 a, u32x2 b, f32x4 c) { return mfma_bf16_k16(a, b, c); }
__device__ __forceinline__ unsigned pc_cvt_pk(float x0, float x1) {      // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
}
__device__ __forceinline__ float pc_lo_f32(unsigned bits) { return __uint_as_float(bits << 16); }
__device__ __forceinline__ float pc_hi_f32(unsigned bits) { return __uint_as_float(bits & 0xffff0000u); }
#endif
// output activation.  fp16 pieces: an overflowed piece (|a| >= 65520) is +-inf and reaches the output-layer sum as inf / NaN; ELU + 1
// would map -inf to a finite 0, so a non-finite sum is made a NaN outright (bf16 pieces share fp32's range: nothing to guard).  The NaN
// is what the epilogue's overflow protocol looks for in the quadrature sum (cc_fwd_shared.h)
__device__ __forceinline__ float pc_out_act(float sd, int kind) {
    const float f = out_act_f(sd, kind);
    if constexpr (PC_F16) return __builtin_fmaf(sd, 0.f, f);      // sd * 0 = NaN for a non-finite sd, +-0 otherwise: one instruction
    return f;
}
#ifdef UMNN_FWD_PIECE_F16
// second fp16 piece of a pair, straight from the values and their packed leading pieces: v_fma_mixlo_f16 / v_fma_mixhi_f16 read an f16
// half of a register as an operand, form x - hi exactly (fp32 fma) and round it to f16 into one half of the destination -- the
// remainder never exists as a separate fp32 value: two instructions per pair where subtracting and converting takes five
// (tools/ubench/mixlo_check.hip: bit-identical to cvt_pk(x - float(hi)) over 1 M random pairs of 40 binades, zeros, subnormals).
// HAZARDS: the compiler's hazard recogniser does not look inside inline assembly, and with MFMA accumulators in VGPRs the register
// allocator hands a just-dead accumulator to the next temporary -- a vector write into a register that an in-flight MFMA still reads
// as its C operand (first version of this function, output "=v": one launch shape returned f(x) 4e-4 off).  So the output is TIED to
// the register of x0 (and, if x0 lives on, to the compiler's own copy of it): a register whose last writer is an instruction the
// recogniser has seen, and which no matrix instruction reads.
__device__ __forceinline__ unsigned pc_lo_pair(float x0, float x1, unsigned hi) {
    unsigned lo = __float_as_uint(x0);
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(x1));
    return lo;
}
// v[I], v[I + 1] -= the two fp16 halves of `bits`, exactly (v_fma_mix_f32 reads an f16 half as an operand); outputs tied to inputs
template <int I>
__device__ __forceinline__ void pc_sub_halves(f32x4& v, unsigned bits) {
    float x0 = v[I], x1 = v[I + 1];
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(x0) : "v"(bits));
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x1) : "v"(bits));
    v[I] = x0; v[I + 1] = x1;
}
#endif
// (x0, x1) -> packed pairs of the NPARTS pieces (piece k of x0 in the low half of out[k], of x1 in the high half): each piece the
// round-to-nearest value of the running remainder (same arithmetic as split_pair of cc_bf16.h for bf16 pieces)
template <int NPARTS>
__device__ __forceinline__ void pc_split_pair(float x0, float x1, unsigned (&out)[NPARTS]) {
#ifdef UMNN_FWD_PIECE_F16
    if constexpr (NPARTS == 2) {
        out[0] = pc_cvt_pk(x0, x1);
        out[1] = pc_lo_pair(x0, x1, out[0]);
        return;
    }
#endif
#pragma unroll
    for (int k = 0; k < NPARTS; ++k) {
        const unsigned bits = pc_cvt_pk(x0, x1);
        out[k] = bits;
        if (k + 1 < NPARTS) { x0 -= pc_lo_f32(bits); x1 -= pc_hi_f32(bits); }
    }
}

// Stage the pre-split, pre-permuted weight fragments.  img16 index: ((t'*ks + s)*NPARTS + part)*512 + lane*8 + j.
// half_in[l] != 0: the input layer has an odd tile count; ks counts its FULL K-steps (tile pairs) only and the last
// tile follows as half fragments (64 lanes x 4 bf16, for the K = 16 MFMA) at ((t'*NPARTS + part)*256 + lane*4 + j)
// behind the full ones -- the padding tile of a full K-step would cost 7 KB of LDS per layer and part at width 100.
// MERGE (four tiles, two pieces, hidden widths 48..51 -- 13 live registers per lane, tile 3 holds ONE feature per lane
// group): the three cross terms need 3 x (H+1) <= 156 k-slots, so FIVE K-steps per layer carry them instead of six.  K-step 0
// (tiles 0,1) keeps its two fragments [Whi] (used with ahi, then alo) and [Wlo] (with ahi); the two fragments of "K-step 1" become
//   part 0:  k-slots 0-3 Whi of tile 2 (x ahi)  |  4: Whi, 5: Whi, 6: Wlo, 7: 0 of the lane group's tile-3 feature (x ahi, alo, ahi)
//   part 1:  k-slots 0-3 Whi of tile 2 (x alo)  |  4-7 Wlo of tile 2 (x ahi)
// and the kernels pack the B operands to match -- each lane group still supplies its own features only.
// MERGE_FROM: first hidden layer whose image uses this layout (0 = none, 1 = every layer, 2 = all but the first: the wide-first family).
// One (fragment group, lane) per thread and iteration: the lane's 8 (or 4) k-slots of every piece are built in registers and leave
// as 16-byte (8-byte) LDS stores -- one index computation per 8 weights (round 1-2 staged one bf16 per iteration: 7x the
// instructions; at the toy shape's 147 KB of images that was a fifth of a workgroup's time).
template <int NPARTS, int MERGE_FROM = 0>
__device__ __forceinline__ void stage_bf16_images(const MlpDev& m, const int* ks32, const int* off16, const int* half_in,
                                                  unsigned short* lds16, int tid, int nthreads) {
    const int L = m.n_linear - 1;
    for (int l = 1; l < L; ++l) {
        const int Hin = m.width[l], Hout = m.width[l + 1];
        const int ks = ks32[l], to = m.t_out[l + 1];
        const float* __restrict__ W = m.W[l];
        const float* __restrict__ b = m.b[l];
        unsigned short* img = lds16 + off16[l];
        // weight of (output feature fo, input feature fi) incl. the bias column and the constant-one carrier
        auto wv = [&](int fo, int fi) {
            float v = 0.f;
            if (fo < Hout) {
                if (fi < Hin) v = W[fo * Hin + fi];
                else if (fi == Hin) v = b[fo];
            } else if (fo == Hout && fi == Hin) {
                v = 1.f;
            }
            return v;
        };
        // pieces of 8 values -> pk[part][0..3] (two bf16 per dword, k-slot 2e in the low half)
        auto split8 = [&](const float (&v)[8], unsigned (&pk)[NPARTS][4]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned q[NPARTS];
                pc_split_pair<NPARTS>(v[2 * e], v[2 * e + 1], q);
#pragma unroll
                for (int part = 0; part < NPARTS; ++part) pk[part][e] = q[part];
            }
        };
        const int total = to * ks * 64;
        for (int idx = tid; idx < total; idx += nthreads) {
            const int ln = idx & 63, ts = idx >> 6;
            const int s = ts % ks, t = ts / ks;
            const int fo = fout_of(t, ln & 15), g = ln >> 4;
            u32x4* dst = reinterpret_cast<u32x4*>(img + (ts * NPARTS) * 512 + ln * 8);       // fragment `part` at dst[part * 64]
            float v[8];
            unsigned pk[NPARTS][4];
            if (MERGE_FROM > 0 && l >= MERGE_FROM && s == 1) {
                // tile 2's four features and the lane group's tile-3 feature; see the layout note above
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) v[jj] = wv(fo, feat_of(2, jj, g));
                v[4] = wv(fo, feat_of(3, 0, g)); v[5] = 0.f; v[6] = 0.f; v[7] = 0.f;
                split8(v, pk);
                const unsigned h3 = pk[0][2] & 0xffffu, l3 = pk[NPARTS - 1][2] & 0xffffu;
                dst[0] = u32x4{pk[0][0], pk[0][1], h3 | (h3 << 16), l3};
                dst[64] = u32x4{pk[0][0], pk[0][1], pk[NPARTS - 1][0], pk[NPARTS - 1][1]};
                continue;
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) v[jj] = wv(fo, feat_of(2 * s + (jj >> 2), jj & 3, g));
            split8(v, pk);
#pragma unroll
            for (int part = 0; part < NPARTS; ++part) dst[part * 64] = u32x4{pk[part][0], pk[part][1], pk[part][2], pk[part][3]};
        }
        if (half_in[l]) {
            unsigned short* himg = img + to * ks * NPARTS * 512;
            for (int idx = tid; idx < to * 64; idx += nthreads) {
                const int ln = idx & 63, t = idx >> 6;
                const int fo = fout_of(t, ln & 15), g = ln >> 4;
                unsigned q0[NPARTS], q1[NPARTS];
                pc_split_pair<NPARTS>(wv(fo, feat_of(2 * ks, 0, g)), wv(fo, feat_of(2 * ks, 1, g)), q0);
                pc_split_pair<NPARTS>(wv(fo, feat_of(2 * ks, 2, g)), wv(fo, feat_of(2 * ks, 3, g)), q1);
#pragma unroll
                for (int part = 0; part < NPARTS; ++part)
                    *reinterpret_cast<u32x2*>(himg + (t * NPARTS + part) * 256 + ln * 4) = u32x2{q0[part], q1[part]};
            }
        }
    }
}

struct Bf16Plan {
    int ks32[UMNN_MAX_LINEAR];     // K-steps of 32 features when hidden layer l is the input
    int off16[UMNN_MAX_LINEAR];    // ushort offset of image l
    int half_in[UMNN_MAX_LINEAR];  // layer l has an odd tile count and its image ends in half fragments (see staging)
    int scratch_off_floats;        // float offset of the NS-reduction scratch (after the images)
};

struct FwdBf16Args {
    FwdArgs f;
    Bf16Plan pl;
};

// ---- software-pipelined node loop (PIPE variants: four tiles, two bf16 pieces, two point tiles per wave) -----------
// tools/ubench/fill.hip: a wave may issue about two independent VALU instructions in the shadow of each
// v_mfma_f32_16x16x32_bf16 for free, while VALU work issued between the matrix phases costs full price -- and the
// activation / split work of a layer is ~130 VALU instructions against 48 MFMAs.  So the two point tiles of a wave run
// half a layer out of phase: while the matrix pipe multiplies tile A by layer l, the VALU activates, splits and packs
// tile B's layer l-1 output, and vice versa.  Every MFMA is followed by its slice of that work and a scheduling fence.
// The weight fragments of a layer stay in registers for both tiles (LDS is read once per layer, during the second
// tile's section, into registers whose last use has passed).
// The VALU is the scarce unit here (3.9 vector instructions per MFMA, the matrix pipe 56 % busy), so the remainder
// a - bf16(a) of the split is taken on the matrix pipe as well: one extra MFMA per tile with C = a, B = the freshly packed
// leading pieces and A = a 0/-1 selection fragment returns the exact fp32 remainders of 16 features x 16 points, which
// replaces shift / and / subtract on the VALU (5 -> 2 vector instructions per register pair, +3 MFMAs per section).
template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
// EXACT: every hidden layer fills exactly TMAX tiles, so tile / K-step counts are compile-time constants and the
// wave-uniform guards (and the accumulator copies they force at every basic-block boundary) disappear.
// NRL (EXACT only): live registers per lane = ceil((H+1)/4) for the common hidden width H; 0 = all 4*TMAX.
// PIPE (EXACT, TMAX = 4, at least two hidden layers): the node loop is software-pipelined, see pipe_layer.
// INV (plain loop, P = 1): the sampling direction.  A tile is one sample of one flow dimension j, its lanes p = 0..9 are the
// ten candidates x = left + p/9 (right - left) of the reference's bracket search (UMNNMAF.invert, UMNNMAF.py:182-232: [-50, 50]
// to start with, `iter` rounds, the new bracket is the pair of candidates around the one whose image is closest to the
// target); every round integrates all ten candidates with the node loop below, the search itself is a 16-lane butterfly.
// The hoisted first-layer term depends on the sample only and is computed once for all rounds.
// WPB: waves per workgroup.  Eight for the shapes whose weight images leave room for ONE workgroup per CU (uniform 6..8-tile nets, deep
// 5-tile ones: 100-wide toy / MonotonicNN integrands stage 98-147 KB): the second wave of every SIMD then comes from the same
// workgroup and shares its images, instead of the SIMD running one wave with nothing to overlap its vector phases with.
template <int TMAX, int NPARTS, int P, bool EXACT, int NRL, bool PIPE = false, bool INV = false, int TREST = 0, int WPB = UMNN_WAVES_PER_BLOCK>
__global__ __launch_bounds__(64 * WPB) void cc_fwd_bf16_kernel(const FwdBf16Args args) {
    static_assert(TREST == 0 || (EXACT && !PIPE && TREST < TMAX && (TREST & 1) == 0), "wide-first-layer variants: exact, plain loop");
    static_assert(!INV || (P == 1 && !PIPE), "inversion variants: plain loop, one tile per wave");
    constexpr int KSM = TMAX / 2;
    // live registers per lane: of every layer, or (TREST > 0, wide-first family) all 4 * TMAX of layer 1 and NRL of the others
    constexpr int NLIVE = (NRL > 0 && TREST == 0) ? NRL : 4 * TMAX;
    constexpr int NREST = TREST > 0 ? (NRL > 0 ? NRL : 4 * TREST) : NLIVE;
    constexpr bool MERGE = EXACT && TMAX == 4 && NRL == 13 && NPARTS == 2 && TREST == 0;      // five K-steps per layer (see staging)
    constexpr bool MERGE_REST = EXACT && TREST == 4 && NRL == 13 && NPARTS == 2;               // ... per layer from layer 2 on
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const FwdArgs& a = args.f;
    const MlpDev& m = a.m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, p = lane & 15;
    const int L = m.n_linear - 1;
    const int H1 = m.width[1], HL = m.width[L];
    const int E = a.E, d = a.d, n = a.n;
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    unsigned short* lds16 = reinterpret_cast<unsigned short*>(lds);
    // queued behind an fp16-piece launch as its overflow fallback: nothing to do unless that launch raised the flag (cc_fwd_shared.h)
    if (a.ovf_mode == 2 && *a.ovf_flag < a.ovf_gen) return;

    stage_bf16_images<NPARTS, MERGE ? 1 : MERGE_REST ? 2 : 0>(m, args.pl.ks32, args.pl.off16, args.pl.half_in, lds16, tid, 64 * WPB);
    __syncthreads();

    const int ns = a.ns;
    const int sub = wid / ns, part = wid % ns;
    const unsigned gpb = WPB / ns;
    const unsigned grp = xcd_remap(blockIdx.x, gridDim.x) * gpb + sub;
    bool live = grp < a.ngroups;
    if (a.ovf_mode == 2 && live) {                                                                    // ... and then only the deferred groups
        if constexpr (INV) { const float v = a.inv_x[(long long)grp * d + a.inv_j]; live = v != v; }   // (tile = sample: its slot of x_inv[:, j])
        else live = fwd_group_marked<P>(a, grp, p);
    }
    const int k_lo = (int)(((long long)part * (n + 1)) / ns);
    const int k_hi = (int)(((long long)(part + 1) * (n + 1)) / ns);

    float Facc[P], fxv[P], fx0v[P], xv[P], x0v[P], dxv[P];
    bool ok[P];
    long long qv[P];
#pragma unroll
    for (int pt = 0; pt < P; ++pt) { Facc[pt] = 0.f; fxv[pt] = 0.f; fx0v[pt] = 0.f; ok[pt] = false; qv[pt] = 0; dxv[pt] = 0.f; }

    if (live) {
        IoView hb[P];
#pragma unroll
        for (int pt = 0; pt < P; ++pt) {
            if constexpr (INV) {
                const long long b = (long long)grp * P + pt;             // the tile's sample
                ok[pt] = b < a.NI;
                qv[pt] = ok[pt] ? b : a.NI - 1;
                xv[pt] = 0.f; x0v[pt] = 0.f; dxv[pt] = 0.f;              // set per round
                hb[pt] = IoView{a.h, a.h_bf16} + (qv[pt] * ((long long)E * d) + a.inv_j);
            } else {
                const long long q = ((long long)grp * P + pt) * 16 + p;
                ok[pt] = q < a.NI;
                const long long qq = ok[pt] ? q : a.NI - 1;
                qv[pt] = qq;
                xv[pt] = io_ld(a.x, qq, a.x_bf16);
                x0v[pt] = a.x0 ? io_ld(a.x0, qq, a.x_bf16) : 0.f;
                dxv[pt] = xv[pt] - x0v[pt];
                const long long bi = qq / d;
                hb[pt] = IoView{a.h, a.h_bf16} + (bi * ((long long)E * d) + (qq - bi * d));
            }
        }

        // per-lane constants and the hoisted first-layer term (fp32 MFMA, natural row order)
        float w1x[TMAX][4], wout[TMAX][4];
        f32x4 c[P][TMAX];
        {
            const float* __restrict__ W0 = m.W[0];
            const float* __restrict__ b0 = m.b[0];
            const float* __restrict__ WL = m.W[L];
            const float bL = m.b[L][0];
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                f32x4 init;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = feat_of(t, r, g);
                    w1x[t][r] = f < H1 ? W0[f * (1 + E)] : 0.f;
                    wout[t][r] = f < HL ? WL[f] : (f == HL ? bL : 0.f);
                    init[r] = f < H1 ? b0[f] : (f == H1 ? 1.f : 0.f);
                }
#pragma unroll
                for (int pt = 0; pt < P; ++pt) c[pt][t] = init;
            }
            const int t1 = EXACT ? TMAX : m.t_out[1];
            // eight K-steps (32 embedding entries) at a time: all their h and W1 loads are issued before the first
            // MFMA needs one -- one memory round trip per chunk instead of one per step (h comes from HBM)
            const int nse = (E + 3) / 4;
            for (int se0 = 0; se0 < nse; se0 += 8) {
                float hv[P][8], Av[TMAX][8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = 4 * (se0 + j) + g;
                    const bool in = se0 + j < nse && e < E;
#pragma unroll
                    for (int pt = 0; pt < P; ++pt) hv[pt][j] = in ? hb[pt][(long long)e * d] : 0.f;
#pragma unroll
                    for (int t = 0; t < TMAX; ++t) {
                        const int fo = fout_of(t, p);
                        Av[t][j] = (in && (EXACT || t < t1) && fo < H1) ? W0[fo * (1 + E) + 1 + e] : 0.f;
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (se0 + j < nse) {
#pragma unroll
                        for (int t = 0; t < TMAX; ++t)
                            if (EXACT || t < t1) {
#pragma unroll
                                for (int pt = 0; pt < P; ++pt) c[pt][t] = mfma16(Av[t][j], hv[pt][j], c[pt][t]);
                            }
                    }
                }
            }
        }

        if constexpr (PIPE) {
            static_assert(TMAX == 4 && EXACT && NPARTS == 2 && P == 2, "pipelined loop: 4 tiles, 2 pieces, 2 point tiles");
            constexpr int NSLOT = MERGE ? 20 : 24;                  // MFMAs of one point tile in one layer
            constexpr int NFULL = NLIVE / 4;                        // tiles with all four registers live
            static_assert(MERGE == (NFULL == 3), "13 live registers <-> merged K-steps");
            static_assert(NLIVE % 4 == 0 || (NLIVE % 4 == 1 && NFULL == 3), "pipelined loop: 13 or 16 live registers");
            using Slots = std::make_integer_sequence<int, NSLOT>;
            u32x4 wf[4][2][2];                                      // [tile][K-step][piece] of the layer in flight
            u32x4 bf[2][2][2];                                      // [point tile][K-step][piece]
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) bf[pt][ks][k2] = u32x4{0u, 0u, 0u, 0u};
            // 0/-1 selection fragments: row rho of an output tile picks k-slot 8*(rho>>2) + (rho&3) (even tile of the
            // K-step) or + 4 (odd tile) -- the slot in which lane group rho>>2 packed that very feature
            u32x4 sel[2];
            {
                const unsigned rho = lane & 15, slot = rho & 3;
                const unsigned v = ((unsigned)(lane >> 4) == (rho >> 2)) ? (PC_MINUS_ONE << (16 * (slot & 1))) : 0u;   // -1 as a piece
                sel[0] = u32x4{slot < 2 ? v : 0u, slot < 2 ? 0u : v, 0u, 0u};
                sel[1] = u32x4{0u, 0u, slot < 2 ? v : 0u, slot < 2 ? 0u : v};
            }
            float rem_a = 0.f;                                      // the single live register of tile 3 (13-register shape)
            unsigned rem_hi = 0u;
            // (every image of this shape is 16 fragments: off16[l] = (l - 1) * 8192 -- computed, not fetched from the kernel
            // arguments: a scalar load per layer and the wait behind it)
            auto frag = [&](int l, int t, int ks, int k2) {
                return *reinterpret_cast<const u32x4*>(lds16 + (l - 1) * (16 * 512) + lane * 8 + ((t * 2 + ks) * 2 + k2) * 512);
            };
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) wf[t][ks][k2] = frag(1, t, ks, k2);
            // slot i of a section: K-step i/12, cross term (i/4)%3 = (W piece, activation piece) (0,0),(0,1),(1,0), tile i%4
            // MERGE: step i/4 of five -- (tiles 0,1) x the same three terms, then the two merged K-steps [ks=1][0], [ks=1][1]
            auto mfma_slot = [&](auto ic, const u32x4 (&bfin)[2][2], f32x4 (&acc)[4]) {
                constexpr int i = decltype(ic)::value;
                constexpr int t = i % 4;
                constexpr int ks = MERGE ? (i / 4 >= 3) : i / 12;
                constexpr int term = MERGE ? (i / 4) % 3 : (i / 4) % 3;
                constexpr int wa = (MERGE && ks == 1) ? i / 4 - 3 : (term == 2 ? 1 : 0);
                constexpr int ba = (MERGE && ks == 1) ? i / 4 - 3 : (term == 1 ? 1 : 0);
                if constexpr (i < 4) acc[t] = pc_mfma(wf[t][ks][wa], bfin[ks][ba], f32x4{0.f, 0.f, 0.f, 0.f});
                else acc[t] = pc_mfma(wf[t][ks][wa], bfin[ks][ba], acc[t]);
            };
            // in the second tile's section: once a fragment has been used for the last time, fetch the same
            // fragment of the layer that runs next into its registers
            auto reload_slot = [&](auto ic, int lnext) {
                constexpr int i = decltype(ic)::value;
                if constexpr (MERGE) {
                    constexpr int step = i / 4, t = i % 4;            // last uses: step 1, 2, 3, 4
                    if constexpr (step >= 1) wf[t][step >= 3][(step - 1) & 1] = frag(lnext, t, step >= 3, (step - 1) & 1);
                } else {
                    constexpr int ks = i / 12, term = (i / 4) % 3, t = i % 4;
                    if constexpr (term == 1) wf[t][ks][0] = frag(lnext, t, ks, 0);
                    if constexpr (term == 2) wf[t][ks][1] = frag(lnext, t, ks, 1);
                }
            };
            // The packing work of one point tile, in place on its raw pre-activations z[4], one slice per MFMA slot so that
            // (almost) every slice fits the two-instruction issue shadow of the slot's MFMA:
            //   act(t,r)  register r of tile t:  z <- LeakyReLU(z)   (FIRST: z = w1x*t_k + c first)         2-3 VALU
            //   hi(t)     leading pieces of tile t packed into bf[t/2][0], then  z[t] <- z[t] - bf16(z[t])  on the matrix
            //             pipe                                                                       2 VALU + 1 MFMA
            //   lo(t)     second pieces packed into bf[t/2][1] (two slots after hi(t): the MFMA has landed)    2 VALU
            // slots   tile 0: act 0-3, hi 4, lo 10     tile 1: act 5-8, hi 9, lo 16     tile 2: act 11-14, hi 15, lo 20
            //         tile 3: act 17-20, hi 21, lo 23  -- or (MERGE: 20 slots), with one live register, split on the VALU in
            //         slots 17, 18 and tile 2's lo in 19
            auto pack_slot = [&](auto ic, auto first, f32x4 (&z)[4], u32x4 (&bfout)[2][2], float tkv, const f32x4 (&cv)[TMAX]) {
                constexpr int i = decltype(ic)::value;
                constexpr int MODE = decltype(first)::value;      // 0: raw MFMA output, 1: layer 1 (fma first), 2: already activated
                constexpr bool FIRST = MODE == 1, PRE = MODE == 2;
                // slot plans: both cvt pairs of a K-step's operand quad are in place BEFORE its first remainder MFMA reads it (a quad
                // that is read between two partial updates costs register copies: 27 of 40 per node in the first version).
                //   20 slots (MERGE)  act 0-3 (tile 0), 4-7 (tile 1), 12-15 (tile 2); cvt hi of tiles 0, 1 in slot 8, their remainder
                //                     MFMAs in 9; tile 3's single register in 10, 11; cvt hi + remainder MFMA of tile 2 in 16;
                //                     lo of tiles 0, 1, 2 in 17, 18, 19
                //   24 slots          act 0-3, 4-7, 10-13 (tile 2), 14-17 (tile 3); cvt hi of tiles 0, 1 in 8 / of tiles 2, 3 in 18,
                //                     remainder MFMAs in 9 / 19; lo of tiles 0..3 in 20, 21, 22, 23
                constexpr int t_act = MERGE ? (i <= 3 ? 0 : (i >= 4 && i <= 7) ? 1 : (i >= 12 && i <= 15) ? 2 : -1)
                                            : (i <= 3 ? 0 : (i >= 4 && i <= 7) ? 1 : (i >= 10 && i <= 13) ? 2 : (i >= 14 && i <= 17) ? 3 : -1);
                constexpr int r_act = MERGE ? (t_act == 0 ? i : t_act == 1 ? i - 4 : i - 12)
                                            : (t_act == 0 ? i : t_act == 1 ? i - 4 : t_act == 2 ? i - 10 : i - 14);
                constexpr int t_hi = MERGE ? (i == 16 ? 2 : -1) : -1;
                constexpr int t_lo = MERGE ? (i == 17 ? 0 : i == 18 ? 1 : i == 19 ? 2 : -1) : (i >= 20 ? i - 20 : -1);
                constexpr int T3A = MERGE ? 10 : -1, T3B = MERGE ? 11 : -1;
                constexpr int QCVT = (!MERGE && i == 18) ? 1 : (i == 8 ? 0 : -1);        // quad (tiles 2q, 2q+1): cvt hi of both tiles
                constexpr int QSEL = (!MERGE && i == 19) ? 1 : (i == 9 ? 0 : -1);        // ... and their remainder MFMAs
                if constexpr (t_act >= 0 && !PRE) {
                    if constexpr (FIRST) z[t_act][r_act] = fmaf(w1x[t_act][r_act], tkv, cv[t_act][r_act]);
                    z[t_act][r_act] = hidden_act_f(z[t_act][r_act], slope);
                }
                if constexpr (QCVT >= 0) {                        // leading pieces of tiles 2q and 2q+1: the whole quad at once
                    constexpr int ta = 2 * QCVT, tb = 2 * QCVT + 1;
                    const unsigned a0 = pc_cvt_pk(z[ta][0], z[ta][1]);
                    const unsigned a1 = pc_cvt_pk(z[ta][2], z[ta][3]);
                    const unsigned b0 = pc_cvt_pk(z[tb][0], z[tb][1]);
                    const unsigned b1 = pc_cvt_pk(z[tb][2], z[tb][3]);
                    bfout[QCVT][0] = u32x4{a0, a1,
                                           b0, b1};
                }
#ifdef UMNN_FWD_PIECE_F16
                // fp16 pieces (round 5): the remainders on the VALU after all -- v_fma_mix_f32 subtracts a packed f16 half straight from
                // the fp32 value, one instruction per register, spread over the slots behind the quad's conversions.  Same-box A/B at
                // C3 against the remainder MFMAs below: 2.470 -> 2.427 ms, 9 of 69 matrix instructions per tile-node fewer, shader
                // clock 2141 -> 2170 MHz under the same power -- the kernel is bound by matrix-pipe cycles and package power, not by
                // vector issue (the bf16 build keeps the MFMA form: its VALU remainder is three instructions per register).
                constexpr int QV = (!MERGE && i >= 19 && i <= 21) ? 1 : ((i >= 9 && i <= 11) ? 0 : -1);
                if constexpr (QV >= 0) {
                    constexpr int k = i - (QV ? 19 : 9);               // 0: first tile of the quad; 1, 2: halves of the second
                    if constexpr (k == 0) { pc_sub_halves<0>(z[2 * QV], bfout[QV][0][0]); pc_sub_halves<2>(z[2 * QV], bfout[QV][0][1]); }
                    if constexpr (k == 1) pc_sub_halves<0>(z[2 * QV + 1], bfout[QV][0][2]);
                    if constexpr (k == 2) pc_sub_halves<2>(z[2 * QV + 1], bfout[QV][0][3]);
                }
#else
                if constexpr (QSEL >= 0) {
                    z[2 * QSEL] = pc_mfma(sel[0], bfout[QSEL][0], z[2 * QSEL]);          // exact remainders
                    z[2 * QSEL + 1] = pc_mfma(sel[1], bfout[QSEL][0], z[2 * QSEL + 1]);
                }
#endif
                if constexpr (t_hi >= 0) {
                    const unsigned h0 = pc_cvt_pk(z[t_hi][0], z[t_hi][1]);
                    const unsigned h1 = pc_cvt_pk(z[t_hi][2], z[t_hi][3]);
                    bfout[t_hi / 2][0][2 * (t_hi % 2)] = h0;
                    bfout[t_hi / 2][0][2 * (t_hi % 2) + 1] = h1;
                    if constexpr (MERGE && t_hi == 2) {           // tile 2's leading pieces also close the last K-step
                        bfout[1][1][2] = h0;
                        bfout[1][1][3] = h1;
                    }
#ifdef UMNN_FWD_PIECE_F16
                    pc_sub_halves<0>(z[t_hi], h0); pc_sub_halves<2>(z[t_hi], h1);
#else
                    z[t_hi] = pc_mfma(sel[t_hi % 2], bfout[t_hi / 2][0], z[t_hi]);          // exact remainders
#endif
                }
                if constexpr (t_lo >= 0) {
                    const unsigned l0 = pc_cvt_pk(z[t_lo][0], z[t_lo][1]);
                    const unsigned l1 = pc_cvt_pk(z[t_lo][2], z[t_lo][3]);
                    bfout[t_lo / 2][1][2 * (t_lo % 2)] = l0;
                    bfout[t_lo / 2][1][2 * (t_lo % 2) + 1] = l1;
                }
                if constexpr (i == T3A) {                        // tile 3 has one live register: split it on the VALU
                    if constexpr (FIRST) z[3][0] = fmaf(w1x[3][0], tkv, cv[3][0]);
                    rem_a = PRE ? z[3][0] : hidden_act_f(z[3][0], slope);
                    const unsigned h = pc_cvt_pk(rem_a, 0.f);
                    rem_hi = h;
                    bfout[1][0][3] = rem_hi;                     // k-slots 6,7: (hi, 0)
                }
                if constexpr (i == T3B) {
#ifdef UMNN_FWD_PIECE_F16
                    float rr = rem_a;
                    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(rr) : "v"(rem_hi));
                    const unsigned l = pc_cvt_pk(rr, 0.f);
#else
                    const unsigned l = pc_cvt_pk(rem_a - pc_lo_f32(rem_hi), 0.f);
#endif
                    bfout[1][0][2] = rem_hi | (l << 16);      // k-slots 4,5: (hi, lo)
                }
            };

            // Layer 1 of the FIRST tile (fma + LeakyReLU per live register) is computed one node ahead, in the shadow of the
            // previous node's last matrix section (whose slots 11..23 carry no other vector work): at the head of a node, where
            // nothing runs on the matrix pipe yet, only the bf16 packing of those values is left.
            constexpr int PRE0 = NSLOT - NLIVE;                          // first slot of the last section that prepares the next node
            f32x4 znext[4];
            {
                const float u0 = a.ccs[k_lo] + 1.f;
                const float t0 = k_lo == 0 ? xv[0] : __fadd_rn(x0v[0], __fmul_rn(dxv[0], u0) * 0.5f);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        znext[t][r] = 4 * t + r < NLIVE ? hidden_act_f(fmaf(w1x[t][r], t0, c[0][t][r]), slope) : 0.f;
            }
            float cs_cur = a.ccs[k_lo], cw_cur = a.ccw[k_lo];       // node abscissa / weight: fetched one node ahead
            for (int k = k_lo; k < k_hi; ++k) {
                const int kn = k + 1 <= n ? k + 1 : n;
                const float cs_next = a.ccs[kn], cw_next = a.ccw[kn];
                const float u = cs_cur + 1.f;
                const float wk = cw_cur;
                cs_cur = cs_next; cw_cur = cw_next;
                float tk[2];
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) tk[pt] = k == 0 ? xv[pt] : __fadd_rn(x0v[pt], __fmul_rn(dxv[pt], u) * 0.5f);
                f32x4 acc0[4], acc1[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc0[t] = znext[t];
                constexpr std::true_type kFirst{};
                constexpr std::false_type kLater{};
                constexpr std::integral_constant<int, 2> kPre{};
                // first tile: its layer 1 is already there, only the packing is left -- nothing to hide behind yet
                static_for(Slots{}, [&](auto ic) { pack_slot(ic, kPre, acc0, bf[0], tk[0], c[0]); });
                __builtin_amdgcn_sched_barrier(0);
                // section A of layer 1: first tile on the matrix pipe, second tile's layer 1 + packing on the VALU
                static_for(Slots{}, [&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    mfma_slot(ic, bf[0], acc0);
                    pack_slot(ic, kFirst, acc1, bf[1], tk[1], c[1]);
                    __builtin_amdgcn_sched_barrier(0);
                });
                for (int l = 1; l + 1 < L; ++l) {
                    // section B of layer l: second tile on the matrix pipe, first tile's output packed for layer l+1
                    static_for(Slots{}, [&](auto ic) {
                        mfma_slot(ic, bf[1], acc1);
                        pack_slot(ic, kLater, acc0, bf[0], 0.f, c[0]);
                        reload_slot(ic, l + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    // section A of layer l+1
                    static_for(Slots{}, [&](auto ic) {
                        mfma_slot(ic, bf[0], acc0);
                        pack_slot(ic, kLater, acc1, bf[1], 0.f, c[1]);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                }
                // section B of the last hidden layer: the first tile's output goes into the output dot product
                const float tkn0 = __fadd_rn(x0v[0], __fmul_rn(dxv[0], cs_next + 1.f) * 0.5f);     // (k + 1 >= 1: never node 0)
                float sd0 = 0.f, sd1 = 0.f;
                static_for(Slots{}, [&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    mfma_slot(ic, bf[1], acc1);
                    if constexpr (i < NLIVE) sd0 = fmaf(wout[i / 4][i % 4], hidden_act_f(acc0[i / 4][i % 4], slope), sd0);
                    if constexpr (i >= PRE0) {
                        constexpr int e = i - PRE0, t = e / 4, r = e % 4;
                        znext[t][r] = hidden_act_f(fmaf(w1x[t][r], tkn0, c[0][t][r]), slope);
                    }
                    reload_slot(ic, 1);
                    __builtin_amdgcn_sched_barrier(0);
                });
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * t + r < NLIVE) sd1 = fmaf(wout[t][r], hidden_act_f(acc1[t][r], slope), sd1);
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    const float sr = group_allreduce(pt == 0 ? sd0 : sd1);
                    const float f = pc_out_act(sr, m.out_act);
                    Facc[pt] = fmaf(wk, maybe_inverse(f, a.inv_f), Facc[pt]);
                    if (k == 0) fxv[pt] = f;
                    if (k == n) fx0v[pt] = f;
                }
            }
        } else {
        // ---- bracket search state (INV): one sample per tile, candidate p on lane p ----
        float br_left = -50.f, br_right = 50.f, br_best = 0.f, inv_target = 0.f, inv_off = 0.f, inv_scale = 1.f, frac = 1.f;
        if constexpr (INV) {
            frac = p < 10 ? (float)((double)p / 9.0) : 1.f;            // x_range of the reference: k * (1/9) in double, cast

            inv_target = a.inv_z[qv[0] * d + a.inv_j];
            inv_off = hb[0][0];                                        // embedding row 0 of dimension j: the offset
            inv_scale = __expf(a.scaling[a.inv_j]);
        }
        const int rounds = INV ? a.inv_iters : 1;
        bool inv_bad = false;            // (fp16 pieces: some candidate integral of some round was not finite -- an overflowed piece)
        for (int round = 0; round < rounds; ++round) {
        if constexpr (INV) {
            xv[0] = __fadd_rn(__fmul_rn(frac, br_right - br_left), br_left);      // x_range * (right - left) + left
            dxv[0] = xv[0];
            Facc[0] = 0.f;
        }
        for (int k = k_lo; k < k_hi; ++k) {
            const float u = a.ccs[k] + 1.f;
            const float wk = a.ccw[k];
            f32x4 act[P][TMAX];
#pragma unroll
            for (int pt = 0; pt < P; ++pt) {
                const float tk = k == 0 ? xv[pt] : __fadd_rn(x0v[pt], __fmul_rn(dxv[pt], u) * 0.5f);
#pragma unroll
                for (int t = 0; t < TMAX; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        act[pt][t][r] = 4 * t + r < NLIVE ? hidden_act_f(fmaf(w1x[t][r], tk, c[pt][t][r]), slope) : 0.f;
            }

            auto layer = [&](auto wide_c, int l) {
                // TREST > 0 (first hidden layer wider than the others, e.g. 100-50-50-50-50): layer 1 contracts over TMAX tiles, every
                // later layer over TREST; every layer produces TREST output tiles.  Compile-time per instantiation of this lambda.
                constexpr bool WIDE = decltype(wide_c)::value;
                constexpr int KT = (TREST > 0 && !WIDE) ? TREST : TMAX;
                constexpr int OT = TREST > 0 ? TREST : TMAX;
                constexpr int KSL = KT / 2;
                constexpr bool HALFL = EXACT && (KT & 1);
                constexpr int NIN = (TREST > 0 && !WIDE) ? NREST : NLIVE;      // live registers of the layer's input / output
                constexpr int NOUT = TREST > 0 ? NREST : NLIVE;
                constexpr bool MERGEL = MERGE || (MERGE_REST && !WIDE);
                const int ks = EXACT ? KSL : args.pl.ks32[l], to = EXACT ? OT : m.t_out[l + 1];
                // (uniform exact shapes: the image offset is (l - 1) x a compile-time stride -- no scalar load from the arguments)
                constexpr int IMG_STRIDE = TMAX * (KSM * NPARTS * 512 + (TMAX & 1) * NPARTS * 256);
                const int off_l = (EXACT && TREST == 0) ? (l - 1) * IMG_STRIDE : args.pl.off16[l];
                const unsigned short* img = lds16 + off_l + lane * 8;
                // split + pack the activations into B fragments: K-step s <- tiles 2s, 2s+1
                u32x4 bf[P][KSL > 0 ? KSL : 1][NPARTS];
#pragma unroll
                for (int pt = 0; pt < P; ++pt)
#pragma unroll
                    for (int s = 0; s < KSL; ++s) {
                        unsigned q0[NPARTS], q1[NPARTS], q2[NPARTS], q3[NPARTS];
#pragma unroll
                        for (int k2 = 0; k2 < NPARTS; ++k2) q0[k2] = q1[k2] = q2[k2] = q3[k2] = 0u;
                        if (8 * s + 0 < NIN) pc_split_pair<NPARTS>(act[pt][2 * s][0], act[pt][2 * s][1], q0);
                        if (8 * s + 2 < NIN) pc_split_pair<NPARTS>(act[pt][2 * s][2], act[pt][2 * s][3], q1);
                        if (8 * s + 4 < NIN) pc_split_pair<NPARTS>(act[pt][2 * s + 1][0], act[pt][2 * s + 1][1], q2);
                        if (8 * s + 6 < NIN) pc_split_pair<NPARTS>(act[pt][2 * s + 1][2], act[pt][2 * s + 1][3], q3);
#pragma unroll
                        for (int k2 = 0; k2 < NPARTS; ++k2) bf[pt][s][k2] = u32x4{q0[k2], q1[k2], q2[k2], q3[k2]};
                        if constexpr (MERGEL) {
                            if (s == 1) {        // (tile 2 hi | tile-3 feature: hi, lo, hi, 0) and (tile 2 lo | tile 2 hi)
                                bf[pt][1][0] = u32x4{q0[0], q1[0], q2[0] | (q2[NPARTS - 1] << 16), q2[0]};
                                bf[pt][1][NPARTS - 1] = u32x4{q0[NPARTS - 1], q1[NPARTS - 1], q0[0], q1[0]};
                            }
                        }
                    }
                // odd tile count (EXACT only): the last tile is a K = 16 step of its own
                u32x2 hb[P][NPARTS];
                if constexpr (HALFL) {
#pragma unroll
                    for (int pt = 0; pt < P; ++pt) {
                        unsigned q0[NPARTS], q1[NPARTS];
#pragma unroll
                        for (int k2 = 0; k2 < NPARTS; ++k2) q0[k2] = q1[k2] = 0u;
                        if (4 * (KT - 1) + 0 < NIN) pc_split_pair<NPARTS>(act[pt][KT - 1][0], act[pt][KT - 1][1], q0);
                        if (4 * (KT - 1) + 2 < NIN) pc_split_pair<NPARTS>(act[pt][KT - 1][2], act[pt][KT - 1][3], q1);
#pragma unroll
                        for (int k2 = 0; k2 < NPARTS; ++k2) hb[pt][k2] = u32x2{q0[k2], q1[k2]};
                    }
                }
                f32x4 acc[P][TMAX];
#pragma unroll
                for (int pt = 0; pt < P; ++pt)
#pragma unroll
                    for (int t = 0; t < TMAX; ++t) acc[pt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KSL; ++s) {
                    if (EXACT || s < ks) {
                        u32x4 wf[TMAX][NPARTS];
#pragma unroll
                        for (int t = 0; t < OT; ++t)
                            if (EXACT || t < to) {
#pragma unroll
                                for (int k2 = 0; k2 < NPARTS; ++k2)
                                    wf[t][k2] = *reinterpret_cast<const u32x4*>(img + ((t * ks + s) * NPARTS + k2) * 512);
                            }
                        // cross terms in order of decreasing magnitude; consecutive MFMAs hit different accumulators
#pragma unroll
                        for (int wa = 0; wa < NPARTS; ++wa)
#pragma unroll
                            for (int ba = 0; ba < NPARTS; ++ba) {
                                if (MERGEL && s == 1) { if (wa != ba) continue; }      // merged K-steps: fragment k x operand k
                                else if (wa + ba >= NPARTS) continue;      // 2 parts: hh,hl,lh ; 3 parts: + h l2, l2 h, l l
#pragma unroll
                                for (int t = 0; t < OT; ++t)
                                    if (EXACT || t < to) {
#pragma unroll
                                        for (int pt = 0; pt < P; ++pt)
                                            acc[pt][t] = pc_mfma(wf[t][wa], bf[pt][s][ba], acc[pt][t]);
                                    }
                            }
                    }
                }
                if constexpr (HALFL) {
                    const unsigned short* himg = lds16 + off_l + OT * KSL * NPARTS * 512 + lane * 4;
                    u32x2 wh[TMAX][NPARTS];
#pragma unroll
                    for (int t = 0; t < OT; ++t)
#pragma unroll
                        for (int k2 = 0; k2 < NPARTS; ++k2)
                            wh[t][k2] = *reinterpret_cast<const u32x2*>(himg + (t * NPARTS + k2) * 256);
#pragma unroll
                    for (int wa = 0; wa < NPARTS; ++wa)
#pragma unroll
                        for (int ba = 0; ba < NPARTS; ++ba) {
                            if (wa + ba >= NPARTS) continue;
#pragma unroll
                            for (int t = 0; t < OT; ++t)
#pragma unroll
                                for (int pt = 0; pt < P; ++pt)
                                    acc[pt][t] = pc_mfma_k16(wh[t][wa], hb[pt][ba], acc[pt][t]);
                        }
                }
                if constexpr (TREST > 0 && WIDE && !INV && P == 1) {
                    if (a.z2_save) {          // (see FwdArgs::z2_save: P = 1, so the tile index is the group index)
                        float* zs = a.z2_save + ((size_t)grp * (size_t)(n + 1) + k) * a.z2_nl2 * 64 + lane;
#pragma unroll
                        for (int t = 0; t < OT; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (4 * t + r < a.z2_nl2) zs[(4 * t + r) * 64] = acc[0][t][r];
                    }
                }
#pragma unroll
                for (int pt = 0; pt < P; ++pt)
#pragma unroll
                    for (int t = 0; t < TMAX; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            act[pt][t][r] = (t < OT && (EXACT || t < to) && 4 * t + r < NOUT) ? hidden_act_f(acc[pt][t][r], slope) : 0.f;
            };
            if constexpr (TREST > 0) {
                layer(std::true_type{}, 1);
                for (int l = 2; l < L; ++l) layer(std::false_type{}, l);
            } else {
                for (int l = 1; l < L; ++l) layer(std::true_type{}, l);
            }

#pragma unroll
            for (int pt = 0; pt < P; ++pt) {
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < TMAX; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * t + r < (TREST > 0 ? NREST : NLIVE)) s = fmaf(wout[t][r], act[pt][t][r], s);
                s = group_allreduce(s);
                const float f = pc_out_act(s, m.out_act);
                Facc[pt] = fmaf(wk, maybe_inverse(f, a.inv_f), Facc[pt]);
                if (k == 0) fxv[pt] = f;
                if (k == n) fx0v[pt] = f;
            }
        }
        if constexpr (INV) {
            if (ns > 1) {
                // small batches (cc_invert.hip): the node range of this sample's ten integrals is split over ALL waves of the workgroup
                // (ns = waves per workgroup, one sample per workgroup: every wave of a live workgroup gets here, the barriers are
                // uniform); partial sums meet in LDS once per round, summed in a fixed order; every wave then runs the same search
                float* red = lds + m.lds_off[L];
                if (g == 0) red[wid * 16 + p] = Facc[0];
                __syncthreads();
                float tot = 0.f;
                for (int jj = 0; jj < ns; ++jj) tot += red[(sub * ns + jj) * 16 + p];
                __syncthreads();
                Facc[0] = tot;
            }
            if constexpr (PC_F16) inv_bad = inv_bad || (p < 10 && !(__builtin_fabsf(Facc[0]) < __builtin_inff()));
            // image of every candidate, then argmin_p |z_est - target| over the ten candidate lanes (ties: lower p)
            const float z_est = inv_scale * (inv_off + Facc[0] * dxv[0] * 0.5f);
            float dist = p < 10 ? fabsf(z_est - inv_target) : __builtin_inff();
            float zm = z_est;
            int m = p;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                const float d2 = __shfl_xor(dist, o, 16), z2 = __shfl_xor(zm, o, 16);
                const int m2 = __shfl_xor(m, o, 16);
                const bool take = d2 < dist || (d2 == dist && m2 < m);
                dist = take ? d2 : dist; zm = take ? z2 : zm; m = take ? m2 : m;
            }
            const float span = br_right - br_left;
            const float lo = __fadd_rn(__fmul_rn((float)((double)(m > 0 ? m - 1 : 0) / 9.0), span), br_left);
            const float hi = __fadd_rn(__fmul_rn((float)((double)(m < 9 ? m + 1 : 9) / 9.0), span), br_left);
            br_best = __fadd_rn(__fmul_rn((float)((double)m / 9.0), span), br_left);
            const bool below = zm < inv_target;
            br_left = below ? br_best : lo;
            br_right = below ? hi : br_best;
        }
        }   // rounds
        if constexpr (INV) {
            // overflow protocol (cc_invert.hip): the sample is left to the queued bf16 build, its slot marked with a NaN
            const bool defer = a.ovf_mode == 1 && __any(inv_bad);
            if (ok[0] && lane == 0 && part == 0) {
                a.inv_x[qv[0] * d + a.inv_j] = defer ? __builtin_nanf("") : br_best;
                if (defer) atomicMax(a.ovf_flag, a.ovf_gen);
            }
        }
        }
    }
    if constexpr (!INV) fwd_epilogue<P>(a, lds, Facc, fxv, fx0v, ok, qv, dxv, live, part, ns, wid, g, p);
}

}  // namespace UMNN_FWD_NS
