// Software-pipelined one-pass backward on the bf16 matrix cores (round 3).  Same arithmetic, operands and accumulation
// order as cc_bwd_bf16_kernel (cc_bwd_bf16_kernel.h; reference lines ParallelNeuralIntegral.py:66-94,110-123): dx, dx0 and the
// output layer's gradient are bit-identical, d_h and the hidden layers' d_theta agree to ~1e-7 of their largest entry (the
// compiler contracts the split residual with the product that formed the value differently in the two code shapes).  What
// changes is the node loop: ONE wave now carries TWO independent dependency chains,
//
//     iteration k :   B(k)   = backward sweep of node k        (delta chain through W^T, dW += delta (x) a)
//                     F(k+1) = forward recompute of node k+1   (six-term split GEMMs, activations, output layer)
//
// and every instruction of the loop body sits in an explicit slot: one MFMA, the vector / LDS work that rides behind it, a
// scheduling fence.  The round-2 loop ran F(k) then B(k) -- split -> GEMM -> activation -> split -> ... is one serial chain
// -- and with one wave per SIMD (the 192 dW accumulators need the whole register file) every MFMA -> VALU -> MFMA
// dependency and every LDS fetch was exposed: matrix pipe 55 % busy, 14 % of the wave's time parked on s_waitcnt.
//
// Stage i of an iteration pairs F's layer i (GEMM i -> i+1 of node k+1) with B's layer l = L - i (delta_{l+1} -> delta_l,
// dW_l of node k), in two regions:
//   region G  72 MFMAs: the W^T GEMM (24) and the forward GEMM (48), interleaved in groups that share a fragment set.  The
//             fragments come through register buffers loaded 8..20 slots ahead of their first use (two W buffers, one W^T
//             buffer, 16 registers each); the split of K-step 1's operands (register pairs 4..7) rides behind the first 16
//             slots -- K-step 1's MFMAs start at slot 36.  At the end: a_l^T and the sign piece of a_l are read out of the
//             LDS slot X, THEN F's a_i(k+1) pieces are stored into X (LDS executes a wave's accesses in order), and delta^T
//             is read back from the delta slot.
//   region D  48 MFMAs dW_l += delta_{l+1} (x) a_l (accumulators only).  Behind them: the activations of both chains (one
//             register per slot), the split of K-step 0 of the NEXT stage's operands, the first W set of the next region G;
//             at the last stage instead B's tail (delta_1 -> dc, dW1[:,0]), F's output layer for node k+1 (delta_L, dwo) and
//             layer 1 of node k+2.
// What makes that fit:
//   * a_l no longer lives in registers between F and B (48 registers in the round-2 kernel): its two leading bf16 pieces go
//     to an LDS slot -- the same [piece][point][slot] tile the dW product transposes through with ds_read_b64_tr_b16 -- and
//     the sign of a_l (activation derivative) is read back from there.  F runs one node ahead and produces a_1..a_{L-1} in
//     the order B consumes them backwards, so L-1 slots rotate (stage i frees the slot of a_{L-i}(k) for a_i(k+1)):
//     (L-1) + 1 slots of 4.5 KB per wave.
//   * ONE weight image serves W and W^T.  The forward fragment image (3 pieces, 24 KB per layer) is read with
//     ds_read_b128 by the recompute and with ds_read_b64_tr_b16 by the delta chain: lane (g', p) of a W^T fragment (t, s)
//     wants k-slots j = 4h + e  <->  W[16 (2s+h) + 4e + g'][16t + 4 (p&3) + (p>>2)]; in the forward image those are, for
//     fixed (h, e), 4 consecutive bf16 of the lane (g'_f = p&3, rho_f = 4 g' + e) of fragment (2s+h, t>>1), half t&1 --
//     exactly the 4 x 16 block a transposing read delivers.  72 KB instead of 120 KB of images (144 KB of LDS in all); the
//     16-byte units of a fragment are rotated by 8 for g'_f >= 2 so the transposing reads are 2-way conflicted (inherent:
//     32 lanes read the same 8-byte half of 16 units) instead of 4-way, and the b128 reads stay conflict-free.
//     The constant-one feature (bias column / unit row of the forward image) forms a closed subspace under W^T -- it only
//     ever feeds the constant feature's own delta, whose dW rows and dc entries are discarded at write-out.
// Measured at C3 (8192 x 63, n = 100; profiles/r03): 14.8-15.1 ms (round-2 loop) -> 13.5-13.9 ms; wave cycles per tile-node
// 10.6 k -> 9.6 k, parked on s_waitcnt 14 % -> 10 %, matrix pipe 55 % -> 60 % busy.  EXPERIMENTS.md has the other variants that
// were measured (slots only in the dW regions: 16.2 ms; the two chains in antiphase: 15.1-15.5 ms) and where this one waits (the dW
// regions run at pipe speed, the GEMM regions at 1.2-1.9x their matrix time; starting the four waves apart: no effect) -- the
// timing / experiment switches of those measurements were removed from this file in round 4.
#pragma once
#include <type_traits>
#include <utility>
#include "cc_bwd_bf16_kernel.h"

// 16-byte unit of lane (gq, rho) inside a 1 KB fragment of the shared image
__device__ __forceinline__ int swz_unit(int gq, int rho) { return gq * 16 + ((rho + 8 * (gq >> 1)) & 15); }

template <int NP>
__device__ __forceinline__ void stage_frag_image_swz(const MlpDev& m, int l, unsigned short* img, int tid, int nthreads) {
    const int Hin = m.width[l], Hout = m.width[l + 1];
    const float* __restrict__ W = m.W[l];
    const float* __restrict__ b = m.b[l];
    for (int idx = tid; idx < BT * BKS * 64; idx += nthreads) {         // one (fragment, lane) per iteration: 16 bytes per piece
        const int ln = idx & 63, ts = idx >> 6;
        const int s = ts % BKS, t = ts / BKS;
        const int fo = fout_of(t, ln & 15);
        unsigned short pc[NP][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int fi = feat_of(2 * s + (j >> 2), j & 3, ln >> 4);
            float v = 0.f;
            if (fo < Hout) {
                if (fi < Hin) v = W[fo * Hin + fi];
                else if (fi == Hin) v = b[fo];
            } else if (fo == Hout && fi == Hin) {
                v = 1.f;
            }
#pragma unroll
            for (int part = 0; part < NP; ++part) {
                const unsigned short hb = bf16_rn_bits(v);
                pc[part][j] = hb;
                v -= bf16_bits_to_f32(hb);
            }
        }
        const int unit = swz_unit(ln >> 4, ln & 15);
#pragma unroll
        for (int part = 0; part < NP; ++part) {
            u32x4 q;
#pragma unroll
            for (int w = 0; w < 4; ++w) q[w] = (unsigned)pc[part][2 * w] | ((unsigned)pc[part][2 * w + 1] << 16);
            *reinterpret_cast<u32x4*>(img + (ts * NP + part) * FRAG + unit * 8) = q;
        }
    }
}

// acc[t] = sum over K-steps and cross terms of W^T fragment (t, s, wa) * bd[s][ba]; the fragments are read out of the
// FORWARD image with transposing reads (see header).  imgT = image base + this lane's transposing-read unit offset.
__device__ __forceinline__ void gemm_frags_T(const unsigned short* imgT, const BFrag<NPB>& bd, f32x4 (&acc)[BT]) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < BKS; ++s) {
        u32x4 wf[BT][NPB];
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int k2 = 0; k2 < NPB; ++k2) {
                const unsigned short* lo = imgT + (((2 * s + 0) * BKS + (t >> 1)) * NPF + k2) * FRAG + 4 * (t & 1);
                const unsigned short* hi = imgT + (((2 * s + 1) * BKS + (t >> 1)) * NPF + k2) * FRAG + 4 * (t & 1);
                const u32x2 a = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4v __attribute__((address_space(3)))*)(lo)));
                const u32x2 b = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4v __attribute__((address_space(3)))*)(hi)));
                wf[t][k2] = u32x4{a[0], a[1], b[0], b[1]};
            }
#pragma unroll
        for (int wa = 0; wa < NPB; ++wa)
#pragma unroll
            for (int ba = 0; ba < NPB; ++ba) {
                if (wa + ba >= NPB) continue;
                const bool first = s == 0 && wa == 0 && ba == 0;
#pragma unroll
                for (int t = 0; t < BT; ++t) acc[t] = mfma_bf16(wf[t][wa], bd.v[s][ba], first ? zero : acc[t]);
            }
    }
}

// activation derivative off the sign of the leading bf16 piece, pieces given as the two u32x4 (K-steps) a lane stored
__device__ __forceinline__ float act_grad_q(const u32x4 (&hi)[BKS], int t, int r, float slope) {
    const unsigned u = hi[t >> 1][(t & 1) * 2 + (r >> 1)];
    // a_l > 0  <=>  its leading bf16 piece > 0: the high half as "dword > 0xffff" (signed), the low half as a signed 16-bit
    // compare of the dword's low word -- no shift / mask instruction in front of the compare
    const bool pos = (r & 1) ? ((int)u > 0xffff) : ((short)(u & 0xffffu) > 0);
    return pos ? 1.f : slope;
}

// one rounding stage of a pair split: returns the packed bf16 pair, leaves the residuals in x0 / x1 (same arithmetic, same
// order as split_pair in cc_bf16.h -- the pieces are bit-identical)
__device__ __forceinline__ unsigned split_stage(float& x0, float& x1) {
    const bf16x2 h = __builtin_convertvector(f32x2{x0, x1}, bf16x2);
    const unsigned bits = __builtin_bit_cast(unsigned, h);
    x0 -= __uint_as_float(bits << 16);
    x1 -= __uint_as_float(bits & 0xffff0000u);
    return bits;
}
__device__ __forceinline__ unsigned split_last(float x0, float x1) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
}
template <class F, int... I>
__device__ __forceinline__ void swp_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void swp_static_for(F&& f) { swp_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int LH, int NRL>
__global__ __launch_bounds__(UMNN_BLOCK, 1) void cc_bwd_swp_kernel(const BwdBf16Args args) {
    constexpr int NLIVE = NRL > 0 ? NRL : 4 * BT;
    constexpr int NPAIR = (NLIVE + 1) / 2;             // register pairs that are split / packed (7 or 8)
    constexpr int L = LH;
    constexpr int NG = LH - 1;                         // hidden -> hidden GEMM layers (1..3)
    constexpr int IMG = BT * BKS * NPF * FRAG;         // ushorts per layer image
    constexpr int SLOT = NPB * 16 * TRS;               // ushorts per transpose slot
    constexpr int NDW = 3 * BT * BT;                   // dW MFMAs per layer and node (3 cross terms x 16 tiles)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const BwdArgs& a = args.b;
    const MlpDev& m = a.m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, p = lane & 15;
    const int H1 = m.width[1], HL = m.width[L];
    const int E = a.E, d = a.d, n = a.n;
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    unsigned short* lds16 = reinterpret_cast<unsigned short*>(lds);

    for (int l = 1; l < L; ++l) stage_frag_image_swz<NPF>(m, l, lds16 + (l - 1) * IMG, tid, blockDim.x);
    __syncthreads();
    const unsigned short* fragF = lds16 + swz_unit(g, p) * 8;                              // forward fragments (ds_read_b128)
    const unsigned short* fragT = lds16 + swz_unit(p & 3, 4 * g + (p >> 2)) * 8;           // the same image, transposing reads
    const int slot0 = NG * IMG + wid * (NG + 1) * SLOT;
    int so[NG];                                        // so[j]: slot (ushort offset) holding a_{j+1} of the node B works on
#pragma unroll
    for (int j = 0; j < NG; ++j) so[j] = slot0 + j * SLOT;
    const int sdel = slot0 + NG * SLOT;                // delta slot
    const int own = p * TRS + g * 16;                  // this lane's own k-slots inside a slot tile (+ piece * 16 * TRS + s * 8)

    float w1x[BT][4], wout[BT][4];
    {
        const float* __restrict__ W0 = m.W[0];
        const float* __restrict__ WL = m.W[L];
        const float bL = m.b[L][0];
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = feat_of(t, r, g);
                w1x[t][r] = f < H1 ? W0[f * (1 + E)] : 0.f;
                wout[t][r] = f < HL ? WL[f] : (f == HL ? bL : 0.f);
            }
    }

    f32x4 dW[NG][BT][BT];
#pragma unroll
    for (int j = 0; j < NG; ++j)
#pragma unroll
        for (int to = 0; to < BT; ++to)
#pragma unroll
            for (int ti = 0; ti < BT; ++ti) dW[j][to][ti] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 dW1x[BT], dwo[BT];
#pragma unroll
    for (int t = 0; t < BT; ++t) { dW1x[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dwo[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const unsigned wave_global = blockIdx.x * (blockDim.x >> 6) + wid;
    const unsigned nwaves = gridDim.x * (blockDim.x >> 6);
    const unsigned nsp = a.ns > 1 ? (unsigned)a.ns : 1u;

    for (unsigned item = wave_global; item < a.ngroups * nsp; item += nwaves) {
        const unsigned grp = item / nsp, part = item - grp * nsp;
        const int k_lo = (int)(((long long)part * (n + 1)) / nsp), k_hi = (int)(((long long)(part + 1) * (n + 1)) / nsp);
        const long long q = (long long)grp * 16 + p;
        const bool ok = q < a.NI;
        const long long qq = ok ? q : a.NI - 1;
        const float xv = io_ld(a.x, qq, a.x_bf16);
        const float x0v = a.x0 ? io_ld(a.x0, qq, a.x_bf16) : 0.f;
        const float dxv = xv - x0v;
        const float gv = ok ? io_ld(a.g, qq, a.x_bf16) : 0.f;
        const float gfxv = (ok && a.gfx) ? io_ld(a.gfx, qq, a.x_bf16) : 0.f;
        const float cotbase = gv * dxv * 0.5f;
        const long long bi = qq / d;
        const IoView hb = IoView{a.h, a.h_bf16} + (bi * ((long long)E * d) + (qq - bi * d));

        f32x4 c[BT];
        {
            const float* __restrict__ W0 = m.W[0];
            const float* __restrict__ b0 = m.b[0];
#pragma unroll
            for (int t = 0; t < BT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = feat_of(t, r, g);
                    c[t][r] = f < H1 ? b0[f] : (f == H1 ? 1.f : 0.f);
                }
            item_embedding_gemm<BT, 8>(hb, W0, H1, E, d, g, p, c);
        }
        if (k_hi <= k_lo) {                             // (cannot happen: the launcher keeps ns <= n + 1)
            if (ok)
                for (int f = g; f < H1; f += 4) a.dc[(size_t)part * a.NI * H1 + q * H1 + f] = 0.f;
            continue;
        }

        f32x4 dcs[BT];
#pragma unroll
        for (int t = 0; t < BT; ++t) dcs[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        float fxv = 0.f, fx0v = 0.f, dfdt = 0.f;

        auto node_t = [&](int k) __attribute__((always_inline)) {
            const float u = a.ccs[k] + 1.f;
            return k == 0 ? xv : __fadd_rn(x0v, __fmul_rn(dxv, u) * 0.5f);
        };
        f32x4 actF[BT], delta[BT];                      // current activations of the F chain / cotangents of the B chain
#pragma unroll
        for (int t = 0; t < BT; ++t) { actF[t] = f32x4{0.f, 0.f, 0.f, 0.f}; delta[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        auto layer1_reg = [&](auto ec, float tk) __attribute__((always_inline)) {
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4;
            if constexpr (e < NLIVE) actF[t][r] = hidden_act_f(fmaf(w1x[t][r], tk, c[t][r]), slope);
        };
        auto layer1 = [&](float tk) __attribute__((always_inline)) {
            swp_static_for<4 * BT>([&](auto ec) { layer1_reg(ec, tk); });
        };

        // operands of a stage, double-buffered by stage parity: F's split activations, B's split cotangents, the
        // transposed a_l operand of the dW product and the sign piece of a_l
        BFrag<NPF> bfv[2];
        BFrag<NPB> bdv[2];
        u32x2 aT[BT][NPB];                              // (single: read at the end of a GEMM region, dead after the dW region)
        u32x4 sgq[BKS];
        unsigned qF[8][NPF], qB[8][NPB];                // packed pieces of pair j (regs 2j, 2j+1) on their way into bfv / bdv
        float rf0[8], rf1[8], rb0[8], rb1[8];           // split residuals in flight
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int k2 = 0; k2 < NPF; ++k2) qF[j][k2] = 0u;
#pragma unroll
            for (int k2 = 0; k2 < NPB; ++k2) qB[j][k2] = 0u;
        }
        // -- micro-operations of "prepare stage i" (S1..S3 of the header), cut so that they can ride behind single MFMAs
        auto read_aT = [&](auto tauc, int X) __attribute__((always_inline)) {              // S1, one slot tile of a_l^T
            constexpr int tau = decltype(tauc)::value;
#pragma unroll
            for (int part2 = 0; part2 < NPB; ++part2) {
                const unsigned short* src = lds16 + X + (part2 * 16 + 4 * g + (p >> 2)) * TRS + 16 * tau + 4 * (p & 3);
                aT[tau][part2] = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4v __attribute__((address_space(3)))*)(src)));
            }
        };
        auto read_sg = [&](int X) __attribute__((always_inline)) {                         // the sign piece of a_l (own k-slots)
#pragma unroll
            for (int s = 0; s < BKS; ++s) sgq[s] = *reinterpret_cast<const u32x4*>(lds16 + X + own + s * 8);
        };
        auto pairF = [&](auto jc, auto stc) __attribute__((always_inline)) {              // S2, one rounding stage of pair j
            constexpr int j = decltype(jc)::value, st = decltype(stc)::value, t = j / 2, r = 2 * (j & 1);
            if constexpr (j < NPAIR) {
                if constexpr (st == 0) { rf0[j] = actF[t][r]; rf1[j] = actF[t][r + 1]; qF[j][0] = split_stage(rf0[j], rf1[j]); }
                if constexpr (st == 1) qF[j][1] = split_stage(rf0[j], rf1[j]);
                if constexpr (st == 2) qF[j][2] = split_last(rf0[j], rf1[j]);
            }
        };
        auto pairB = [&](auto jc, auto stc) __attribute__((always_inline)) {              // S3
            constexpr int j = decltype(jc)::value, st = decltype(stc)::value, t = j / 2, r = 2 * (j & 1);
            if constexpr (j < NPAIR) {
                if constexpr (st == 0) { rb0[j] = delta[t][r]; rb1[j] = delta[t][r + 1]; qB[j][0] = split_stage(rb0[j], rb1[j]); }
                if constexpr (st == 1) qB[j][1] = split_last(rb0[j], rb1[j]);
            }
        };
        auto commit = [&](auto curc, auto sc) __attribute__((always_inline)) {            // K-step s split: operands, delta to LDS
            constexpr int cur = decltype(curc)::value, s = decltype(sc)::value;
#pragma unroll
            for (int k2 = 0; k2 < NPF; ++k2) bfv[cur].v[s][k2] = u32x4{qF[4 * s][k2], qF[4 * s + 1][k2], qF[4 * s + 2][k2], qF[4 * s + 3][k2]};
#pragma unroll
            for (int k2 = 0; k2 < NPB; ++k2) bdv[cur].v[s][k2] = u32x4{qB[4 * s][k2], qB[4 * s + 1][k2], qB[4 * s + 2][k2], qB[4 * s + 3][k2]};
#pragma unroll
            for (int k2 = 0; k2 < NPB; ++k2)
                *reinterpret_cast<u32x4*>(lds16 + sdel + k2 * 16 * TRS + own + s * 8) = bdv[cur].v[s][k2];
        };
        auto store_bf = [&](auto curc, auto sc, int X) __attribute__((always_inline)) {   // a_i(k+1) pieces 0/1 into the slot B just read
            constexpr int cur = decltype(curc)::value, s = decltype(sc)::value;
#pragma unroll
            for (int k2 = 0; k2 < NPB; ++k2)
                *reinterpret_cast<u32x4*>(lds16 + X + k2 * 16 * TRS + own + s * 8) = bfv[cur].v[s][k2];
        };
        // -- S7 for one register: activation of the F chain, cotangent of the B chain through act'(a_l)
        auto s7_reg = [&](auto ec, const f32x4 (&acc)[BT], const f32x4 (&nd)[BT]) __attribute__((always_inline)) {
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4;
            if constexpr (e < NLIVE) {
                actF[t][r] = hidden_act_f(acc[t][r], slope);
                delta[t][r] = nd[t][r] * act_grad_q(sgq, t, r, slope);
            }
        };
        // -- output layer of the F chain's node kn in three cuts: dot product, scalar part, cotangent of the last layer
        float sdot = 0.f, doutN = 0.f, fpN = 0.f;
        auto out_dot = [&](auto ec) __attribute__((always_inline)) {
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4;
            if constexpr (e == 0) sdot = 0.f;
            if constexpr (e < NLIVE) sdot = fmaf(wout[t][r], actF[t][r], sdot);
        };
        auto out_scalar = [&](int kn, float scale) __attribute__((always_inline)) {
            const float sd = group_allreduce(sdot);
            // out_act_f / out_grad_f of cc_common.h (same expressions, same bits) without their branches: a branch here would
            // cut the slot-scheduled region into several basic blocks
            const bool sig = m.out_act != UMNN_OUT_ELU_PLUS_ONE;
            const float ex = __expf(sig ? -sd : sd);
            const float s1 = 1.f / (1.f + ex);
            const float f = sig ? s1 : (sd > 0.f ? sd + 1.f : ex);
            fpN = sig ? s1 * (1.f - s1) : (sd > 0.f ? 1.f : ex);
            if (kn == 0) fxv = f;
            if (kn == n) fx0v = f;
            const float rinv = -__frcp_rn(f * f);
            const float invs = a.inv_f ? rinv : 1.f;
            const float cot = fmaf(cotbase * invs, a.ccw[kn], kn == 0 ? gfxv : 0.f) * scale;
            doutN = cot * fpN;
        };
        auto out_reg = [&](auto ec) __attribute__((always_inline)) {       // dwo += dout a_L ; delta_L = dout wout act'(a_L)
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4;
            if constexpr (e < NLIVE) {
                dwo[t][r] = fmaf(doutN, actF[t][r], dwo[t][r]);
                delta[t][r] = doutN * wout[t][r] * (actF[t][r] > 0.f ? 1.f : slope);
            }
        };
        auto tail_reg = [&](auto ec, float tk) __attribute__((always_inline)) {   // B finishes node k: delta = delta_1
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4;
            if constexpr (e < NLIVE) {
                dcs[t][r] += delta[t][r];
                dW1x[t][r] = fmaf(delta[t][r], tk, dW1x[t][r]);
            }
        };

        // ---------------- prologue: F(k_lo) un-overlapped ----------------
        float tkB = node_t(k_lo);
        layer1(tkB);
#pragma unroll
        for (int i = 1; i <= NG; ++i) {
            BFrag<NPF> bf;
            split_regs<NRL, NPF>(actF, bf);
#pragma unroll
            for (int k2 = 0; k2 < NPB; ++k2)
#pragma unroll
                for (int s = 0; s < BKS; ++s)
                    *reinterpret_cast<u32x4*>(lds16 + so[i - 1] + k2 * 16 * TRS + own + s * 8) = bf.v[s][k2];
            f32x4 acc[BT];
            gemm_frags<NPF>(fragF + (i - 1) * IMG, bf, acc);
#pragma unroll
            for (int t = 0; t < BT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * t + r < NLIVE) actF[t][r] = hidden_act_f(acc[t][r], slope);
        }
        swp_static_for<4 * BT>(out_dot);
        out_scalar(k_lo, 1.f);
        // tangent pass at node 0 (d f / d x for the g_fx term): forward-mode through the same fragments
        if (k_lo == 0 && a.gfx) {
            f32x4 ta[BT];
            u32x4 sg[BKS];
#pragma unroll
            for (int s = 0; s < BKS; ++s) sg[s] = *reinterpret_cast<const u32x4*>(lds16 + so[0] + own + s * 8);
#pragma unroll
            for (int t = 0; t < BT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) ta[t][r] = 4 * t + r < NLIVE ? w1x[t][r] * act_grad_q(sg, t, r, slope) : 0.f;
#pragma unroll
            for (int l = 1; l < L; ++l) {
                BFrag<NPF> bf;
                split_regs<NRL, NPF>(ta, bf);
                f32x4 tz[BT];
                gemm_frags<NPF>(fragF + (l - 1) * IMG, bf, tz);
                if (l + 1 < L) {
#pragma unroll
                    for (int s = 0; s < BKS; ++s) sg[s] = *reinterpret_cast<const u32x4*>(lds16 + so[l + 1 < L ? l : 0] + own + s * 8);
                }
#pragma unroll
                for (int t = 0; t < BT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float fac = l + 1 < L ? act_grad_q(sg, t, r, slope) : (actF[t][r] > 0.f ? 1.f : slope);
                        ta[t][r] = 4 * t + r < NLIVE ? tz[t][r] * fac : 0.f;
                    }
            }
            float ds = 0.f;
#pragma unroll
            for (int t = 0; t < BT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) ds = fmaf(wout[t][r], ta[t][r], ds);
            dfdt = fpN * group_allreduce(ds);
        }
        swp_static_for<4 * BT>(out_reg);               // delta = delta_L(k_lo), dwo
        float tkF = node_t(k_lo + 1 < k_hi ? k_lo + 1 : k_hi - 1);
        layer1(tkF);                                    // actF = a_1(k_lo + 1)
        // fragment buffers of the GEMM regions (two per operand kind, see the load plan below)
        u32x4 bufW[2][BT], bufT[BT];
        auto loadW = [&](auto layerc, auto kc, auto tc) __attribute__((always_inline)) {        // W set k = 3 s + piece, tile t
            constexpr int li = decltype(layerc)::value, kk = decltype(kc)::value, t = decltype(tc)::value;
            constexpr int s = kk / 3, piece = kk % 3;
            bufW[kk & 1][t] = *reinterpret_cast<const u32x4*>(fragF + (li - 1) * IMG + ((t * BKS + s) * NPF + piece) * FRAG);
        };
        auto loadT = [&](auto layerc, auto kc, auto tc) __attribute__((always_inline)) {        // W^T set k = 2 s + piece, tile t
            constexpr int li = decltype(layerc)::value, kk = decltype(kc)::value, t = decltype(tc)::value;
            constexpr int s = kk / 2, piece = kk % 2;
            const unsigned short* lo = fragT + (li - 1) * IMG + (((2 * s + 0) * BKS + (t >> 1)) * NPF + piece) * FRAG + 4 * (t & 1);
            const unsigned short* hi = fragT + (li - 1) * IMG + (((2 * s + 1) * BKS + (t >> 1)) * NPF + piece) * FRAG + 4 * (t & 1);
            const u32x2 x0 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4v __attribute__((address_space(3)))*)(lo)));
            const u32x2 x1 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4v __attribute__((address_space(3)))*)(hi)));
            bufT[t] = u32x4{x0[0], x0[1], x1[0], x1[1]};
        };
        // preparation of a stage's K-step 0 (pairs 0..3), cut into 16 micro-operations: m -> pair m / 4, stage m % 4
        auto pair_op = [&](auto jc, auto opc) __attribute__((always_inline)) {
            constexpr int op = decltype(opc)::value;
            if constexpr (op == 0) pairF(jc, std::integral_constant<int, 0>{});
            if constexpr (op == 1) pairB(jc, std::integral_constant<int, 0>{});
            if constexpr (op == 2) pairF(jc, std::integral_constant<int, 1>{});
            if constexpr (op == 3) { pairF(jc, std::integral_constant<int, 2>{}); pairB(jc, std::integral_constant<int, 1>{}); }
        };
        // prologue of the pipeline: stage 1 of the first iteration, K-step 0 (K-step 1 is split inside the GEMM region)
        {
            constexpr std::integral_constant<int, 1> c1{};
            swp_static_for<16>([&](auto mc) {
                constexpr int mm = decltype(mc)::value;
                pair_op(std::integral_constant<int, mm / 4>{}, std::integral_constant<int, mm % 4>{});
            });
            commit(c1, std::integral_constant<int, 0>{});
            swp_static_for<BT>([&](auto tc) { loadW(c1, std::integral_constant<int, 0>{}, tc); });
        }

        // ---------------- pipelined node loop: B(k) with F(k+1) ----------------
        for (int k = k_lo; k < k_hi; ++k) {
            const bool has_next = k + 1 < k_hi;
            const int kn = has_next ? k + 1 : k;
            const float tkN = node_t(k + 2 < k_hi ? k + 2 : k_hi - 1);
            swp_static_for<NG>([&](auto iic) {
                constexpr int i = decltype(iic)::value + 1;          // F's layer (GEMM i -> i+1)
                constexpr int l = L - i;                                // B's layer: delta_{l+1} -> delta_l, dW_l
                constexpr int cur = i & 1, nxt = i < NG ? (i + 1) & 1 : 1;
                constexpr int inext = i < NG ? i + 1 : 1;               // F's layer of the stage that follows
                constexpr std::integral_constant<int, cur> curc{};
                constexpr std::integral_constant<int, nxt> nxtc{};
                constexpr std::integral_constant<int, i> ic{};
                constexpr std::integral_constant<int, l> lc{};
                const int X = so[l - 1];                                // slot of a_l(k), receiving a_i(k+1)
                f32x4 nd[BT], acc[BT];
                u32x2 dT[BT][NPB];
                // ---- region G: the 72 MFMAs of the two GEMMs, one per slot.  Fragments arrive through four register
                // buffers loaded 8..20 slots ahead of their first use (W sets k = 3s + piece -> bufW[k & 1], W^T sets
                // k = 2s + piece -> bufT[k & 1]; W set 0 was loaded in the previous region D).  The vector work of K-step 1's
                // split (pairs 4..7; K-step 1's MFMAs start at slot 36) rides behind the first slots.
                swp_static_for<72>([&](auto nc) {
                    constexpr int nn = decltype(nc)::value, s = nn / 36, idx = nn % 36;
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (idx < 12) {
                        constexpr int t = idx % 4, ba = idx / 4;
                        acc[t] = mfma_bf16(bufW[(3 * s) & 1][t], bfv[cur].v[s][ba], (s == 0 && idx < 4) ? zero : acc[t]);
                    } else if constexpr (idx < 20) {
                        constexpr int t = (idx - 12) % 4, ba = (idx - 12) / 4;
                        nd[t] = mfma_bf16(bufT[t], bdv[cur].v[s][ba], (s == 0 && idx < 16) ? zero : nd[t]);
                    } else if constexpr (idx < 28) {
                        constexpr int t = (idx - 20) % 4, ba = (idx - 20) / 4;
                        acc[t] = mfma_bf16(bufW[(3 * s + 1) & 1][t], bfv[cur].v[s][ba], acc[t]);
                    } else if constexpr (idx < 32) {
                        constexpr int t = idx - 28;
                        nd[t] = mfma_bf16(bufT[t], bdv[cur].v[s][0], nd[t]);
                    } else {
                        constexpr int t = idx - 32;
                        acc[t] = mfma_bf16(bufW[(3 * s + 2) & 1][t], bfv[cur].v[s][0], acc[t]);
                    }
                    // fragment loads, one fragment per slot: W set k -> bufW[k & 1] (k = 0 came with the previous region D),
                    // W^T set k -> bufT, each 8+ slots ahead of its first MFMA and after the last MFMA of the set it replaces
                    if constexpr (nn >= 0 && nn < 4) loadT(lc, std::integral_constant<int, 0>{}, std::integral_constant<int, nn - 0>{});      // used 12..19
                    if constexpr (nn >= 4 && nn < 8) loadW(ic, std::integral_constant<int, 1>{}, std::integral_constant<int, nn - 4>{});      // used 20..27
                    if constexpr (nn >= 12 && nn < 16) loadW(ic, std::integral_constant<int, 2>{}, std::integral_constant<int, nn - 12>{});   // used 32..35
                    if constexpr (nn >= 20 && nn < 24) loadT(lc, std::integral_constant<int, 1>{}, std::integral_constant<int, nn - 20>{});   // used 28..31
                    if constexpr (nn >= 28 && nn < 32) loadW(ic, std::integral_constant<int, 3>{}, std::integral_constant<int, nn - 28>{});   // used 36..47
                    if constexpr (nn >= 32 && nn < 36) loadT(lc, std::integral_constant<int, 2>{}, std::integral_constant<int, nn - 32>{});   // used 48..55
                    if constexpr (nn >= 40 && nn < 44) loadW(ic, std::integral_constant<int, 4>{}, std::integral_constant<int, nn - 40>{});   // used 56..63
                    if constexpr (nn >= 48 && nn < 52) loadW(ic, std::integral_constant<int, 5>{}, std::integral_constant<int, nn - 48>{});   // used 68..71
                    if constexpr (nn >= 56 && nn < 60) loadT(lc, std::integral_constant<int, 3>{}, std::integral_constant<int, nn - 56>{});   // used 64..67
                    // K-step 1 of this stage's operands: pairs 4..7, then delta's K-step 1 to LDS
                    if constexpr (nn < 16) pair_op(std::integral_constant<int, 4 + nn / 4>{}, std::integral_constant<int, nn % 4>{});
                    if constexpr (nn == 16) commit(curc, std::integral_constant<int, 1>{});
                    // operands of region D: a_l^T and its sign piece out of X, THEN a_i(k+1) into X; delta^T out of the delta slot
                    if constexpr (nn >= 60 && nn < 64) read_aT(std::integral_constant<int, nn - 60>{}, X);
                    if constexpr (nn == 64) read_sg(X);
                    if constexpr (nn == 65) store_bf(curc, std::integral_constant<int, 0>{}, X);
                    if constexpr (nn == 66) store_bf(curc, std::integral_constant<int, 1>{}, X);
                    if constexpr (nn >= 67 && nn < 71) {
                        constexpr int to = nn - 67;
#pragma unroll
                        for (int part2 = 0; part2 < NPB; ++part2) {
                            const unsigned short* src = lds16 + sdel + (part2 * 16 + 4 * g + (p >> 2)) * TRS + 16 * to + 4 * (p & 3);
                            dT[to][part2] = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4v __attribute__((address_space(3)))*)(src)));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                // ---- region D: dW_l += delta_{l+1} (x) a_l, one MFMA per slot; behind them S7 of this stage and K-step 0 of the
                // NEXT stage's operands (at the last stage: the tail of node k, the output layer of node k+1, layer 1 of k+2 first)
                swp_static_for<NDW>([&](auto nc) {
                    constexpr int nn = decltype(nc)::value;
                    // (output-row tile outermost: delta^T tile `to` is dead after its 12 slots; per accumulator the order of
                    // the three cross terms is the round-2 kernel's)
                    constexpr int to = nn / (3 * BT), term = (nn % (3 * BT)) / BT, ti = nn % BT;
                    constexpr int wa = term == 2 ? 1 : 0, ba = term == 1 ? 1 : 0;
                    dW[l - 1][to][ti] = mfma_bf16_k16(dT[to][wa], aT[ti][ba], dW[l - 1][to][ti]);
                    if constexpr (nn < 16) s7_reg(std::integral_constant<int, nn>{}, acc, nd);
                    if constexpr (i < NG) {
                        if constexpr (nn >= 16 && nn < 32) pair_op(std::integral_constant<int, (nn - 16) / 4>{}, std::integral_constant<int, (nn - 16) % 4>{});
                        if constexpr (nn == 33) commit(nxtc, std::integral_constant<int, 0>{});
                    } else {
                        // B's tail rides with S7 (tail_reg of registers 2m, 2m+1 after both have their delta_1)
                        if constexpr (nn >= 8 && nn < 16) { tail_reg(std::integral_constant<int, 2 * (nn - 8)>{}, tkB); tail_reg(std::integral_constant<int, 2 * (nn - 8) + 1>{}, tkB); }
                        if constexpr (nn >= 16 && nn < 20) swp_static_for<4>([&](auto uc) { out_dot(std::integral_constant<int, 4 * (nn - 16) + decltype(uc)::value>{}); });
                        if constexpr (nn == 20) out_scalar(kn, has_next ? 1.f : 0.f);
                        if constexpr (nn >= 21 && nn < 29) { out_reg(std::integral_constant<int, 2 * (nn - 21)>{}); out_reg(std::integral_constant<int, 2 * (nn - 21) + 1>{}); }
                        if constexpr (nn >= 29 && nn < 37) { layer1_reg(std::integral_constant<int, 2 * (nn - 29)>{}, tkN); layer1_reg(std::integral_constant<int, 2 * (nn - 29) + 1>{}, tkN); }
                        // K-step 0 of stage 1 of the next iteration: B's pairs as soon as out_reg wrote them, F's after layer 1
                        if constexpr (nn >= 25 && nn < 29) pairB(std::integral_constant<int, nn - 25>{}, std::integral_constant<int, 0>{});
                        if constexpr (nn >= 29 && nn < 33) pairB(std::integral_constant<int, nn - 29>{}, std::integral_constant<int, 1>{});
                        if constexpr (nn >= 33 && nn < 37) pairF(std::integral_constant<int, nn - 33>{}, std::integral_constant<int, 0>{});
                        if constexpr (nn >= 37 && nn < 41) pairF(std::integral_constant<int, nn - 37>{}, std::integral_constant<int, 1>{});
                        if constexpr (nn >= 41 && nn < 45) pairF(std::integral_constant<int, nn - 41>{}, std::integral_constant<int, 2>{});
                        if constexpr (nn == 45) commit(nxtc, std::integral_constant<int, 0>{});
                    }
                    // first W set of the next GEMM region
                    if constexpr (nn >= 40 && nn < 44) loadW(std::integral_constant<int, inext>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, nn - 40>{});
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            tkB = tkF;
            tkF = tkN;
            // slot rotation: stage i stored a_i(k+1) where a_{L-i}(k) was
            if constexpr (NG >= 2) { const int tmp = so[0]; so[0] = so[NG - 1]; so[NG - 1] = tmp; }
        }

        if (ok) {
#pragma unroll
            for (int t = 0; t < BT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = feat_of(t, r, g);
                    if (f < H1) a.dc[(size_t)part * a.NI * H1 + q * H1 + f] = dcs[t][r];
                }
            if (g == 0) {
                if (a.dx && k_lo == 0) io_st(a.dx, q, fmaf(gfxv, dfdt, fxv * gv), a.x_bf16);
                if (a.dx0 && k_hi == n + 1) io_st(a.dx0, q, -fx0v * gv, a.x_bf16);
            }
        }
    }

    // ---------------- write this wave's partial d_theta (rows / columns of the dW tiles run over register slots) ----------------
    float* part = a.partials + (size_t)wave_global * a.n_params;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int l = 1 + j;
        const int Hin = m.width[l], Hout = m.width[l + 1];
#pragma unroll
        for (int to = 0; to < BT; ++to)
#pragma unroll
            for (int ti = 0; ti < BT; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int fo = slot_feature(16 * to + 4 * g + r);
                    const int fi = slot_feature(16 * ti + (lane & 15));
                    if (fo < Hout) {
                        const int idx = fi < Hin ? a.poffW[l] + fo * Hin + fi : (fi == Hin ? a.poffb[l] + fo : -1);
                        if (idx >= 0) part[idx] = dW[j][to][ti][r];
                    }
                }
    }
#pragma unroll
    for (int t = 0; t < BT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v1 = dW1x[t][r], v2 = dwo[t][r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { v1 += __shfl_xor(v1, o); v2 += __shfl_xor(v2, o); }
            const int f = feat_of(t, r, g);
            if (p == 0) {
                if (f < H1) part[a.poffW[0] + f * (1 + E)] = v1;
                const int idx = f < HL ? a.poffW[L] + f : (f == HL ? a.poffb[L] : -1);
                if (idx >= 0) part[idx] = v2;
            }
        }
}
