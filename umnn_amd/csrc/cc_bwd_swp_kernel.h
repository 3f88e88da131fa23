// Software-pipelined one-pass backward on the bf16 matrix cores (round 3).  Same arithmetic, operands and accumulation
// order as cc_bwd_bf16_kernel (cc_bwd_bf16_kernel.h; reference lines ParallelNeuralIntegral.py:66-94,110-123) -- results are
// bit-identical -- but the node loop is restructured so that ONE wave carries TWO independent dependency chains:
//
//     iteration k :   B(k)   = backward sweep of node k        (delta chain through W^T, dW += delta (x) a)
//                     F(k+1) = forward recompute of node k+1   (six-term split GEMMs, activations, output layer)
//
// The old loop ran F(k) then B(k): split -> GEMM -> activation -> split -> ... is one serial chain, and with one wave per
// SIMD (the 192 dW accumulators need the whole register file) every MFMA -> VALU -> MFMA dependency was exposed: matrix
// pipe 53 % busy.  Here stage i of an iteration pairs F's layer i with B's layer L-i:
//     S1  a_{L-i}(k): transposed dW operand + sign piece read from its LDS slot X
//     S2  split a_i(k+1) (3 pieces), pieces 0/1 stored INTO X (B's reads of X were issued first; LDS is in order)
//     S3  split delta_{L-i+1}(k) (2 pieces), stored to the delta slot
//     S4  W^T GEMM (24 MFMA)        S5  forward GEMM (48 MFMA)
//     S6  dW += delta (x) a (48 MFMA) -- feeds accumulators only: S7 and the next stage's S1..S3 (all VALU / LDS) run in
//     S7  activations of both chains      its issue shadow.
// What makes that fit:
//   * a_l no longer lives in registers between F and B (48 registers in the old kernel): its two leading bf16 pieces go to
//     an LDS slot when they are split -- the same [piece][point][slot] tile the dW product transposes through with
//     ds_read_b64_tr_b16 -- and the sign of a_l (activation derivative) is read back from there.  F runs one node ahead
//     and produces a_1..a_{L-1} in the order B consumes them backwards, so L-1 slots rotate (stage i frees the slot of
//     a_{L-i}(k) for a_i(k+1)): (L-1) + 1 slots of 4.5 KB per wave.
//   * ONE weight image serves W and W^T.  The forward fragment image (3 pieces, 24 KB per layer) is read with
//     ds_read_b128 by the recompute and with ds_read_b64_tr_b16 by the delta chain: lane (g', p) of a W^T fragment (t, s)
//     wants k-slots j = 4h + e  <->  W[16 (2s+h) + 4e + g'][16t + 4 (p&3) + (p>>2)]; in the forward image those are, for
//     fixed (h, e), 4 consecutive bf16 of the lane (g'_f = p&3, rho_f = 4 g' + e) of fragment (2s+h, t>>1), half t&1 --
//     exactly the 4 x 16 block a transposing read delivers.  72 KB instead of 120 KB of images; the 16-byte units of a
//     fragment are rotated by 8 for g'_f >= 2 so the transposing reads are 2-way conflicted (inherent: 32 lanes read
//     the same 8-byte half of 16 units) instead of 4-way, and the b128 reads stay conflict-free.
//     The constant-one feature (bias column / unit row of the forward image) forms a closed subspace under W^T -- it only
//     ever feeds the constant feature's own delta, whose dW rows and dc entries are discarded at write-out.
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
    const int hi16 = (r & 1) ? (int)(u & 0xffff0000u) : (int)(u << 16);
    return hi16 > 0 ? 1.f : slope;
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
            for (int se = 0; se < (E + 3) / 4; ++se) {
                const int e = 4 * se + g;
                const float hv = e < E ? hb[(long long)e * d] : 0.f;
#pragma unroll
                for (int t = 0; t < BT; ++t) {
                    const int fo = fout_of(t, p);
                    const float A = (fo < H1 && e < E) ? W0[fo * (1 + E) + 1 + e] : 0.f;
                    c[t] = mfma16(A, hv, c[t]);
                }
            }
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
        u32x2 aTv[2][BT][NPB];
        u32x4 sgv[2][BKS];
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
        auto prep_reads = [&](auto curc, int X) __attribute__((always_inline)) {          // S1
            constexpr int cur = decltype(curc)::value;
            tr_tile_read(lds16 + X, g, p, aTv[cur]);
#pragma unroll
            for (int s = 0; s < BKS; ++s) sgv[cur][s] = *reinterpret_cast<const u32x4*>(lds16 + X + own + s * 8);
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
        auto commit = [&](auto curc, auto sc, int X) __attribute__((always_inline)) {     // K-step s complete: operands + LDS
            constexpr int cur = decltype(curc)::value, s = decltype(sc)::value;
#pragma unroll
            for (int k2 = 0; k2 < NPF; ++k2) bfv[cur].v[s][k2] = u32x4{qF[4 * s][k2], qF[4 * s + 1][k2], qF[4 * s + 2][k2], qF[4 * s + 3][k2]};
#pragma unroll
            for (int k2 = 0; k2 < NPB; ++k2) bdv[cur].v[s][k2] = u32x4{qB[4 * s][k2], qB[4 * s + 1][k2], qB[4 * s + 2][k2], qB[4 * s + 3][k2]};
#pragma unroll
            for (int k2 = 0; k2 < NPB; ++k2) {
                *reinterpret_cast<u32x4*>(lds16 + X + k2 * 16 * TRS + own + s * 8) = bfv[cur].v[s][k2];
                *reinterpret_cast<u32x4*>(lds16 + sdel + k2 * 16 * TRS + own + s * 8) = bdv[cur].v[s][k2];
            }
        };
        auto prep_plain = [&](auto curc, int X) __attribute__((always_inline)) {          // the whole preparation, un-overlapped
            prep_reads(curc, X);
            swp_static_for<8>([&](auto jc) {
                pairF(jc, std::integral_constant<int, 0>{}); pairF(jc, std::integral_constant<int, 1>{}); pairF(jc, std::integral_constant<int, 2>{});
                pairB(jc, std::integral_constant<int, 0>{}); pairB(jc, std::integral_constant<int, 1>{});
            });
            commit(curc, std::integral_constant<int, 0>{}, X);
            commit(curc, std::integral_constant<int, 1>{}, X);
        };
        // -- S7 for one register: activation of the F chain, cotangent of the B chain through act'(a_l)
        auto s7_reg = [&](auto ec, auto curc, const f32x4 (&acc)[BT], const f32x4 (&nd)[BT]) __attribute__((always_inline)) {
            constexpr int e = decltype(ec)::value, cur = decltype(curc)::value, t = e / 4, r = e % 4;
            if constexpr (e < NLIVE) {
                actF[t][r] = hidden_act_f(acc[t][r], slope);
                delta[t][r] = nd[t][r] * act_grad_q(sgv[cur], t, r, slope);
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
        prep_plain(std::integral_constant<int, 1>{}, so[L - 2]);       // stage 1 of the first iteration

        // ---------------- pipelined node loop: B(k) with F(k+1) ----------------
        for (int k = k_lo; k < k_hi; ++k) {
            const bool has_next = k + 1 < k_hi;
            const int kn = has_next ? k + 1 : k;
            const float tkN = node_t(k + 2 < k_hi ? k + 2 : k_hi - 1);
            swp_static_for<NG>([&](auto iic) {
                constexpr int i = decltype(iic)::value + 1;          // F's layer (GEMM i -> i+1)
                constexpr int l = L - i;                                // B's layer: delta_{l+1} -> delta_l, dW_l
                constexpr int cur = i & 1, nxt = (i + 1) & 1;
                constexpr std::integral_constant<int, cur> curc{};
                constexpr std::integral_constant<int, nxt> nxtc{};
                // ---- region G: the two GEMMs the chains wait for (S4, S5), back to back on the matrix pipe
                f32x4 nd[BT], acc[BT];
                gemm_frags_T(fragT + (l - 1) * IMG, bdv[cur], nd);
                gemm_frags<NPF>(fragF + (i - 1) * IMG, bfv[cur], acc);
                u32x2 dT[BT][NPB];
                tr_tile_read(lds16 + sdel, g, p, dT);
                __builtin_amdgcn_sched_barrier(0);
                // ---- region D: dW_l += delta_{l+1} (x) a_l, one MFMA per slot; the vector work of S7 and of the NEXT stage's
                // S1..S3 (or, at the last stage, the tail of the node) rides behind the MFMAs, one cut per slot
                swp_static_for<NDW>([&](auto nc) {
                    constexpr int nn = decltype(nc)::value;
                    constexpr int term = nn / (BT * BT), to = (nn % (BT * BT)) / BT, ti = nn % BT;
                    constexpr int wa = term == 2 ? 1 : 0, ba = term == 1 ? 1 : 0;
                    dW[l - 1][to][ti] = mfma_bf16_k16(dT[to][wa], aTv[cur][ti][ba], dW[l - 1][to][ti]);
                    if constexpr (i < NG) {
                        // per pair j six cuts: S7 of its two registers, then the rounding stages of both splits
                        constexpr int j = nn / 6, op = nn % 6;
                        constexpr std::integral_constant<int, j> jc{};
                        const int Xn = so[l - 2];                       // slot of a_{l-1}(k): next stage's X
                        if constexpr (nn == 1) prep_reads(nxtc, Xn);
                        if constexpr (op == 0) s7_reg(std::integral_constant<int, 2 * j>{}, curc, acc, nd);
                        if constexpr (op == 1) s7_reg(std::integral_constant<int, 2 * j + 1>{}, curc, acc, nd);
                        if constexpr (op == 2) pairF(jc, std::integral_constant<int, 0>{});
                        if constexpr (op == 3) pairB(jc, std::integral_constant<int, 0>{});
                        if constexpr (op == 4) pairF(jc, std::integral_constant<int, 1>{});
                        if constexpr (op == 5) { pairF(jc, std::integral_constant<int, 2>{}); pairB(jc, std::integral_constant<int, 1>{}); }
                        if constexpr (nn == 6 * 3 + 5) commit(nxtc, std::integral_constant<int, 0>{}, Xn);
                        if constexpr (nn == 6 * 7 + 5) commit(nxtc, std::integral_constant<int, 1>{}, Xn);
                    } else {
                        // last stage: S7 (delta becomes delta_1(k), actF a_L(k+1)), B's tail, F's output layer, layer 1 of node k+2
                        if constexpr (nn < 16) s7_reg(std::integral_constant<int, nn>{}, curc, acc, nd);
                        else if constexpr (nn < 24) { tail_reg(std::integral_constant<int, 2 * (nn - 16)>{}, tkB); tail_reg(std::integral_constant<int, 2 * (nn - 16) + 1>{}, tkB); }
                        else if constexpr (nn < 28) swp_static_for<4>([&](auto uc) { out_dot(std::integral_constant<int, 4 * (nn - 24) + decltype(uc)::value>{}); });
                        else if constexpr (nn == 28) out_scalar(kn, has_next ? 1.f : 0.f);
                        else if constexpr (nn < 37) { out_reg(std::integral_constant<int, 2 * (nn - 29)>{}); out_reg(std::integral_constant<int, 2 * (nn - 29) + 1>{}); }
                        else if constexpr (nn < 45) { layer1_reg(std::integral_constant<int, 2 * (nn - 37)>{}, tkN); layer1_reg(std::integral_constant<int, 2 * (nn - 37) + 1>{}, tkN); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            tkB = tkF;
            tkF = tkN;
            // slot rotation: stage i stored a_i(k+1) where a_{L-i}(k) was
            if constexpr (NG >= 2) { const int tmp = so[0]; so[0] = so[NG - 1]; so[NG - 1] = tmp; }
            // stage 1 of the next iteration (its vector work is what is still exposed per node)
            prep_plain(std::integral_constant<int, 1>{}, so[L - 2]);
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
