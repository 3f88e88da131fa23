// Forward quadrature for the flagship shape (every hidden layer 48..51 wide, bf16x3) on v_mfma_f32_32x32x16_bf16: the wave's 32
// integrals are the N dimension of ONE matrix instruction (cc_fwd_bf16_kernel.h runs them as two 16-point tiles of 16x16x32
// instructions).  Same matrix time per node, HALF the matrix instructions (69 instead of 138), and every one of them leaves a
// 32-cycle issue shadow that hides about five vector instructions instead of two (tools/ubench/fill.hip: 16x16x32 + 3 VALU =
// 19.6 cycles per 16 K FLOP, 32x32x16 + 6 VALU = 31.7 cycles per 32 K FLOP at two waves per SIMD).
// Same reference lines as cc_forward.hip (ParallelNeuralIntegral.py:37-65, UMNNMAF.py:263-284).
//
// Layout (probed with tools/ubench/mfma32.hip): lane l = (n = l & 31, hf = l >> 5).  D: column n, register v = 4i + r <-> row
// 8i + 4 hf + r.  A: row l & 31, k-slots 8 (l >> 5) + j.  B: column l & 31, k-slots 8 (l >> 5) + j.
//   point      n (one integral per lane pair)
//   features   register q = 16 mt + 4 i + r of half hf  <->  feature 2 q + hf      (26 live registers for widths <= 51, dense)
//   K-step c   (registers 8c .. 8c+7): lane (n, hf) supplies its OWN 8 registers as k-slots 8 hf + j -- the accumulators of
//              one layer are, after activation / split / packing, the B operands of the next; nothing moves across lanes.
// Three cross terms x 26 registers = 78 k-slots per half -> ten K-steps: (c = 0,1,2) x {Whi ahi, Wlo ahi, Whi alo} and ONE merged
// K-step for registers 24, 25: k-slots [ahi24 ahi25 | alo24 alo25 | ahi24 ahi25 | 0 0] against [Whi | Whi | Wlo | 0].
// Per layer 2 M-tiles x 10 = 20 matrix instructions + 3 that return the split remainders (a - bf16(a), exact, as in the
// pipelined 16x16x32 kernel).  Weight fragments: 7 per M-tile and layer (Whi c0..2, Wlo c0..2, merged), 14 KB per layer in LDS,
// streamed (each is used by one or two consecutive instructions: no fragment cache in registers).
#ifndef UMNN_ASM_TIED
#define UMNN_ASM_TIED 1      // cc_common.h: inline-assembly outputs tied to inputs in the forward translation units
#endif
#include "cc_bf16.h"
#include "cc_fwd_bf16_kernel.h"
using namespace UMNN_FWD_NS;

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int P32_NL = 26;                 // live registers per lane
constexpr int P32_FRAGS = 14;              // fragments per layer image: [mt][Whi c0..2, Wlo c0..2, merged]
constexpr int P32_IMG = P32_FRAGS * 512;   // ushorts per layer image

// output feature of accumulator row m (= lane & 31 of an A operand) of M-tile mt
__device__ __forceinline__ int p32_fout(int mt, int m) { return 2 * (16 * mt + 4 * (m >> 3) + (m & 3)) + ((m >> 2) & 1); }

__device__ __forceinline__ void p32_stage(const MlpDev& m, unsigned short* lds16, int tid, int nthreads) {
    const int L = m.n_linear - 1;
    for (int l = 1; l < L; ++l) {
        const int Hin = m.width[l], Hout = m.width[l + 1];
        const float* __restrict__ W = m.W[l];
        const float* __restrict__ b = m.b[l];
        unsigned short* img = lds16 + (l - 1) * P32_IMG;
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
        // one (M-tile, register group c or the merged pair, lane) per iteration: the lane's 8 k-slots of both pieces
        for (int idx = tid; idx < 2 * 4 * 64; idx += nthreads) {
            const int la = idx & 63, c = (idx >> 6) & 3, mt = idx >> 8;
            const int fo = p32_fout(mt, la & 31), hfk = la >> 5;
            u32x4* dst = reinterpret_cast<u32x4*>(img + (mt * 7) * 512 + la * 8);
            if (c < 3) {
                unsigned hi[4], lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned q[2];
                    split_pair<2>(wv(fo, 2 * (8 * c + 2 * e) + hfk), wv(fo, 2 * (8 * c + 2 * e + 1) + hfk), q);
                    hi[e] = q[0]; lo[e] = q[1];
                }
                dst[c * 64] = u32x4{hi[0], hi[1], hi[2], hi[3]};
                dst[(3 + c) * 64] = u32x4{lo[0], lo[1], lo[2], lo[3]};
            } else {
                unsigned q[2];
                split_pair<2>(wv(fo, 2 * 24 + hfk), wv(fo, 2 * 25 + hfk), q);
                dst[6 * 64] = u32x4{q[0], q[0], q[1], 0u};
            }
        }
    }
}

template <int NG>          // hidden->hidden layers (1..4)
__global__ __launch_bounds__(UMNN_BLOCK) void cc_fwd_p32_kernel(const FwdBf16Args args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const FwdArgs& a = args.f;
    const MlpDev& m = a.m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hf = lane >> 5;
    const int L = m.n_linear - 1;
    const int H1 = m.width[1], HL = m.width[L];
    const int E = a.E, d = a.d, nb = a.n;
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    unsigned short* lds16 = reinterpret_cast<unsigned short*>(lds);

    p32_stage(m, lds16, tid, UMNN_BLOCK);
    __syncthreads();

    const unsigned grp = xcd_remap(blockIdx.x, gridDim.x) * UMNN_WAVES_PER_BLOCK + wid;      // 32 integrals
    const bool live = grp < a.ngroups;
    float Facc = 0.f, fxv = 0.f, fx0v = 0.f, xv = 0.f, x0v = 0.f, dxv = 0.f;
    if (live) {
        const long long q = (long long)grp * 32 + n;
        const long long qq = q < a.NI ? q : a.NI - 1;
        xv = io_ld(a.x, qq, a.x_bf16);
        x0v = a.x0 ? io_ld(a.x0, qq, a.x_bf16) : 0.f;
        dxv = xv - x0v;
        const long long bi = qq / d;
        const IoView hb = IoView{a.h, a.h_bf16} + (bi * ((long long)E * d) + (qq - bi * d));

        // ---- per-lane constants and the hoisted first-layer term (fp32 32x32x2 MFMA: k = hf)
        float w1x[P32_NL], wout[P32_NL];
        f32x16 c[2];
        {
            const float* __restrict__ W0 = m.W[0];
            const float* __restrict__ b0 = m.b[0];
            const float* __restrict__ WL = m.W[L];
            const float bL = m.b[L][0];
#pragma unroll
            for (int qi = 0; qi < 32; ++qi) {
                const int f = 2 * qi + hf;
                if (qi < P32_NL) {
                    w1x[qi] = f < H1 ? W0[f * (1 + E)] : 0.f;
                    wout[qi] = f < HL ? WL[f] : (f == HL ? bL : 0.f);
                }
                c[qi >> 4][qi & 15] = f < H1 ? b0[f] : (f == H1 ? 1.f : 0.f);
            }
            const int nse = (E + 1) / 2;
            for (int s0 = 0; s0 < nse; s0 += 8) {
                float hv[8], Av[2][8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = 2 * (s0 + j) + hf;
                    const bool in = s0 + j < nse && e < E;
                    const int ec = in ? e : 0;
                    const float hl = hb[(long long)ec * d];
                    hv[j] = in ? hl : 0.f;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const int fo = p32_fout(mt, n);
                        const bool ina = in && fo < H1;
                        const float al = W0[(ina ? fo : 0) * (1 + E) + 1 + ec];
                        Av[mt][j] = ina ? al : 0.f;
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (s0 + j < nse) {
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) c[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Av[mt][j], hv[j], c[mt], 0, 0, 0);
                    }
                }
            }
        }
        // 0/-1 selection fragments of the remainder instructions: row m (i = m>>3, half (m>>2)&1, r = m&3) picks the k-slot in
        // which that very register was packed -- 8 * half + 4 (i & 1) + r of the K-step holding registers with (i >> 1) == par
        u32x4 sel[2];
        {
            const int mrow = n, hfk = hf, i = mrow >> 3, hfp = (mrow >> 2) & 1, r = mrow & 3;
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const bool on = hfk == hfp && (i >> 1) == par;
                const int j = 4 * (i & 1) + r;
                unsigned w[4] = {0u, 0u, 0u, 0u};
                if (on) w[j >> 1] = 0xBF80u << (16 * (j & 1));        // bf16(-1)
                sel[par] = u32x4{w[0], w[1], w[2], w[3]};
            }
        }
        const u32x4* img0 = reinterpret_cast<const u32x4*>(lds16) + lane;      // fragment f of layer l: img0[((l-1)*14 + f) * 64]

        // ---- software-pipelined node loop.  A layer's GEMM is 20 matrix instructions ("slots"); slot i also carries its slice of
        // the vector work that prepares the NEXT layer's operands from the previous layer's output zin (activation, bf16 pieces,
        // exact remainders on the matrix pipe), fenced so that it sits in the 32-cycle shadow of the slot's MFMA:
        //   MFMA order   (M-tiles alternate)  0-3: (Whi0, Wlo0) x Bh0   4-7: (Whi1, Wlo1) x Bh1   8-11: (Whi2, Wlo2) x Bh2
        //                12-13: Whi0 x Bl0   14-15: Whi1 x Bl1   16-17: Whi2 x Bl2   18-19: Wm x Bm
        //   vector work  0-3: registers 8..15 of zin (act, hi) -> Bh1, then the remainder MFMAs of c0 (slot 3) and c1 (slot 4)
        //                4-7: registers 16..23 (act, hi) -> Bh2      8: registers 24, 25 (act, hi), remainder MFMA of c2
        //                9, 10: lo of c0, c1 -> Bl0, Bl1      11: split of 24, 25 -> Bm      12: lo of c2 -> Bl2
        //                after the section: registers 0..7 of this layer's M-tile 0 (act, hi) -> Bh0 of the next layer
        // Fragments stream from LDS three slots ahead through a four-deep ring.
        constexpr int NSL = 20;
        using Slots = std::make_integer_sequence<int, NSL>;
        u32x4 Bh[3], Bl[3], Bm = {0u, 0u, 0u, 0u};
        u32x4 fb[4];
        unsigned mhi = 0u;
        // fragment of slot s: index into the layer image (M-tile * 7 + kind)
        auto frag_of = [](int sl) {
            constexpr int tab[NSL] = {0, 7, 3, 10, 1, 8, 4, 11, 2, 9, 5, 12, 0, 7, 1, 8, 2, 9, 6, 13};
            return tab[sl];
        };
        auto frag_ld = [&](int l, int sl) { return img0[((l - 1) * P32_FRAGS + frag_of(sl)) * 64]; };
        auto act2 = [&](f32x16& zz, int v, auto pre) {          // activation of registers v, v+1 (skipped for a pre-activated input)
            if constexpr (!decltype(pre)::value) {
                zz[v] = hidden_act_f(zz[v], slope);
                zz[v + 1] = hidden_act_f(zz[v + 1], slope);
            }
        };
        auto cvt2 = [&](const f32x16& zz, int v) {
            const f32x2 pr = {zz[v], zz[v + 1]};
            return __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2));
        };
        // one GEMM section: acc = layer l applied to the operands; zin (the previous layer's output) is packed on the way;
        // PRE: zin is already activated (layer 1); LAST: acc's M-tile 0 is left raw (the output dot follows)
        float a24 = 0.f, a25 = 0.f;
        // free slots 13..19 of every section: layer 1 of the NEXT node (fma + activation of registers zN, in order 0..25), and in
        // the node's last section the leading pieces of its registers 0..7 (the next node's first operand)
        // The node's FIRST section (pre-activated input: little packing work) also carries the output dot product of the PREVIOUS
        // node, whose last layer (zprev, raw) completed just before: activation + fma per register, two registers per slot.
        float sdot = 0.f;
        auto section = [&](auto pre, auto last, auto dot_c, auto sidx_c, int l, int lnext, f32x16 (&zin)[2], f32x16 (&acc)[2],
                           f32x16 (&zN)[2], float tkn, const f32x16 (&zprev)[2]) {
            static_for(Slots{}, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int mt = i & 1;
                // ---- the slot's matrix instruction
                {
                    const u32x4 w = fb[i & 3];
                    u32x4 bop;
                    if constexpr (i < 4) bop = Bh[0];
                    else if constexpr (i < 8) bop = Bh[1];
                    else if constexpr (i < 12) bop = Bh[2];
                    else if constexpr (i < 14) bop = Bl[0];
                    else if constexpr (i < 16) bop = Bl[1];
                    else if constexpr (i < 18) bop = Bl[2];
                    else bop = Bm;
                    if constexpr (i < 2) {
                        f32x16 zero;
#pragma unroll
                        for (int v = 0; v < 16; ++v) zero[v] = 0.f;
                        acc[mt] = mfma32(w, bop, zero);
                    } else {
                        acc[mt] = mfma32(w, bop, acc[mt]);
                    }
                    // ring: the fragment of slot i + 3 (of the next section when past the end)
                    if constexpr (i + 3 < NSL) fb[(i + 3) & 3] = frag_ld(l, i + 3);
                    else fb[(i + 3) & 3] = frag_ld(lnext, i + 3 - NSL);
                }
                // ---- the slot's vector work.  A remainder MFMA rewrites its whole 16-register tuple, so it is issued only after
                // every raw read of that tuple, and the tuple is read again no sooner than three slots later.
                if constexpr (i < 4) { act2(zin[0], 8 + 2 * i, pre); Bh[1][i] = cvt2(zin[0], 8 + 2 * i); }
                if constexpr (i == 3) zin[0] = mfma32(sel[0], Bh[0], zin[0]);
                if constexpr (i == 4) zin[0] = mfma32(sel[1], Bh[1], zin[0]);
                if constexpr (i >= 4 && i <= 7) { act2(zin[1], 2 * (i - 4), pre); Bh[2][i - 4] = cvt2(zin[1], 2 * (i - 4)); }
                if constexpr (i == 8) {
                    a24 = zin[1][8]; a25 = zin[1][9];
                    if constexpr (!decltype(pre)::value) { a24 = hidden_act_f(a24, slope); a25 = hidden_act_f(a25, slope); }
                    const f32x2 pr = {a24, a25};
                    mhi = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2));
                    zin[1] = mfma32(sel[0], Bh[2], zin[1]);
                }
                if constexpr (i == 9) {
                    Bl[0][0] = cvt2(zin[0], 0); Bl[0][1] = cvt2(zin[0], 2); Bl[0][2] = cvt2(zin[0], 4); Bl[0][3] = cvt2(zin[0], 6);
                }
                if constexpr (i == 10) {
                    Bl[1][0] = cvt2(zin[0], 8); Bl[1][1] = cvt2(zin[0], 10); Bl[1][2] = cvt2(zin[0], 12); Bl[1][3] = cvt2(zin[0], 14);
                }
                if constexpr (i == 11) {
                    const f32x2 rr = {a24 - __uint_as_float(mhi << 16), a25 - __uint_as_float(mhi & 0xffff0000u)};
                    const unsigned mlo = __builtin_bit_cast(unsigned, __builtin_convertvector(rr, bf16x2));
                    Bm = u32x4{mhi, mlo, mhi, 0u};
                }
                if constexpr (i == 12) {
                    Bl[2][0] = cvt2(zin[1], 0); Bl[2][1] = cvt2(zin[1], 2); Bl[2][2] = cvt2(zin[1], 4); Bl[2][3] = cvt2(zin[1], 6);
                }
                if constexpr (decltype(dot_c)::value) {           // previous node's dot: registers 2i, 2i+1 (13 slots: 0..12)
                    if constexpr (i < 13) {
                        sdot = fmaf(wout[2 * i], hidden_act_f(zprev[(2 * i) >> 4][(2 * i) & 15], slope), sdot);
                        sdot = fmaf(wout[2 * i + 1], hidden_act_f(zprev[(2 * i + 1) >> 4][(2 * i + 1) & 15], slope), sdot);
                    }
                }
                // next node's layer 1: in the free slots of the sections after the first (of the only section when NG = 1)
                constexpr int SI = decltype(sidx_c)::value;
                if constexpr (i >= 13 && (NG == 1 || SI >= 1)) {
                    constexpr int nsg = NG == 1 ? 7 : (NG - 1) * 7, sg = (NG == 1 ? 0 : SI - 1) * 7 + (i - 13);
                    constexpr int q_lo = sg * P32_NL / nsg, q_hi = (sg + 1) * P32_NL / nsg;
#pragma unroll
                    for (int qi = q_lo; qi < q_hi; ++qi)
                        zN[qi >> 4][qi & 15] = hidden_act_f(fmaf(w1x[qi], tkn, c[qi >> 4][qi & 15]), slope);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            // the layer's M-tile 0, registers 0..7: the next layer's first operand (last section: the next node's, from zN)
            if constexpr (!decltype(last)::value) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { act2(acc[0], 2 * e, std::false_type{}); Bh[0][e] = cvt2(acc[0], 2 * e); }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) Bh[0][e] = cvt2(zN[0], 2 * e);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        fb[0] = frag_ld(1, 0); fb[1] = frag_ld(1, 1); fb[2] = frag_ld(1, 2);
        constexpr std::true_type kYes{};
        constexpr std::false_type kNo{};
        // buffers: layer 1's output zN -> zA -> zB -> zC (-> zA): the last layer's output is never the first section's accumulator,
        // so that the previous node's dot can read it during that section
        f32x16 zN[2], zA[2], zB[2], zC[2];
#pragma unroll
        for (int qi = 0; qi < 32; ++qi) {
            zN[qi >> 4][qi & 15] = qi < P32_NL ? hidden_act_f(fmaf(w1x[qi], xv, c[qi >> 4][qi & 15]), slope) : 0.f;     // node 0: t = x
            zA[qi >> 4][qi & 15] = 0.f; zB[qi >> 4][qi & 15] = 0.f; zC[qi >> 4][qi & 15] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) Bh[0][e] = cvt2(zN[0], 2 * e);
        constexpr std::integral_constant<int, 0> k0{};
        constexpr std::integral_constant<int, 1> k1{};
        constexpr std::integral_constant<int, 2> k2{};
        constexpr std::integral_constant<int, 3> k3{};
        f32x16 (&zo)[2] = NG == 1 ? zB : NG == 2 ? zB : NG == 3 ? zC : zB;          // output of the last hidden layer (raw)
        auto finish_dot = [&](float wprev, bool first, bool lastn) {
            float sr = sdot;
            const unsigned w = __float_as_uint(sr);
            auto b2 = __builtin_amdgcn_permlane32_swap(w, w, false, false);
            sr = __uint_as_float(b2[0]) + __uint_as_float(b2[1]);
            const float f = out_act_f(sr, m.out_act);
            Facc = fmaf(wprev, maybe_inverse(f, a.inv_f), Facc);
            if (first) fxv = f;
            if (lastn) fx0v = f;
            sdot = 0.f;
        };
        float wprev = 0.f;            // quadrature weight of the previous node (0 before the first: its "dot" reads zeros)
        for (int k = 0; k <= nb; ++k) {
            const int kn = k < nb ? k + 1 : nb;
            const float tkn = __fadd_rn(x0v, __fmul_rn(dxv, a.ccs[kn] + 1.f) * 0.5f);        // (k + 1 >= 1: never node 0)
            if constexpr (NG == 1) {
                // (one hidden->hidden layer: its accumulator IS the previous output -- the dot runs after the section instead)
                section(kYes, kYes, kNo, k0, 1, 1, zN, zB, zN, tkn, zo);
#pragma unroll
                for (int qi = 0; qi < P32_NL; ++qi) sdot = fmaf(wout[qi], hidden_act_f(zo[qi >> 4][qi & 15], slope), sdot);
                finish_dot(a.ccw[k], k == 0, k == nb);
            } else {
                section(kYes, kNo, kYes, k0, 1, 2, zN, zA, zN, tkn, zo);
                finish_dot(wprev, k == 1, false);
                if constexpr (NG == 2) {
                    section(kNo, kYes, kNo, k1, 2, 1, zA, zB, zN, tkn, zo);
                } else {
                    section(kNo, kNo, kNo, k1, 2, 3, zA, zB, zN, tkn, zo);
                    if constexpr (NG == 3) {
                        section(kNo, kYes, kNo, k2, 3, 1, zB, zC, zN, tkn, zo);
                    } else {
                        section(kNo, kNo, kNo, k2, 3, 4, zB, zC, zN, tkn, zo);
                        section(kNo, kYes, kNo, k3, 4, 1, zC, zB, zN, tkn, zo);
                    }
                }
            }
            wprev = a.ccw[k];
        }
        if constexpr (NG > 1) {          // the last node's dot
#pragma unroll
            for (int qi = 0; qi < P32_NL; ++qi) sdot = fmaf(wout[qi], hidden_act_f(zo[qi >> 4][qi & 15], slope), sdot);
            finish_dot(wprev, false, true);
        }
    }
    // ---- hand the 32 results to the common epilogue in its (tile, lane p) form: lane p of group 0 owns point p of both tiles
    const int g = lane >> 4, p = lane & 15;
    float F2[2], fx2[2], fx02[2], dx2[2];
    bool ok2[2];
    long long q2[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        F2[pt] = __shfl(Facc, 16 * pt + p);
        fx2[pt] = __shfl(fxv, 16 * pt + p);
        fx02[pt] = __shfl(fx0v, 16 * pt + p);
        dx2[pt] = __shfl(dxv, 16 * pt + p);
        const long long q = ((long long)grp * 2 + pt) * 16 + p;
        ok2[pt] = live && q < a.NI;
        q2[pt] = q < a.NI ? q : a.NI - 1;
    }
    fwd_epilogue<2>(a, lds, F2, fx2, fx02, ok2, q2, dx2, live, 0, 1, wid, g, p);
}

// Launches the 32x32x16 kernel when the shape is the flagship one: >= 2 hidden layers, every one 48..51 wide (13 K-steps of 4
// in the fp32 numbering), bf16x3, one node range per wave.  UMNN_EUNSUPPORTED otherwise (the caller goes on with its own plan).
int umnn_launch_forward_p32(FwdArgs& a, const umnn_mlp* net, int nb_steps, hipStream_t stream) {
    const int L = a.m.n_linear - 1;
    if (L < 2 || L > 5) return UMNN_EUNSUPPORTED;
    for (int l = 1; l <= L; ++l)
        if (a.m.ks_in[l] != 13) return UMNN_EUNSUPPORTED;
    FwdBf16Args args;
    args.f = a;
    const size_t img_bytes = (size_t)(L - 1) * P32_IMG * sizeof(unsigned short);
    args.f.m.lds_off[L] = (int)((img_bytes / 4 + 3) & ~(size_t)3);
    const size_t lds_bytes = (size_t)args.f.m.lds_off[L] * sizeof(float);
    if (lds_bytes > 160 * 1024) return UMNN_EUNSUPPORTED;
    void (*kfn)(const FwdBf16Args) = L == 2 ? cc_fwd_p32_kernel<1> : L == 3 ? cc_fwd_p32_kernel<2> : L == 4 ? cc_fwd_p32_kernel<3>
                                                                                                            : cc_fwd_p32_kernel<4>;
    if (int rc = umnn_allow_lds((const void*)kfn, lds_bytes)) return rc;
    args.f.ns = 1;
    args.f.ngroups = (unsigned)((a.NI + 31) / 32);
    const unsigned nblk = (args.f.ngroups + UMNN_WAVES_PER_BLOCK - 1) / UMNN_WAVES_PER_BLOCK;
    umnn_prof_begin(stream);
    hipLaunchKernelGGL(kfn, dim3(nblk), dim3(UMNN_BLOCK), lds_bytes, stream, args);
    umnn_prof_end(stream, umnn_cc_forward_flops_per_integral(net, nb_steps) * (double)a.NI);
    umnn_note_launch("cc_fwd_bf16<32x32x16,PARTS=2,P=2,EXACT=1,LIVE=26,PIPE32>");
    return umnn_check(hipGetLastError(), "cc_fwd_p32 launch");
}
