// Backward of the quadrature on the bf16 matrix cores -- the same gradient convention and the same reference lines
// as cc_backward.hip, for the shapes the headline configurations use: every hidden layer 48..62 wide (4 tiles of 16)
// and at most 3 hidden->hidden layers.  One pass, persistent waves (one per SIMD, the whole register file), one tile
// of 16 integrals per wave.  Per quadrature node:
//   forward recompute  W_l fragments x split(a_l), THREE bf16 pieces / 6 cross terms: fp32-level accuracy (the sign of
//                      each activation is later read off its leading bf16 piece, which is kept for dW anyway).  This one
//                      must be that accurate: the backward needs the SIGN of every pre-activation (LeakyReLU/ReLU
//                      kink), and a recompute that is only 1e-5 accurate flips enough signs to move small-batch
//                      gradients by 5e-4 relative to the reference (measured with a 3-term recompute).
//   delta chain        W_l^T fragments x split(delta), two pieces / 3 terms (smooth: errors stay ~1e-5)
//   dW_l += delta_{l+1} (x) a_l   contracts over the 16 POINTS, which sit on the wrong lane axis.  The transposition
//                      is done by the matrix core itself: the packed bf16 fragments of a_l / delta (already there as
//                      B operands) are reused as A operands against a 0/1 selection fragment, which returns
//                      "feature on lane&15, points on (lane>>4, r)" -- exact, since it multiplies by one.  No LDS round
//                      trip (the fp32 kernel's transposes cost ~770 LDS cycles per layer and tile, CU-wide).
//                      The product then runs as 16 output tiles x 3 terms on the K = 16 instruction (16x16x16).
// What depends on an integral only once leaves as dc = sum_k delta_1 (finishing kernels of cc_backward.hip);
// every wave writes its partial d_theta slice (deterministic reduction, no atomics).
#pragma once
#include "cc_bf16.h"
#include "cc_bwd_shared.h"

constexpr int BT = 4;            // tiles of 16 features per hidden layer
constexpr int BKS = 2;           // K-steps of 32 features
constexpr int FRAG = 512;        // ushorts per fragment (64 lanes x 8)
// Pieces of the delta chain and the dW products: 2 (three cross terms, the default) or, for translation units built with
// -DUMNN_BWD_NPB=3 / a wrapper that defines it (cc_backward_front_p3.hip: the three-stage backward under bwd_precision = fp32),
// 3 (six cross terms, fp32-level like the recompute).  Everything below the argument structs lives in a namespace named after
// the piece count, so that both builds link into one library.
#ifndef UMNN_BWD_NPB
#define UMNN_BWD_NPB 2
#endif
#if UMNN_BWD_NPB == 3
#define UMNN_BWD_NS bwd_p3
#elif UMNN_BWD_NPB == 2
#define UMNN_BWD_NS bwd_p2
#else
#error "UMNN_BWD_NPB: 2 or 3"
#endif

struct BwdBf16Args {
    BwdArgs b;
    int off_fwd[UMNN_MAX_LINEAR];    // ushort offset of the forward fragment image of hidden layer l -> l+1 (NPF pieces)
    int off_tr[UMNN_MAX_LINEAR];     // ushort offset of the transposed image (delta_{l+1} -> delta_l, NPB pieces)
    // FRONT variants (cc_backward_front.hip): this kernel is the MIDDLE stage of a three-stage backward for nets whose first
    // hidden layer is wider than four tiles.  `b.m` then describes the net from hidden layer 2 on; the pre-activation of that
    // layer comes from HBM (written by the front-forward kernel) and its gradient goes back to HBM (read by the
    // front-backward kernel), both in this kernel's own register layout [tile][node][register][lane].
    const float* z2;                 // [chunk tiles][n+1][nl2][64]
    float* d2;                       // same shape
    const float* tz2;                // [chunk tiles][nl2][64]: d z2 / d t at node 0 (tangent pass of the g_fx term); nullable
    unsigned grp0;                   // first tile of the chunk (HBM fragments are indexed by grp - grp0)
    int nl2;                         // live registers of that layer: ceil((H2 + 1) / 4)
    int accumulate;                  // d_theta slices: 0 overwrite (first chunk), 1 add
    int off_trtile;                  // ushort offset in LDS of the per-wave transpose tiles (one-pass kernels)
    unsigned* scal;                  // launch scalars of the fp16-piece pipeline (Ws16Scal, cc_bwd_ws16_kernel.h); nullable
    const unsigned* only_if;         // non-null: the kernel runs only if *only_if != 0 (the queued fallback behind the fp16-piece pipeline)
};

namespace UMNN_BWD_NS {
constexpr int NPF = 3;           // bf16 pieces in the forward recompute (6 cross terms)
constexpr int NPB = UMNN_BWD_NPB; // bf16 pieces in the delta chain and the dW product (3 or 6 cross terms)

// fragment (tile t, K-step s, piece) of W (TRANSPOSED = false: rows = out features, K = in features incl. the
// constant-one feature / bias column) or of W^T (rows = in features, K = out features; no bias/constant entries:
// gradients must not flow through the constant feature).
template <bool TRANSPOSED, int NP>
__device__ __forceinline__ void stage_frag_image(const MlpDev& m, int l, unsigned short* img, int tid, int nthreads) {
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
#pragma unroll
        for (int part = 0; part < NP; ++part) {
            const unsigned short hb = bf16_rn_bits(v);
            img[(ts * NP + part) * FRAG + ln * 8 + j] = hb;
            v -= bf16_bits_to_f32(hb);
        }
    }
}

template <int NP>
struct BFrag { u32x4 v[BKS][NP]; };

template <int NRL, int NP>
__device__ __forceinline__ void split_regs(const f32x4 (&act)[BT], BFrag<NP>& bf) {
    constexpr int NLIVE = NRL > 0 ? NRL : 4 * BT;
#pragma unroll
    for (int s = 0; s < BKS; ++s) {
        unsigned q0[NP], q1[NP], q2[NP], q3[NP];
#pragma unroll
        for (int k2 = 0; k2 < NP; ++k2) q0[k2] = q1[k2] = q2[k2] = q3[k2] = 0u;
        if (8 * s + 0 < NLIVE) split_pair<NP>(act[2 * s][0], act[2 * s][1], q0);
        if (8 * s + 2 < NLIVE) split_pair<NP>(act[2 * s][2], act[2 * s][3], q1);
        if (8 * s + 4 < NLIVE) split_pair<NP>(act[2 * s + 1][0], act[2 * s + 1][1], q2);
        if (8 * s + 6 < NLIVE) split_pair<NP>(act[2 * s + 1][2], act[2 * s + 1][3], q3);
#pragma unroll
        for (int k2 = 0; k2 < NP; ++k2) bf.v[s][k2] = u32x4{q0[k2], q1[k2], q2[k2], q3[k2]};
    }
}

// acc[t] = sum over K-steps and the cross terms (wa + ba < NP) of frag(t, s, wa) * bf[s][ba]
template <int NP>
__device__ __forceinline__ void gemm_frags(const unsigned short* img, const BFrag<NP>& bf, f32x4 (&acc)[BT]) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};        // the first MFMA of every accumulator takes the literal 0 as C
#pragma unroll
    for (int s = 0; s < BKS; ++s) {
        u32x4 wf[BT][NP];
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int k2 = 0; k2 < NP; ++k2)
                wf[t][k2] = *reinterpret_cast<const u32x4*>(img + ((t * BKS + s) * NP + k2) * FRAG);
#pragma unroll
        for (int wa = 0; wa < NP; ++wa)
#pragma unroll
            for (int ba = 0; ba < NP; ++ba) {
                if (wa + ba >= NP) continue;
                const bool first = s == 0 && wa == 0 && ba == 0;
#pragma unroll
                for (int t = 0; t < BT; ++t) acc[t] = mfma_bf16(wf[t][wa], bf.v[s][ba], first ? zero : acc[t]);
            }
    }
}

// d act / d pre-activation of the feature held in register (t, r), read off the SIGN of its leading bf16 piece
// (a_l > 0  <=>  its round-to-nearest bf16 is > 0): 1 for a positive activation, `slope` otherwise -- the same
// convention as torch's LeakyReLU / ReLU backward (x > 0 ? 1 : slope).
template <int NP>
__device__ __forceinline__ float act_grad(const BFrag<NP>& a, int t, int r, float slope) {
    const unsigned u = a.v[t >> 1][0][(t & 1) * 2 + (r >> 1)];
    const int hi16 = (r & 1) ? (int)(u & 0xffff0000u) : (int)(u << 16);
    return hi16 > 0 ? 1.f : slope;
}

// Transposition on the matrix core.  `x` holds, as packed bf16 k-slots, the features of two tiles (2s, 2s+1) for the
// point lane&15 -- i.e. it is a valid A operand with rows = points.  Multiplying by the 0/1 fragment sel[h] (k-slot of
// feature 16(2s+h)+n  ->  column n) returns D[point][n]: lane (g, n) gets feature 16(2s+h)+n at points 4g..4g+3.
// Both pieces are transposed and re-packed as the 4 k-slots of a 16x16x16 operand (K = the 16 points of the tile).
__device__ __forceinline__ void transpose_pieces(const BFrag<NPB>& x, const u32x4 (&sel)[2], u32x2 (&out)[BT][NPB]) {
#pragma unroll
    for (int s = 0; s < BKS; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int part = 0; part < NPB; ++part) {
                const f32x4 tr = mfma_bf16(x.v[s][part], sel[h], f32x4{0.f, 0.f, 0.f, 0.f});
                const bf16x2 lo = __builtin_convertvector(f32x2{tr[0], tr[1]}, bf16x2);     // exact: values are bf16 already
                const bf16x2 hi = __builtin_convertvector(f32x2{tr[2], tr[3]}, bf16x2);
                out[2 * s + h][part] = u32x2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
            }
}

// The same transposition through a wave-private LDS tile and the transposing LDS read of gfx950 (ds_read_b64_tr_b16):
// every lane stores its packed k-slots as they are -- Mem[piece][point p][slot g*16 + s*8 + j], two 16-byte stores per piece --
// and reads back, for "slot tile" tau, the 4 bf16 at Mem[piece][point 4g + (p>>2)][16 tau + 4 (p&3) ..]; the instruction
// hands lane (g, n) element j = Mem[point 4g + j][16 tau + n]: slot 16 tau + n on lane n, points 4g..4g+3 in the k-slots
// of a 16x16x16 operand.  (Semantics probed in tools/ubench/trread.hip.)  Rows / columns of the dW tiles then run over
// SLOTS, slot sigma <-> feature 16 (2 s + (j>>2)) + 4 (j&3) + G with G = sigma>>4, s = (sigma>>3)&1, j = sigma&7: a
// permutation that only the final write-out of the accumulators has to know (slot_feature).  No MFMA, no v_cvt_pk.
constexpr int TRS = 72;          // ushorts per point row of the tile: 64 slots + 8 padding (144 B: 16-byte aligned rows)
typedef short s16x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int slot_feature(int sigma) {
    const int G = sigma >> 4, ss = (sigma >> 3) & 1, j = sigma & 7;
    return 16 * (2 * ss + (j >> 2)) + 4 * (j & 3) + G;
}
__device__ __forceinline__ void tr_tile_store(const BFrag<NPB>& x, unsigned short* tile, int g, int p) {
#pragma unroll
    for (int part = 0; part < NPB; ++part)
#pragma unroll
        for (int s = 0; s < BKS; ++s)
            *reinterpret_cast<u32x4*>(tile + (part * 16 + p) * TRS + g * 16 + s * 8) = x.v[s][part];
}
__device__ __forceinline__ void tr_tile_read(const unsigned short* tile, int g, int p, u32x2 (&out)[BT][NPB]) {
#pragma unroll
    for (int tau = 0; tau < BT; ++tau)
#pragma unroll
        for (int part = 0; part < NPB; ++part) {
            const unsigned short* src = tile + (part * 16 + 4 * g + (p >> 2)) * TRS + 16 * tau + 4 * (p & 3);
            const s16x4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4v __attribute__((address_space(3)))*)(src));
            out[tau][part] = __builtin_bit_cast(u32x2, v);
        }
}
__device__ __forceinline__ void transpose_pieces_lds(const BFrag<NPB>& x, unsigned short* tile, int g, int p,
                                                     u32x2 (&out)[BT][NPB]) {
    tr_tile_store(x, tile, g, p);
    tr_tile_read(tile, g, p, out);
}

// LH = number of hidden layers (compile-time: the layer loops are unrolled so that every register array is
// statically indexed); NACC = LH - 1 hidden->hidden layers, all accumulated in this one pass (l_lo = 1).
template <int LH, bool EDGE, int NRL, bool FRONT = false>
__global__ __launch_bounds__(UMNN_BLOCK, 1) void cc_bwd_bf16_kernel(const BwdBf16Args args) {
    constexpr int NLIVE = NRL > 0 ? NRL : 4 * BT;
    constexpr int NACC = LH - 1;
    constexpr int NA = NACC > 0 ? NACC : 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const BwdArgs& a = args.b;
    const MlpDev& m = a.m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, p = lane & 15;
    constexpr int L = LH;
    const int H1 = m.width[1], HL = m.width[L];
    const int E = a.E, d = a.d, n = a.n;
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    unsigned short* lds16 = reinterpret_cast<unsigned short*>(lds);

    for (int l = 1; l < L; ++l) {
        stage_frag_image<false, NPF>(m, l, lds16 + args.off_fwd[l], tid, blockDim.x);
        stage_frag_image<true, NPB>(m, l, lds16 + args.off_tr[l], tid, blockDim.x);
    }
    __syncthreads();
    const unsigned short* frag_base = lds16 + lane * 8;
    constexpr bool TRL = true;                         // dW operands transposed through LDS (ds_read_b64_tr_b16)
    unsigned short* tr_tile = lds16 + args.off_trtile + wid * (NPB * 16 * TRS);

    // selection fragments of the matrix-core transpose: lane (g, n) sets k-slot 4h + (n>>2) to 1.0 iff (n&3) == g
    u32x4 sel[2];
    {
        const unsigned one_lo = 0x3f80u, one_hi = 0x3f800000u;      // bf16 1.0 in the low / high half of a dword
        const int slot = p >> 2;                                    // k-slot inside the 4 slots of tile h
        const unsigned w0 = (p & 3) == g ? (slot == 0 ? one_lo : slot == 1 ? one_hi : 0u) : 0u;
        const unsigned w1 = (p & 3) == g ? (slot == 2 ? one_lo : slot == 3 ? one_hi : 0u) : 0u;
        sel[0] = u32x4{w0, w1, 0u, 0u};
        sel[1] = u32x4{0u, 0u, w0, w1};
    }

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
                w1x[t][r] = (!FRONT && f < H1) ? W0[f * (1 + E)] : 0.f;
                wout[t][r] = f < HL ? WL[f] : (f == HL ? bL : 0.f);
            }
    }

    f32x4 dW[NA][BT][BT];
#pragma unroll
    for (int j = 0; j < NA; ++j)
#pragma unroll
        for (int to = 0; to < BT; ++to)
#pragma unroll
            for (int ti = 0; ti < BT; ++ti) dW[j][to][ti] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 dW1x[BT], dwo[BT];
#pragma unroll
    for (int t = 0; t < BT; ++t) { dW1x[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dwo[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const unsigned wave_global = blockIdx.x * (blockDim.x >> 6) + wid;
    const unsigned nwaves = gridDim.x * (blockDim.x >> 6);

    const unsigned nsp = a.ns > 1 ? (unsigned)a.ns : 1u;       // node-range split (small batches), see BwdArgs::ns
    for (unsigned item = wave_global; item < a.ngroups * nsp; item += nwaves) {
        const unsigned gl = item / nsp, part = item - gl * nsp;
        const unsigned grp = (FRONT ? args.grp0 : 0u) + gl;               // (FRONT: a.ngroups counts the tiles of the chunk)
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
        if constexpr (FRONT) {
#pragma unroll
            for (int t = 0; t < BT; ++t) c[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
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

        f32x4 dcs[BT];
#pragma unroll
        for (int t = 0; t < BT; ++t) dcs[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        float fxv = 0.f, fx0v = 0.f, dfdt = 0.f;
        // FRONT: this tile's fragments in HBM, [node][register][lane]; the next node's pre-activations are fetched one
        // node ahead (a global load costs ~1 us here against ~5 us of work per node)
        const int nl2 = NRL > 0 ? NRL : args.nl2;      // (LIVE=13 variants: a compile-time count, the guards below fold away)
        const size_t frag0 = FRONT ? (size_t)(grp - args.grp0) * (size_t)(n + 1) * nl2 * 64 + lane : 0;
        // (FRONT) z_2 of node k, register j: unconditional loads (register index clamped, value masked afterwards) -- guarded ones
        // become a branch and a memory wait per register
        auto ld_z2 = [&](int k, int t, int r) {
            const int j = 4 * t + r, jj = j < nl2 ? j : nl2 - 1;
            const float v = args.z2[frag0 + ((size_t)k * nl2 + jj) * 64];
            return j < nl2 ? v : 0.f;
        };
        f32x4 znext[BT];
        if constexpr (FRONT) {
#pragma unroll
            for (int t = 0; t < BT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) znext[t][r] = 4 * t + r < NLIVE ? ld_z2(k_lo, t, r) : 0.f;
        }

        for (int k = k_lo; k < k_hi; ++k) {
            const float u = a.ccs[k] + 1.f;
            const float wk = a.ccw[k];
            const float tk = k == 0 ? xv : __fadd_rn(x0v, __fmul_rn(dxv, u) * 0.5f);
            f32x4 act[BT];
            BFrag<NPB> asave[NA];            // packed (hi, next) pieces of a_l for the layers whose dW this pass owns
            // ---------------- forward recompute (6 cross terms) ----------------
            if constexpr (FRONT) {
#pragma unroll
                for (int t = 0; t < BT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) act[t][r] = 4 * t + r < NLIVE ? hidden_act_f(znext[t][r], slope) : 0.f;
                {
                    const int kn = k + 1 < k_hi ? k + 1 : k;        // (the last node re-reads itself: no branch)
#pragma unroll
                    for (int t = 0; t < BT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (4 * t + r < NLIVE) znext[t][r] = ld_z2(kn, t, r);
                }
            } else {
#pragma unroll
                for (int t = 0; t < BT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        act[t][r] = 4 * t + r < NLIVE ? hidden_act_f(fmaf(w1x[t][r], tk, c[t][r]), slope) : 0.f;
            }
#pragma unroll
            for (int l = 1; l < L; ++l) {
                BFrag<NPF> bf;
                split_regs<NRL, NPF>(act, bf);
#pragma unroll
                for (int s = 0; s < BKS; ++s)
#pragma unroll
                    for (int k2 = 0; k2 < NPB; ++k2) asave[l - 1].v[s][k2] = bf.v[s][k2];
                f32x4 acc[BT];
                gemm_frags<NPF>(frag_base + args.off_fwd[l], bf, acc);
#pragma unroll
                for (int t = 0; t < BT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) act[t][r] = 4 * t + r < NLIVE ? hidden_act_f(acc[t][r], slope) : 0.f;
            }
            float sdot = 0.f;
#pragma unroll
            for (int t = 0; t < BT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * t + r < NLIVE) sdot = fmaf(wout[t][r], act[t][r], sdot);
            sdot = group_allreduce(sdot);
            const float f = out_act_f(sdot, m.out_act);
            const float fp = out_grad_f(sdot, m.out_act);
            if (k == 0) fxv = f;
            if (k == n) fx0v = f;

            // ---------------- tangent pass at node 0: d f / d x for the g_fx term ----------------
            if (EDGE && k == 0 && a.gfx) {
                f32x4 ta[BT];
#pragma unroll
                for (int t = 0; t < BT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // d(pre-activation)/dt of this kernel's first layer: the x-column of W1, or (FRONT) what the front
                        // kernel propagated through the wide first hidden layer
                        float dz = w1x[t][r];
                        if constexpr (FRONT)
                            dz = (4 * t + r < nl2 && args.tz2)
                                     ? args.tz2[((size_t)(grp - args.grp0) * nl2 + 4 * t + r) * 64 + lane] : 0.f;
                        ta[t][r] = 4 * t + r < NLIVE ? dz * act_grad(asave[0], t, r, slope) : 0.f;
                    }
#pragma unroll
                for (int l = 1; l < L; ++l) {
                    BFrag<NPF> bf;
                    split_regs<NRL, NPF>(ta, bf);
                    f32x4 tz[BT];
                    gemm_frags<NPF>(frag_base + args.off_fwd[l], bf, tz);
#pragma unroll
                    for (int t = 0; t < BT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // layer l+1: its activation is a saved fragment, except the last layer whose act is still live
                            const float fac = l + 1 < L ? act_grad(asave[l + 1 < L ? l : 0], t, r, slope)
                                                        : (act[t][r] > 0.f ? 1.f : slope);
                            ta[t][r] = 4 * t + r < NLIVE ? tz[t][r] * fac : 0.f;
                        }
                }
                float ds = 0.f;
#pragma unroll
                for (int t = 0; t < BT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ds = fmaf(wout[t][r], ta[t][r], ds);
                dfdt = fp * group_allreduce(ds);
            }

            // ---------------- backward sweep ----------------
            const float invs = a.inv_f ? -__frcp_rn(f * f) : 1.f;
            const float cot = fmaf(cotbase * invs, wk, k == 0 ? gfxv : 0.f);
            const float dout = cot * fp;
            f32x4 delta[BT];
#pragma unroll
            for (int t = 0; t < BT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (EDGE) dwo[t][r] = fmaf(dout, act[t][r], dwo[t][r]);
                    delta[t][r] = dout * wout[t][r] * (act[t][r] > 0.f ? 1.f : slope);
                }
#pragma unroll
            for (int l = L - 1; l >= 1; --l) {
                u32x2 dT[BT][NPB], aT[BT][NPB];
                // (LDS route: a_l has been ready since the forward pass -- its round trip is issued first, so that it runs
                // under the VALU work that packs delta)
                if constexpr (TRL) transpose_pieces_lds(asave[l - 1], tr_tile, g, p, aT);
                BFrag<NPB> bd;
                split_regs<NRL, NPB>(delta, bd);
                // the GEMM through W_l^T first: it is the critical path to the next layer; the dW product below only feeds
                // accumulators and can run under the vector work that follows
                f32x4 nd[BT];
                if constexpr (TRL) tr_tile_store(bd, tr_tile, g, p);
                gemm_frags<NPB>(frag_base + args.off_tr[l], bd, nd);
                {
                    if constexpr (TRL) {
                        tr_tile_read(tr_tile, g, p, dT);
                    } else {
                        transpose_pieces(bd, sel, dT);
                        transpose_pieces(asave[l - 1], sel, aT);
                    }
#pragma unroll
                    for (int wa = 0; wa < NPB; ++wa)
#pragma unroll
                        for (int ba = 0; ba < NPB; ++ba) {
                            if (wa + ba >= NPB) continue;
#pragma unroll
                            for (int to = 0; to < BT; ++to)
#pragma unroll
                                for (int ti = 0; ti < BT; ++ti)
                                    dW[l - 1][to][ti] = mfma_bf16_k16(dT[to][wa], aT[ti][ba], dW[l - 1][to][ti]);
                        }
                }
#pragma unroll
                for (int t = 0; t < BT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        delta[t][r] = 4 * t + r < NLIVE ? nd[t][r] * act_grad(asave[l - 1], t, r, slope) : 0.f;
            }
            if constexpr (FRONT) {
#pragma unroll
                for (int t = 0; t < BT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * t + r < NLIVE && 4 * t + r < nl2)
                            args.d2[frag0 + ((size_t)k * nl2 + 4 * t + r) * 64] = delta[t][r];
            } else if (EDGE) {
#pragma unroll
                for (int t = 0; t < BT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dcs[t][r] += delta[t][r];
                        dW1x[t][r] = fmaf(delta[t][r], tk, dW1x[t][r]);
                    }
            }
        }

        if (EDGE && ok) {
#pragma unroll
            for (int t = 0; t < BT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = feat_of(t, r, g);
                    if (!FRONT && f < H1) a.dc[(size_t)part * a.NI * H1 + q * H1 + f] = dcs[t][r];
                }
            if (g == 0) {
                if (a.dx && k_lo == 0) io_st(a.dx, q, fmaf(gfxv, dfdt, fxv * gv), a.x_bf16);
                if (a.dx0 && k_hi == n + 1) io_st(a.dx0, q, -fx0v * gv, a.x_bf16);
            }
        }
    }

    // ---------------- write this wave's partial d_theta ----------------
    // dW tile (to, ti): row 4g+r is the row (lane&15 = 4g+r) of the delta^T operand = feature 16 to + 4g+r (natural, the
    // transpose delivers features in natural order inside a tile); column lane&15 likewise feature 16 ti + (lane&15).
    float* part = a.partials + (size_t)wave_global * a.n_params;
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
        const int l = 1 + j;
        {
            const int Hin = m.width[l], Hout = m.width[l + 1];
#pragma unroll
            for (int to = 0; to < BT; ++to)
#pragma unroll
                for (int ti = 0; ti < BT; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int fo = TRL ? slot_feature(16 * to + 4 * g + r) : 16 * to + 4 * g + r;
                        const int fi = TRL ? slot_feature(16 * ti + (lane & 15)) : 16 * ti + (lane & 15);
                        if (fo < Hout) {
                            const int idx = fi < Hin ? a.poffW[l] + fo * Hin + fi : (fi == Hin ? a.poffb[l] + fo : -1);
                            if (idx >= 0) part[idx] = (FRONT && args.accumulate ? part[idx] : 0.f) + dW[j][to][ti][r];
                        }
                    }
        }
    }
    if (EDGE) {
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v1 = dW1x[t][r], v2 = dwo[t][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { v1 += __shfl_xor(v1, o); v2 += __shfl_xor(v2, o); }
                const int f = feat_of(t, r, g);
                if (p == 0) {
                    if (!FRONT && f < H1) part[a.poffW[0] + f * (1 + E)] = v1;
                    const int idx = f < HL ? a.poffW[L] + f : (f == HL ? a.poffb[L] : -1);
                    if (idx >= 0) part[idx] = (FRONT && args.accumulate ? part[idx] : 0.f) + v2;
                }
            }
    }
}

}  // namespace UMNN_BWD_NS
using namespace UMNN_BWD_NS;
