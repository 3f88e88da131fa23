// Backward of the quadrature for nets whose FIRST hidden layer is wider than four 16-feature tiles and whose other hidden
// layers fit four (MNISTExperiment's integrand 31-100-50-50-50-50-1, /root/reference MNISTExperiment.py:18,238; same
// reference lines as cc_backward.hip for the arithmetic: ParallelNeuralIntegral.py:66-94,110-123).
//
// One pass cannot hold this shape: the dW accumulators alone would be 28 + 3 x 16 tiles = 304 registers next to a live
// set of ~310, and the fragment images 190 KB of 160.  The register-spilling generic kernels were ~100x slower than
// they should be, so round 1 sent this family to a materialised ATen chain.  Here the pass is cut in three at the
// 50-wide pre-activation of hidden layer 2 -- and the cut goes through HBM, which this GPU has 288 GB of:
//
//   A  cc_front_fwd_kernel   per tile and node: a1 = act(W1[:,0] t + c) (T1 tiles, VALU), z2 = G1 a1 + b2 with six bf16 cross
//                            terms (fp32-level: the signs of z2 decide LeakyReLU kinks downstream); z2 is written to HBM in
//                            the register layout of the middle kernel, [tile][node][register][lane]: every store is a
//                            coalesced 256-B row.  Also d z2 / d t at node 0 (tangent of the g_fx term).
//   B  cc_bwd_bf16_kernel<LH, EDGE, NRL, FRONT=true>   the flagship one-pass kernel on the net from hidden layer 2 on: reads z2
//                            (one node ahead of its use), does everything it does for a 4-tile net (recompute, delta chain, dW
//                            of the 50x50 layers, output layer, Leibniz terms) and writes delta_2 = dL/dz2 back to HBM.
//   C  cc_front_bwd_kernel   per tile and node: recomputes a1, reads delta_2, accumulates dG1 += delta_2 (x) a1 (28 tiles on the
//                            K=16 MFMA, operands transposed by the matrix core as in the flagship kernel), back-propagates
//                            delta_1 = (G1^T delta_2) act'(z1), and leaves dc = sum_k delta_1 and dW1[:,0] like the one-pass
//                            kernels do; the usual finishing kernels (d_h, dW1[:,1:], d_theta reduction) follow.
//
// Scratch: 2 x (n+1) x ceil((H2+1)/4) x 256 B per tile (339 KB at n = 50); tiles are processed in chunks that fit the
// scratch the workspace provides (2 GiB), every wave's d_theta slice accumulating across chunks.  HBM traffic per
// tile-node 4 x 3.3 KB against ~5 us of matrix work: three orders of magnitude below the bandwidth roof.
//
// This file builds twice: as it is (two bf16 pieces in the delta chain and the dW products: bwd_precision = bf16x3), and through
// cc_backward_front_p3.hip with UMNN_BWD_NPB = 3 (three pieces / six cross terms everywhere: fp32-level, what bwd_precision = fp32
// gets for this family -- no workgroup pipeline there, the one-pass middle kernel).
#include "cc_bwd_bf16_kernel.h"
#if UMNN_BWD_NPB == 2
#include "cc_bwd_ws16_kernel.h"      // (the workgroup pipelines; the fp16 helpers of stage A on fp16 pieces)
#endif

namespace UMNN_BWD_NS {
struct FrontArgs {
    BwdArgs b;              // the FULL net
    float* z2;
    float* d2;
    float* tz2;             // nullable (no g_fx)
    unsigned grp0;          // first tile of the chunk; b.ngroups = tiles in the chunk
    int nl2;                // live registers of hidden layer 2
    int accumulate;
    const unsigned* only_if;    // non-null: stage A runs only if *only_if != 0 (queued behind its fp16 build as the overflow fallback)
    int tz_only;                // z_2 came with the call (BwdArgs::z2_saved): stage A only forms the tangent element of node 0 (g_fx term)
};

// fragment image of G1 = W[1] (hidden 1 -> hidden 2).  TRANSPOSED = false: rows = hidden-2 features (BT tiles, incl. the
// constant-one row), K = hidden-1 features: KSF = T1/2 full K-steps + a half K-step (K = 16) when T1 is odd, laid out
// behind the full ones.  TRANSPOSED = true: rows = hidden-1 features (T1 tiles), K = hidden-2 features (two K-steps), no
// bias / constant entries (gradients do not flow through the constant feature).
template <int T1, bool TRANSPOSED, int NP>
__device__ __forceinline__ void stage_g1_image(const MlpDev& m, unsigned short* img, int tid, int nthreads) {
    constexpr int KSF = T1 / 2;
    const int Hin = m.width[1], Hout = m.width[2];
    const float* __restrict__ W = m.W[1];
    const float* __restrict__ b = m.b[1];
    constexpr int ROWT = TRANSPOSED ? T1 : BT;
    constexpr int KS = TRANSPOSED ? BKS : KSF;
    for (int idx = tid; idx < ROWT * KS * FRAG; idx += nthreads) {
        const int j = idx & 7, ln = (idx >> 3) & 63, ts = idx >> 9;
        const int s = ts % KS, t = ts / KS;
        const int frow = fout_of(t, ln & 15);
        const int fk = feat_of(2 * s + (j >> 2), j & 3, ln >> 4);
        float v = 0.f;
        if (!TRANSPOSED) {
            if (frow < Hout) v = fk < Hin ? W[frow * Hin + fk] : (fk == Hin ? b[frow] : 0.f);
            else if (frow == Hout && fk == Hin) v = 1.f;
        } else {
            if (fk < Hout && frow < Hin) v = W[fk * Hin + frow];
        }
#pragma unroll
        for (int part = 0; part < NP; ++part) {
            const unsigned short hb = bf16_rn_bits(v);
            img[(ts * NP + part) * FRAG + ln * 8 + j] = hb;
            v -= bf16_bits_to_f32(hb);
        }
    }
    if (!TRANSPOSED && (T1 & 1)) {
        unsigned short* himg = img + BT * KSF * NP * FRAG;
        for (int idx = tid; idx < BT * 256; idx += nthreads) {
            const int j = idx & 3, ln = (idx >> 2) & 63, t = idx >> 8;
            const int frow = fout_of(t, ln & 15);
            const int fk = feat_of(T1 - 1, j, ln >> 4);
            float v = 0.f;
            if (frow < Hout) v = fk < Hin ? W[frow * Hin + fk] : (fk == Hin ? b[frow] : 0.f);
            else if (frow == Hout && fk == Hin) v = 1.f;
#pragma unroll
            for (int part = 0; part < NP; ++part) {
                const unsigned short hb = bf16_rn_bits(v);
                himg[(t * NP + part) * 256 + ln * 4 + j] = hb;
                v -= bf16_bits_to_f32(hb);
            }
        }
    }
}

// hoisted first-layer term of one tile: c = W1[:,1:] h + b1 (and the constant-one feature), fp32 MFMA, T1 tiles
template <int T1>
__device__ __forceinline__ void front_prologue(const MlpDev& m, const IoView& hb, int E, int d, int g, int p, f32x4 (&c)[T1]) {
    const int H1 = m.width[1];
    const float* __restrict__ W0 = m.W[0];
    const float* __restrict__ b0 = m.b[0];
#pragma unroll
    for (int t = 0; t < T1; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = feat_of(t, r, g);
            c[t][r] = f < H1 ? b0[f] : (f == H1 ? 1.f : 0.f);
        }
    item_embedding_gemm<T1, 4>(hb, W0, H1, E, d, g, p, c);
}

// out[BT] = G1-fragments x split(act[T1]) with NP pieces / the cross terms wa + ba < NP
template <int T1, int NP>
__device__ __forceinline__ void front_gemm(const unsigned short* base, int lane, const f32x4 (&act)[T1], f32x4 (&out)[BT]) {
    const unsigned short* img = base + lane * 8;            // full fragments: 8 bf16 per lane; half fragments below: 4
    constexpr int KSF = T1 / 2;
    u32x4 bf[KSF][NP];
#pragma unroll
    for (int s = 0; s < KSF; ++s) {
        unsigned q0[NP], q1[NP], q2[NP], q3[NP];
        split_pair<NP>(act[2 * s][0], act[2 * s][1], q0);
        split_pair<NP>(act[2 * s][2], act[2 * s][3], q1);
        split_pair<NP>(act[2 * s + 1][0], act[2 * s + 1][1], q2);
        split_pair<NP>(act[2 * s + 1][2], act[2 * s + 1][3], q3);
#pragma unroll
        for (int k2 = 0; k2 < NP; ++k2) bf[s][k2] = u32x4{q0[k2], q1[k2], q2[k2], q3[k2]};
    }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KSF; ++s) {
        u32x4 wf[BT][NP];
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int k2 = 0; k2 < NP; ++k2)
                wf[t][k2] = *reinterpret_cast<const u32x4*>(img + ((t * KSF + s) * NP + k2) * FRAG);
#pragma unroll
        for (int wa = 0; wa < NP; ++wa)
#pragma unroll
            for (int ba = 0; ba < NP; ++ba) {
                if (wa + ba >= NP) continue;
                const bool first = s == 0 && wa == 0 && ba == 0;
#pragma unroll
                for (int t = 0; t < BT; ++t) out[t] = mfma_bf16(wf[t][wa], bf[s][ba], first ? zero : out[t]);
            }
    }
    if constexpr (T1 & 1) {
        u32x2 hb[NP];
        {
            unsigned q0[NP], q1[NP];
            split_pair<NP>(act[T1 - 1][0], act[T1 - 1][1], q0);
            split_pair<NP>(act[T1 - 1][2], act[T1 - 1][3], q1);
#pragma unroll
            for (int k2 = 0; k2 < NP; ++k2) hb[k2] = u32x2{q0[k2], q1[k2]};
        }
        const unsigned short* himg = base + BT * KSF * NP * FRAG + lane * 4;
        u32x2 wh[BT][NP];
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int k2 = 0; k2 < NP; ++k2) wh[t][k2] = *reinterpret_cast<const u32x2*>(himg + (t * NP + k2) * 256);
#pragma unroll
        for (int wa = 0; wa < NP; ++wa)
#pragma unroll
            for (int ba = 0; ba < NP; ++ba) {
                if (wa + ba >= NP) continue;
#pragma unroll
                for (int t = 0; t < BT; ++t) out[t] = mfma_bf16_k16(wh[t][wa], hb[ba], out[t]);
            }
    }
}

// ---------------------------------------------------------------------------------------------- stage A
// NL2 > 0: the live-register count of hidden layer 2 at compile time (13 = widths 48..51) -- the per-register guards of the HBM
// fragments then fold away (as runtime conditions each of them is a branch); 0 = read it from the arguments.
template <int T1, int NL2>
__global__ __launch_bounds__(UMNN_BLOCK, 1) void cc_front_fwd_kernel(const FrontArgs fa) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const BwdArgs& a = fa.b;
    const MlpDev& m = a.m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, p = lane & 15;
    const int H1 = m.width[1];
    const int E = a.E, d = a.d, n = a.n, nl2 = NL2 > 0 ? NL2 : fa.nl2;
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    unsigned short* lds16 = reinterpret_cast<unsigned short*>(lds);
    if (fa.only_if && *fa.only_if == 0) return;      // queued as the fallback of the fp16 build: nothing overflowed
    stage_g1_image<T1, false, NPF>(m, lds16, tid, blockDim.x);
    __syncthreads();

    float w1x[T1][4];
    {
        const float* __restrict__ W0 = m.W[0];
#pragma unroll
        for (int t = 0; t < T1; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = feat_of(t, r, g);
                w1x[t][r] = f < H1 ? W0[f * (1 + E)] : 0.f;
            }
    }
    const unsigned wave_global = blockIdx.x * (blockDim.x >> 6) + wid;
    const unsigned nwaves = gridDim.x * (blockDim.x >> 6);
    for (unsigned item = wave_global; item < a.ngroups; item += nwaves) {
        const unsigned grp = fa.grp0 + item;
        const long long q = (long long)grp * 16 + p;
        const long long qq = q < a.NI ? q : a.NI - 1;
        const float xv = io_ld(a.x, qq, a.x_bf16);
        const float x0v = a.x0 ? io_ld(a.x0, qq, a.x_bf16) : 0.f;
        const float dxv = xv - x0v;
        const long long bi = qq / d;
        const IoView hb = IoView{a.h, a.h_bf16} + (bi * ((long long)E * d) + (qq - bi * d));
        f32x4 c[T1];
        front_prologue<T1>(m, hb, E, d, g, p, c);
        const size_t frag0 = (size_t)item * (size_t)(n + 1) * nl2 * 64 + lane;
        for (int k = 0; k <= (fa.tz_only ? 0 : n); ++k) {
            const float u = a.ccs[k] + 1.f;
            const float tk = k == 0 ? xv : __fadd_rn(x0v, __fmul_rn(dxv, u) * 0.5f);
            f32x4 z1[T1], act[T1];
#pragma unroll
            for (int t = 0; t < T1; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    z1[t][r] = fmaf(w1x[t][r], tk, c[t][r]);
                    act[t][r] = hidden_act_f(z1[t][r], slope);
                }
            f32x4 z2[BT];
            front_gemm<T1, NPF>(lds16, lane, act, z2);
#pragma unroll
            for (int t = 0; t < BT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * t + r < nl2 && !fa.tz_only) fa.z2[frag0 + ((size_t)k * nl2 + 4 * t + r) * 64] = z2[t][r];
            if (k == 0 && fa.tz2) {       // d z2 / d t = G1 (W1[:,0] . act'(z1)): linear in the tangent, no bias
                f32x4 ta[T1];
#pragma unroll
                for (int t = 0; t < T1; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ta[t][r] = w1x[t][r] * (z1[t][r] > 0.f ? 1.f : slope);
                f32x4 tz[BT];
                front_gemm<T1, NPF>(lds16, lane, ta, tz);
#pragma unroll
                for (int t = 0; t < BT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * t + r < nl2) fa.tz2[((size_t)item * nl2 + 4 * t + r) * 64 + lane] = tz[t][r];
            }
        }
    }
}

#if UMNN_BWD_NPB == 2
// ---------------------------------------------------------------------------------------------- stage A on fp16 pieces
// The same stage with G1 and a_1 as two fp16 pieces each (cc_bwd_ws16_kernel.h on why three cross terms of 11-bit pieces are
// fp32-level): 42 matrix instructions per tile-node instead of 84, a four-instruction split per pair instead of eight.  The low
// piece of G1 is stored times 2^11 (W16_LOSCALE: a plain remainder of a weight below 2^-3 would be subnormal), its products in
// their own accumulators.  Used together with the fp16 middle stage only: an activation beyond the fp16 range makes z_2 inf / NaN,
// the middle stage's checks see that and raise the launch flag, and the launcher has the bf16 builds of BOTH stages queued
// behind them with "run only if the flag is set".
template <int T1>
__device__ __forceinline__ void stage_g1_image16(const MlpDev& m, unsigned short* img, int tid, int nthreads) {
    constexpr int KSF = T1 / 2;
    const int Hin = m.width[1], Hout = m.width[2];
    const float* __restrict__ W = m.W[1];
    const float* __restrict__ b = m.b[1];
    auto wv = [&](int frow, int fk) {
        float v = 0.f;
        if (frow < Hout) v = fk < Hin ? W[frow * Hin + fk] : (fk == Hin ? b[frow] : 0.f);
        else if (frow == Hout && fk == Hin) v = 1.f;
        return v;
    };
    for (int idx = tid; idx < BT * KSF * FRAG; idx += nthreads) {
        const int j = idx & 7, ln = (idx >> 3) & 63, ts = idx >> 9;
        const int s = ts % KSF, t = ts / KSF;
        const float v = wv(fout_of(t, ln & 15), feat_of(2 * s + (j >> 2), j & 3, ln >> 4));
        const _Float16 hi = (_Float16)v, lo = (_Float16)((v - (float)hi) * W16_LOSCALE);
        img[(ts * 2 + 0) * FRAG + ln * 8 + j] = __builtin_bit_cast(unsigned short, hi);
        img[(ts * 2 + 1) * FRAG + ln * 8 + j] = __builtin_bit_cast(unsigned short, lo);
    }
    if (T1 & 1) {
        unsigned short* himg = img + BT * KSF * 2 * FRAG;
        for (int idx = tid; idx < BT * 256; idx += nthreads) {
            const int j = idx & 3, ln = (idx >> 2) & 63, t = idx >> 8;
            const float v = wv(fout_of(t, ln & 15), feat_of(T1 - 1, j, ln >> 4));
            const _Float16 hi = (_Float16)v, lo = (_Float16)((v - (float)hi) * W16_LOSCALE);
            himg[(t * 2 + 0) * 256 + ln * 4 + j] = __builtin_bit_cast(unsigned short, hi);
            himg[(t * 2 + 1) * 256 + ln * 4 + j] = __builtin_bit_cast(unsigned short, lo);
        }
    }
}
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_f16_k16(u32x2 a, u32x2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h16x4, a), __builtin_bit_cast(h16x4, b), c, 0, 0, 0);
}
// out[BT] = G1 x act[T1]: hi.hi + hi.lo in `out`, (2^11 lo).hi in its own accumulators, added back times 2^-11
// (act(t, r): the B operand's value of feature register r of tile t, produced K-step by K-step -- the whole of a_1 is never live)
template <int T1, class ActFn>
__device__ __forceinline__ void front_gemm16(const unsigned short* base, int lane, ActFn&& act, f32x4 (&out)[BT]) {
    const unsigned short* img = base + lane * 8;
    constexpr int KSF = T1 / 2;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc2[BT];
#pragma unroll
    for (int s = 0; s < KSF; ++s) {
        u32x4 bf[KSF][2];
        {
            unsigned q[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x0 = act(2 * s + (j >> 1), 2 * (j & 1)), x1 = act(2 * s + (j >> 1), 2 * (j & 1) + 1);
                q[j][0] = h16_split_stage(x0, x1);
                q[j][1] = h16_split_last(x0, x1);
            }
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) bf[s][k2] = u32x4{q[0][k2], q[1][k2], q[2][k2], q[3][k2]};
        }
        u32x4 wf[BT][2];
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) wf[t][k2] = *reinterpret_cast<const u32x4*>(img + ((t * KSF + s) * 2 + k2) * FRAG);
#pragma unroll
        for (int t = 0; t < BT; ++t) out[t] = mfma_f16(wf[t][0], bf[s][0], s == 0 ? zero : out[t]);
#pragma unroll
        for (int t = 0; t < BT; ++t) out[t] = mfma_f16(wf[t][0], bf[s][1], out[t]);
#pragma unroll
        for (int t = 0; t < BT; ++t) acc2[t] = mfma_f16(wf[t][1], bf[s][0], s == 0 ? zero : acc2[t]);
    }
    if constexpr (T1 & 1) {
        u32x2 hb[2];
        {
            float x0 = act(T1 - 1, 0), x1 = act(T1 - 1, 1), x2 = act(T1 - 1, 2), x3 = act(T1 - 1, 3);
            const unsigned a0 = h16_split_stage(x0, x1), a1 = h16_split_stage(x2, x3);
            hb[0] = u32x2{a0, a1};
            hb[1] = u32x2{h16_split_last(x0, x1), h16_split_last(x2, x3)};
        }
        const unsigned short* himg = base + BT * KSF * 2 * FRAG + lane * 4;
#pragma unroll
        for (int t = 0; t < BT; ++t) {
            const u32x2 wh = *reinterpret_cast<const u32x2*>(himg + (t * 2 + 0) * 256), wl = *reinterpret_cast<const u32x2*>(himg + (t * 2 + 1) * 256);
            out[t] = mfma_f16_k16(wh, hb[0], KSF == 0 ? zero : out[t]);
            out[t] = mfma_f16_k16(wh, hb[1], out[t]);
            acc2[t] = mfma_f16_k16(wl, hb[0], KSF == 0 ? zero : acc2[t]);
        }
    }
#pragma unroll
    for (int t = 0; t < BT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[t][r] = fmaf(acc2[t][r], W16_LOUNSCALE, out[t][r]);
}

template <int T1, int NL2>
__global__ __launch_bounds__(UMNN_BLOCK, 2) void cc_front_fwd16_kernel(const FrontArgs fa) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const BwdArgs& a = fa.b;
    const MlpDev& m = a.m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, p = lane & 15;
    const int H1 = m.width[1];
    const int E = a.E, d = a.d, n = a.n, nl2 = NL2 > 0 ? NL2 : fa.nl2;
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    unsigned short* lds16 = reinterpret_cast<unsigned short*>(lds);
    stage_g1_image16<T1>(m, lds16, tid, blockDim.x);
    __syncthreads();
    float w1x[T1][4];
    {
        const float* __restrict__ W0 = m.W[0];
#pragma unroll
        for (int t = 0; t < T1; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = feat_of(t, r, g);
                w1x[t][r] = f < H1 ? W0[f * (1 + E)] : 0.f;
            }
    }
    const unsigned wave_global = blockIdx.x * (blockDim.x >> 6) + wid;
    const unsigned nwaves = gridDim.x * (blockDim.x >> 6);
    for (unsigned item = wave_global; item < a.ngroups; item += nwaves) {
        const unsigned grp = fa.grp0 + item;
        const long long q = (long long)grp * 16 + p;
        const long long qq = q < a.NI ? q : a.NI - 1;
        const float xv = io_ld(a.x, qq, a.x_bf16);
        const float x0v = a.x0 ? io_ld(a.x0, qq, a.x_bf16) : 0.f;
        const float dxv = xv - x0v;
        const long long bi = qq / d;
        const IoView hb = IoView{a.h, a.h_bf16} + (bi * ((long long)E * d) + (qq - bi * d));
        f32x4 c[T1];
        front_prologue<T1>(m, hb, E, d, g, p, c);
        const size_t frag0 = (size_t)item * (size_t)(n + 1) * nl2 * 64 + lane;
        for (int k = 0; k <= (fa.tz_only ? 0 : n); ++k) {
            const float u = a.ccs[k] + 1.f;
            const float tk = k == 0 ? xv : __fadd_rn(x0v, __fmul_rn(dxv, u) * 0.5f);
            f32x4 z2[BT];
            front_gemm16<T1>(lds16, lane, [&](int t, int r) { return hidden_act_f(fmaf(w1x[t][r], tk, c[t][r]), slope); }, z2);
#pragma unroll
            for (int t = 0; t < BT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * t + r < nl2 && !fa.tz_only) fa.z2[frag0 + ((size_t)k * nl2 + 4 * t + r) * 64] = z2[t][r];
            if (k == 0 && fa.tz2) {
                f32x4 tz[BT];
                front_gemm16<T1>(lds16, lane, [&](int t, int r) { return w1x[t][r] * (fmaf(w1x[t][r], tk, c[t][r]) > 0.f ? 1.f : slope); }, tz);
#pragma unroll
                for (int t = 0; t < BT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * t + r < nl2) fa.tz2[((size_t)item * nl2 + 4 * t + r) * 64 + lane] = tz[t][r];
            }
        }
    }
}
// fragment image of G1^T on two fp16 pieces (stage C on fp16 pieces): rows = hidden-1 features (T1 tiles), K = hidden-2 features (two
// K-steps), nothing through the constant feature; the low piece plain (ws16_stage_image<TRANSPOSED = true>: the delta chain's convention)
template <int T1>
__device__ __forceinline__ void stage_g1t_image16(const MlpDev& m, unsigned short* img, int tid, int nthreads) {
    const int Hin = m.width[1], Hout = m.width[2];
    const float* __restrict__ W = m.W[1];
    for (int idx = tid; idx < T1 * BKS * FRAG; idx += nthreads) {
        const int j = idx & 7, ln = (idx >> 3) & 63, ts = idx >> 9;
        const int s = ts % BKS, t = ts / BKS;
        const int frow = fout_of(t, ln & 15);
        const int fk = feat_of(2 * s + (j >> 2), j & 3, ln >> 4);
        const float v = (fk < Hout && frow < Hin) ? W[fk * Hin + frow] : 0.f;
        const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
        img[(ts * 2 + 0) * FRAG + ln * 8 + j] = __builtin_bit_cast(unsigned short, hi);
        img[(ts * 2 + 1) * FRAG + ln * 8 + j] = __builtin_bit_cast(unsigned short, lo);
    }
}
#endif

// ---------------------------------------------------------------------------------------------- stage C
// The body works on the hidden-1 feature tiles [T0, T0 + TN) of a tile of 16 integrals: its columns of dG_1, its rows of
// delta_1 = (G_1^T delta_2) . act'(z_1), its entries of dc and dW_1[:, 0].  TN = T1: one wave does the whole tile (the kernel of
// rounds 2-5: 480 registers, one wave per SIMD, 39 % of the matrix pipe).  Round 6: the tiles of a_1 are INDEPENDENT of each other in
// everything this stage computes -- only delta_2 is common -- so two waves share a tile of integrals, tiles [0, 4) and [4, T1) of the
// hidden-1 features (two whole K-steps of 32 features first, so the halves stay K-step aligned), each at <= 256 registers, two per
// SIMD: the second wave of a SIMD fills the first one's split / activation phases and dependency stalls.  Nothing is exchanged between
// the halves; both fetch and split delta_2 (13 registers), and they write DISJOINT entries of the same d_theta slice and of dc.
// The two-wave form (TN < T1, two-piece build) also takes dG_1 the way the workgroup pipelines take their dW products: the packed
// fragments of delta_2 and of this wave's a_1 go into a wave-private pair of LDS tiles ([piece][point][slot], 9 KB), come back as
// transposed operands (ds_read_b64_tr_b16) and meet on v_mfma_f32_32x32x16 (K = the 16 points) -- 24 matrix instructions of 32 cycles
// per tile-node for both halves together.  The one-wave form transposes on the matrix core (22 MFMAs + 44 conversions) and accumulates
// on the K = 16 16x16 instruction, which gfx950 runs at HALF rate (16 cycles for 8 k FLOP: counters, round 6): 84 of them were 53 % of
// this stage's matrix time.
// F16 (two-wave form only, behind the fp16 middle stage): every matrix operand on two fp16 pieces instead of two bf16 pieces -- a
// four-instruction split per pair instead of eleven (with the transposes gone, 308 of this stage's ~600 vector instructions per tile-node
// are splits).  delta_2 arrives un-scaled from the middle stage and is multiplied by the launch's power-of-two sigma (cc_bwd_ws16_kernel.h:
// the cotangent scale of the fp16 pipeline) before it is split; dG_1, dc and dW_1[:, 0] leave times 1 / sigma (exact).  A piece beyond
// fp16's range reaches a dc sum or a dG_1 accumulator as inf / NaN: checked (once per tile / once per launch) and raised in the launch
// flag, behind which the bf16 build of this stage is queued with "run only if the flag is set" -- same outputs, rewritten.
template <int T1, int NL2, int T0, int TN, bool F16 = false>
__device__ __forceinline__ void front_bwd_body(const FrontArgs& fa, const unsigned short* lds16, unsigned slot_global, unsigned nslots,
                                               unsigned short* wave_tiles = nullptr) {
#if UMNN_BWD_NPB == 2
    constexpr bool LDSDW = TN < T1;
#else
    constexpr bool LDSDW = false;
#endif
    static_assert(!F16 || LDSDW, "fp16 pieces: the two-wave form of the two-piece build");
    constexpr int KSN = (TN + 1) / 2;          // K-steps of 32 hidden-1 features of THIS wave's tiles (the last one half empty when TN is odd)
    static_assert((T0 & 1) == 0 && T0 + TN <= T1, "feature-tile range: K-step aligned, inside the layer");
    const BwdArgs& a = fa.b;
    const MlpDev& m = a.m;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, p = lane & 15;
    const int H1 = m.width[1], H2 = m.width[2];
    const int E = a.E, d = a.d, n = a.n, nl2 = NL2 > 0 ? NL2 : fa.nl2;
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    const unsigned short* frag_base = lds16 + lane * 8;

    u32x4 sel[2];        // selection fragments of the matrix-core transpose (see cc_bwd_bf16_kernel)
    {
        const unsigned one_lo = 0x3f80u, one_hi = 0x3f800000u;
        const int slot = p >> 2;
        const unsigned w0 = (p & 3) == g ? (slot == 0 ? one_lo : slot == 1 ? one_hi : 0u) : 0u;
        const unsigned w1 = (p & 3) == g ? (slot == 2 ? one_lo : slot == 3 ? one_hi : 0u) : 0u;
        sel[0] = u32x4{w0, w1, 0u, 0u};
        sel[1] = u32x4{0u, 0u, w0, w1};
    }
    float sigma = 1.f, inv_sigma = 1.f;
    bool bad = false;
#if UMNN_BWD_NPB == 2
    if constexpr (F16) sigma = ws16_sigma(reinterpret_cast<const Ws16Scal*>(a.scal), inv_sigma);
#endif
    auto mm32 = [&](u32x4 x, u32x4 y, f32x4 z) __attribute__((always_inline)) {      // the 16x16x32 instruction of the piece type
#if UMNN_BWD_NPB == 2
        if constexpr (F16) return mfma_f16(x, y, z);
#endif
        return mfma_bf16(x, y, z);
    };
    auto split2 = [&](float x0, float x1, unsigned (&q)[NPB]) __attribute__((always_inline)) {
#if UMNN_BWD_NPB == 2
        if constexpr (F16) { q[0] = h16_split_stage(x0, x1); q[1] = h16_split_last(x0, x1); return; }
#endif
        split_pair<NPB>(x0, x1, q);
    };
    float w1x[TN][4];
    {
        const float* __restrict__ W0 = m.W[0];
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = feat_of(T0 + t, r, g);
                w1x[t][r] = f < H1 ? W0[f * (1 + E)] : 0.f;
            }
    }
    f32x4 dG1[LDSDW ? 1 : BT][LDSDW ? 1 : TN], dW1x[TN];
#pragma unroll
    for (int to = 0; to < (LDSDW ? 1 : BT); ++to)
#pragma unroll
        for (int ti = 0; ti < (LDSDW ? 1 : TN); ++ti) dG1[to][ti] = f32x4{0.f, 0.f, 0.f, 0.f};
#if UMNN_BWD_NPB == 2
    ws_f32x16 dGw[2][2];                         // (LDSDW) rows: slots of the delta_2 tile, columns: slots of this wave's a_1 tile
#pragma unroll
    for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int v = 0; v < 16; ++v) dGw[to][ti][v] = 0.f;
    const int trb = (8 * (g >> 1) + (p >> 2)) * TRS + 16 * (g & 1) + 4 * (p & 3);
#endif
#pragma unroll
    for (int t = 0; t < TN; ++t) dW1x[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (unsigned item = slot_global; item < a.ngroups; item += nslots) {
        const unsigned grp = fa.grp0 + item;
        const long long q = (long long)grp * 16 + p;
        const bool ok = q < a.NI;
        const long long qq = ok ? q : a.NI - 1;
        const float xv = io_ld(a.x, qq, a.x_bf16);
        const float x0v = a.x0 ? io_ld(a.x0, qq, a.x_bf16) : 0.f;
        const float dxv = xv - x0v;
        const long long bi = qq / d;
        const IoView hb = IoView{a.h, a.h_bf16} + (bi * ((long long)E * d) + (qq - bi * d));
        f32x4 c[TN], dcs[TN];
        {
            // (the hoisted first-layer term of THIS wave's tiles: the products of the other tiles have no user and are not emitted)
            f32x4 call[T1];
            front_prologue<T1>(m, hb, E, d, g, p, call);
#pragma unroll
            for (int t = 0; t < TN; ++t) c[t] = call[T0 + t];
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) dcs[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const size_t frag0 = (size_t)item * (size_t)(n + 1) * nl2 * 64 + lane;
        // delta_2 of node k, register j: fetched UNCONDITIONALLY (register index clamped, value masked afterwards) -- a guarded
        // load per register turns into a branch and a full memory wait each, which is what this kernel used to spend half its
        // time in -- and one node ahead
        auto ld_d2 = [&](int k, int t, int r) {
            const int j = 4 * t + r, jj = j < nl2 ? j : nl2 - 1;
            const float v = fa.d2[frag0 + ((size_t)k * nl2 + jj) * 64];
            return j < nl2 ? v : 0.f;
        };
        f32x4 dnext[BT];
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) dnext[t][r] = ld_d2(0, t, r);

        for (int k = 0; k <= n; ++k) {
            const float u = a.ccs[k] + 1.f;
            const float tk = k == 0 ? xv : __fadd_rn(x0v, __fmul_rn(dxv, u) * 0.5f);
            f32x4 a1[TN], delta2[BT];
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) a1[t][r] = hidden_act_f(fmaf(w1x[t][r], tk, c[t][r]), slope);
#pragma unroll
            for (int t = 0; t < BT; ++t) delta2[t] = F16 ? dnext[t] * sigma : dnext[t];
            {
                const int kn = k < n ? k + 1 : n;         // (the last node re-reads itself: harmless, and no branch)
#pragma unroll
                for (int t = 0; t < BT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dnext[t][r] = ld_d2(kn, t, r);
            }
            // ---- packed two-piece fragments: delta_2 as a BFrag (two K-steps), a_1 as KSN K-steps
            BFrag<NPB> bd;
#pragma unroll
            for (int s2 = 0; s2 < BKS; ++s2) {
                unsigned q0[NPB], q1[NPB], q2[NPB], q3[NPB];
                split2(delta2[2 * s2][0], delta2[2 * s2][1], q0);
                split2(delta2[2 * s2][2], delta2[2 * s2][3], q1);
                split2(delta2[2 * s2 + 1][0], delta2[2 * s2 + 1][1], q2);
                split2(delta2[2 * s2 + 1][2], delta2[2 * s2 + 1][3], q3);
#pragma unroll
                for (int k2 = 0; k2 < NPB; ++k2) bd.v[s2][k2] = u32x4{q0[k2], q1[k2], q2[k2], q3[k2]};
            }
            u32x4 ba[KSN][NPB];
#pragma unroll
            for (int s = 0; s < KSN; ++s) {
                unsigned q0[NPB], q1[NPB], q2[NPB], q3[NPB];
#pragma unroll
                for (int k2 = 0; k2 < NPB; ++k2) q2[k2] = q3[k2] = 0u;
                split2(a1[2 * s][0], a1[2 * s][1], q0);
                split2(a1[2 * s][2], a1[2 * s][3], q1);
                if (2 * s + 1 < TN) {
                    split2(a1[2 * s + 1][0], a1[2 * s + 1][1], q2);
                    split2(a1[2 * s + 1][2], a1[2 * s + 1][3], q3);
                }
#pragma unroll
                for (int k2 = 0; k2 < NPB; ++k2) ba[s][k2] = u32x4{q0[k2], q1[k2], q2[k2], q3[k2]};
            }
            // ---- dG1 += delta_2 (x) a_1 over the 16 points
#if UMNN_BWD_NPB == 2
            if constexpr (LDSDW) {
                unsigned short* Dt = wave_tiles;
                unsigned short* At = wave_tiles + WS_TILE;
                BFrag<NPB> af;                                  // this wave's a_1 as two K-steps (TN = 3: the last half K-step empty)
#pragma unroll
                for (int s2 = 0; s2 < BKS; ++s2)
#pragma unroll
                    for (int k2 = 0; k2 < NPB; ++k2) af.v[s2][k2] = s2 < KSN ? ba[s2 < KSN ? s2 : 0][k2] : u32x4{0u, 0u, 0u, 0u};
                tr_tile_store(bd, Dt, g, p);
                tr_tile_store(af, At, g, p);
                WsOps ops;
                swp_static_for<8>([&](auto ic) { ws_load_op<decltype(ic)::value>(ops, Dt + trb, At + trb); });
                if constexpr (F16) swp_static_for<12>([&](auto ic) { ws16_dw_mfma<decltype(ic)::value>(dGw, ops); });
                else swp_static_for<12>([&](auto ic) { ws_dw_mfma<decltype(ic)::value>(dGw, ops); });
            } else
#endif
            {
                u32x2 dT[BT][NPB], aT[TN][NPB];
                transpose_pieces(bd, sel, dT);
#pragma unroll
                for (int s = 0; s < KSN; ++s)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        if (2 * s + hh >= TN) continue;
#pragma unroll
                        for (int part = 0; part < NPB; ++part) {
                            const f32x4 tr = mfma_bf16(ba[s][part], sel[hh], f32x4{0.f, 0.f, 0.f, 0.f});
                            const bf16x2 lo = __builtin_convertvector(f32x2{tr[0], tr[1]}, bf16x2);
                            const bf16x2 hi = __builtin_convertvector(f32x2{tr[2], tr[3]}, bf16x2);
                            aT[2 * s + hh][part] = u32x2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
                        }
                    }
#pragma unroll
                for (int wa = 0; wa < NPB; ++wa)
#pragma unroll
                    for (int bb = 0; bb < NPB; ++bb) {
                        if (wa + bb >= NPB) continue;
#pragma unroll
                        for (int to = 0; to < BT; ++to)
#pragma unroll
                            for (int ti = 0; ti < TN; ++ti) dG1[to][ti] = mfma_bf16_k16(dT[to][wa], aT[ti][bb], dG1[to][ti]);
                    }
            }
            // ---- delta_1 = (G1^T delta_2) . act'(z_1)
            f32x4 nd[TN];
            {
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < BKS; ++s) {
#pragma unroll
                    for (int t = 0; t < TN; ++t) {
                        u32x4 wf[NPB];
#pragma unroll
                        for (int k2 = 0; k2 < NPB; ++k2)
                            wf[k2] = *reinterpret_cast<const u32x4*>(frag_base + (((T0 + t) * BKS + s) * NPB + k2) * FRAG);
#pragma unroll
                        for (int wa = 0; wa < NPB; ++wa)
#pragma unroll
                            for (int bb = 0; bb < NPB; ++bb) {
                                if (wa + bb >= NPB) continue;
                                const bool first = s == 0 && wa == 0 && bb == 0;
                                nd[t] = mm32(wf[wa], bd.v[s][bb], first ? zero : nd[t]);
                            }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dl = nd[t][r] * (a1[t][r] > 0.f ? 1.f : slope);
                    dcs[t][r] += dl;
                    dW1x[t][r] = fmaf(dl, tk, dW1x[t][r]);
                }
        }
        if constexpr (F16) {
            float chk = 0.f;
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) chk = fmaf(dcs[t][r], 0.f, chk);         // (NaN iff some entry is inf / NaN)
            bad = bad || !(chk == 0.f);
        }
        if (ok) {
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = feat_of(T0 + t, r, g);
                    if (f < H1) a.dc[q * H1 + f] = dcs[t][r] * inv_sigma;
                }
        }
    }
#if UMNN_BWD_NPB == 2
    if constexpr (F16) {
        float chk = 0.f;
#pragma unroll
        for (int to = 0; to < 2; ++to)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int v = 0; v < 16; ++v) chk = fmaf(dGw[to][ti][v], 0.f, chk);
        bad = bad || !(chk == 0.f);
        if (__any(bad) && lane == 0) atomicOr(&reinterpret_cast<Ws16Scal*>(a.scal)->flag, 1u);
    }
#endif
    // ---- this slot's partial d_theta: the G1 columns (weights + bias column) and the entries of the x-column of W1 that belong to
    // these feature tiles
    float* part = a.partials + (size_t)slot_global * a.n_params;
#if UMNN_BWD_NPB == 2
    if constexpr (LDSDW) {
        // 32 x 32 result layout (ws_write_dw): column lane & 31, register v = 4 i + r <-> row 8 i + 4 (lane >> 5) + r; rows / columns run
        // over tile SLOTS (slot_feature), the columns being the features of THIS wave's tiles
#pragma unroll
        for (int to = 0; to < 2; ++to)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int fo = slot_feature(32 * to + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3));
                    const int fi = 16 * T0 + slot_feature(32 * ti + (lane & 31));
                    if (fo < H2) {
                        const int idx = fi < H1 ? a.poffW[1] + fo * H1 + fi : (fi == H1 ? a.poffb[1] + fo : -1);
                        if (idx >= 0) part[idx] = (fa.accumulate ? part[idx] : 0.f) + dGw[to][ti][v] * inv_sigma;
                    }
                }
    } else
#endif
#pragma unroll
    for (int to = 0; to < BT; ++to)
#pragma unroll
        for (int ti = 0; ti < TN; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int fo = 16 * to + 4 * g + r, fi = 16 * (T0 + ti) + (lane & 15);
                if (fo < H2) {
                    const int idx = fi < H1 ? a.poffW[1] + fo * H1 + fi : (fi == H1 ? a.poffb[1] + fo : -1);
                    if (idx >= 0) part[idx] = (fa.accumulate ? part[idx] : 0.f) + dG1[to][ti][r];
                }
            }
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v1 = dW1x[t][r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) v1 += __shfl_xor(v1, o);
            const int f = feat_of(T0 + t, r, g);
            if (p == 0 && f < H1) {
                const int idx = a.poffW[0] + f * (1 + E);
                part[idx] = (fa.accumulate ? part[idx] : 0.f) + v1 * inv_sigma;
            }
        }
}

// one wave per tile of integrals (what bwd_precision = fp32 runs: the six-term build does not fit 256 registers in halves either)
template <int T1, int NL2>
__global__ __launch_bounds__(UMNN_BLOCK, 1) void cc_front_bwd_kernel(const FrontArgs fa) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned short* lds16 = reinterpret_cast<unsigned short*>(lds);
    stage_g1_image<T1, true, NPB>(fa.b.m, lds16, threadIdx.x, blockDim.x);
    __syncthreads();
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    front_bwd_body<T1, NL2, 0, T1>(fa, lds16, blockIdx.x * (blockDim.x >> 6) + wid, gridDim.x * (blockDim.x >> 6));
}

// two waves per tile of integrals (eight per workgroup; waves w and w + 4 share a SIMD and a tile): feature tiles [0, 4) and [4, T1)
template <int T1, int NL2>
__global__ __launch_bounds__(2 * UMNN_BLOCK, 1) void cc_front_bwd2_kernel(const FrontArgs fa) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned short* lds16 = reinterpret_cast<unsigned short*>(lds);
    if (fa.only_if && *fa.only_if == 0) return;      // queued as the fallback of the fp16 build: nothing overflowed
    stage_g1_image<T1, true, NPB>(fa.b.m, lds16, threadIdx.x, blockDim.x);
    __syncthreads();
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned slot = blockIdx.x * UMNN_WAVES_PER_BLOCK + (wid & (UMNN_WAVES_PER_BLOCK - 1)), nslots = gridDim.x * UMNN_WAVES_PER_BLOCK;
    // (behind the G1^T image: a pair of operand tiles per wave)
    unsigned short* wave_tiles = lds16 + T1 * BKS * NPB * FRAG + wid * 2 * (NPB * 16 * TRS);
    if (wid < UMNN_WAVES_PER_BLOCK) front_bwd_body<T1, NL2, 0, 4>(fa, lds16, slot, nslots, wave_tiles);
    else front_bwd_body<T1, NL2, 4, T1 - 4>(fa, lds16, slot, nslots, wave_tiles);
}

#if UMNN_BWD_NPB == 2
// the same on fp16 pieces (behind the fp16 middle stage only: the launch's sigma and flag are its)
template <int T1, int NL2>
__global__ __launch_bounds__(2 * UMNN_BLOCK, 1) void cc_front_bwd16_kernel(const FrontArgs fa) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned short* lds16 = reinterpret_cast<unsigned short*>(lds);
    stage_g1t_image16<T1>(fa.b.m, lds16, threadIdx.x, blockDim.x);
    __syncthreads();
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned slot = blockIdx.x * UMNN_WAVES_PER_BLOCK + (wid & (UMNN_WAVES_PER_BLOCK - 1)), nslots = gridDim.x * UMNN_WAVES_PER_BLOCK;
    unsigned short* wave_tiles = lds16 + T1 * BKS * NPB * FRAG + wid * 2 * (NPB * 16 * TRS);
    if (wid < UMNN_WAVES_PER_BLOCK) front_bwd_body<T1, NL2, 0, 4, true>(fa, lds16, slot, nslots, wave_tiles);
    else front_bwd_body<T1, NL2, 4, T1 - 4, true>(fa, lds16, slot, nslots, wave_tiles);
}
#endif

}  // namespace UMNN_BWD_NS

// ------------------------------------------------------------------------------------------ host side
typedef void (*front_kernel_t)(const FrontArgs);
typedef void (*mid_kernel_t)(const BwdBf16Args);
#if UMNN_BWD_NPB == 2
struct FrontVariant { int t1, nl2; front_kernel_t fwd, bwd, fwd16, bwd2, bwd16; };
#define FRONT_VARIANT(T, N) {T, N, cc_front_fwd_kernel<T, N>, cc_front_bwd_kernel<T, N>, cc_front_fwd16_kernel<T, N>, cc_front_bwd2_kernel<T, N>, cc_front_bwd16_kernel<T, N>}
#else
struct FrontVariant { int t1, nl2; front_kernel_t fwd, bwd, fwd16, bwd2, bwd16; };
#define FRONT_VARIANT(T, N) {T, N, cc_front_fwd_kernel<T, N>, cc_front_bwd_kernel<T, N>, nullptr, nullptr, nullptr}
#endif
static const FrontVariant kFrontVariants[] = {
    FRONT_VARIANT(5, 13), FRONT_VARIANT(6, 13), FRONT_VARIANT(7, 13), FRONT_VARIANT(8, 13),
    FRONT_VARIANT(5, 0), FRONT_VARIANT(6, 0), FRONT_VARIANT(7, 0), FRONT_VARIANT(8, 0),
};
struct MidVariant { int lh, nrl; mid_kernel_t fn; const char* name; };
#if UMNN_BWD_NPB == 2
#define MID_VARIANT(LHH, NR) { LHH, NR, cc_bwd_bf16_kernel<LHH, true, NR, true>, "cc_bwd_bf16<L=" #LHH ",EDGE=1,LIVE=" #NR ",FRONT>" }
#else
#define MID_VARIANT(LHH, NR) { LHH, NR, cc_bwd_bf16_kernel<LHH, true, NR, true>, "cc_bwd_bf16x6<L=" #LHH ",EDGE=1,LIVE=" #NR ",FRONT>" }
#endif
static const MidVariant kMidVariants[] = { MID_VARIANT(4, 13), MID_VARIANT(3, 13), MID_VARIANT(2, 13),
                                           MID_VARIANT(4, 0), MID_VARIANT(3, 0), MID_VARIANT(2, 0) };
// the middle stage as a weight-stationary workgroup pipeline (cc_bwd_ws_kernel.h): four hidden layers after the cut, chunks
// of at least four tiles per workgroup
#if UMNN_BWD_NPB == 2
static const MidVariant kMidWsVariants[] = {
    { 4, 13, cc_bwd_ws_kernel<13, true>, "cc_bwd_bf16<L=4,LIVE=13,WS,FRONT>" },
    { 4, 0, cc_bwd_ws_kernel<0, true>, "cc_bwd_bf16<L=4,LIVE=0,WS,FRONT>" },
};
#endif
// the same on fp16 pieces (cc_bwd_ws16_kernel.h, instantiated in cc_backward_bf16.hip under that file's scheduling flags):
// single-chunk calls only -- its d_theta slices are written, not accumulated, so that the bf16 pipeline queued behind it as the
// overflow fallback can rewrite them
int umnn_ws16_front_eligible(const BwdArgs& base, int nrl, const char** name);
int umnn_ws16_front_prepare(const BwdArgs& base, int nblocks_max, hipStream_t stream);
int umnn_ws16_front_launch(const BwdBf16Args& mid, int nrl, int nblocks, hipStream_t stream);

// Does this net belong to the family?  hidden layer 1: 5..8 tiles; hidden layers 2..L: three or four tiles at most (zero-padded
// to four), 2..4 of them.
#if UMNN_BWD_NPB == 2
int umnn_backward_front_shape(const MlpDev& m) {
    const int L = m.n_linear - 1;
    if (L < 3 || L > 5) return 0;
    if (m.t_out[1] < 5 || m.t_out[1] > 8) return 0;
    for (int l = 2; l <= L; ++l) if (m.t_out[l] > BT) return 0;
    return 1;
}

// bytes of HBM scratch the staged backward wants for B*d integrals (nb_steps is not known when the workspace is sized: the
// chunking adapts to whatever n the call brings)
long long umnn_backward_front_scratch_bytes(const MlpDev& m, long long NI) {
    const long long tiles = (NI + 15) / 16;
    const long long nl2 = (m.width[2] + 1 + 3) / 4;
    const long long per_tile = (2LL * 257 + 1) * nl2 * 64 * 4;          // sized for n = 256
    const long long want = tiles * per_tile, cap = 2LL << 30;
    return want < cap ? want : cap;
}
#define UMNN_FRONT_LAUNCH umnn_launch_backward_front
#else
int umnn_backward_front_shape(const MlpDev& m);
#define UMNN_FRONT_LAUNCH umnn_launch_backward_front_p3
#endif

// Runs the three stages chunk by chunk.  `base` is the fully populated BwdArgs of umnn_cc_backward (partials zeroed).
int UMNN_FRONT_LAUNCH(const BwdArgs& base, const umnn_mlp* net, int nblocks_max, void* scratch, long long scratch_bytes,
                               hipStream_t stream) {
    const MlpDev& m = base.m;
    const int L = m.n_linear - 1, n = base.n;
    if (!umnn_backward_front_shape(m)) return UMNN_EUNSUPPORTED;
    const int T1 = m.t_out[1], LH = L - 1;
    const int nl2 = (m.width[2] + 1 + 3) / 4;
    const FrontVariant* fv = nullptr;
    for (const FrontVariant& v : kFrontVariants) if (v.t1 == T1 && v.nl2 == (nl2 == 13 ? 13 : 0)) fv = &v;
    int nrl = m.ks_in[2];
    for (int l = 2; l <= L; ++l) if (m.ks_in[l] != nrl) nrl = 0;
    if (nrl != 13) nrl = 0;
    const MidVariant* mv = nullptr;
    for (const MidVariant& v : kMidVariants) if (v.lh == LH && v.nrl == nrl) { mv = &v; break; }
    if (!fv || !mv) return UMNN_EUNSUPPORTED;

    const long long tiles = (base.NI + 15) / 16;
    // (z_2 left by the training forward: the scratch holds delta_2 and the tangent only, stage A runs for the tangent element alone)
    const bool saved = base.z2_saved != nullptr;
    const long long per_tile = ((saved ? 1LL : 2LL) * (n + 1) + (base.gfx ? 1 : 0)) * nl2 * 64 * 4;
    long long chunk = scratch_bytes / per_tile;
    if (chunk < 1 || !scratch) return UMNN_EUNSUPPORTED;
    if (chunk > tiles) chunk = tiles;
    else {
        // several chunks: equal shares, rounded up to whole rounds of the persistent grid where the scratch allows (a chunk
        // of 3.06 rounds costs 4)
        const long long nchunks = (tiles + chunk - 1) / chunk, waves = (long long)nblocks_max * UMNN_WAVES_PER_BLOCK;
        long long even = (tiles + nchunks - 1) / nchunks;
        const long long rounded = (even + waves - 1) / waves * waves;
        chunk = rounded <= chunk ? rounded : even;
    }

    // ---- the middle stage sees the net from hidden layer 2 on
    BwdBf16Args mid;
    mid.b = base;
    mid.scal = nullptr; mid.only_if = nullptr;
    {
        MlpDev& s = mid.b.m;
        s.n_linear = m.n_linear - 1;
        for (int l = 0; l <= s.n_linear; ++l) s.width[l] = m.width[l + 1];
        for (int l = 0; l < s.n_linear; ++l) { s.W[l] = m.W[l + 1]; s.b[l] = m.b[l + 1]; }
        for (int l = 1; l <= s.n_linear - 1; ++l) { s.t_out[l] = m.t_out[l + 1]; s.ks_in[l] = m.ks_in[l + 1]; s.t_mfma[l] = m.t_mfma[l + 1]; }
        for (int l = 0; l < s.n_linear; ++l) { mid.b.poffW[l] = base.poffW[l + 1]; mid.b.poffb[l] = base.poffb[l + 1]; }
    }
    int off16 = 0;
    for (int l = 1; l < LH; ++l) { mid.off_fwd[l] = off16; off16 += BT * BKS * NPF * FRAG; }
    for (int l = 1; l < LH; ++l) { mid.off_tr[l] = off16; off16 += BT * BKS * NPB * FRAG; }
    mid.off_trtile = off16;
    // (waves per workgroup of the one-pass middle kernel: four, or as many as leave room for their transpose tiles next to the weight
    // images -- the six-term build with four hidden layers after the cut holds two; the grid grows by the same factor, so the
    // d_theta slices stay one per wave of the plan)
    int wpb_mid = UMNN_WAVES_PER_BLOCK;
    while (wpb_mid > 1 && (size_t)(off16 + wpb_mid * NPB * 16 * TRS) * sizeof(unsigned short) > 160 * 1024) wpb_mid >>= 1;
    off16 += wpb_mid * NPB * 16 * TRS;
    const size_t lds_mid = (size_t)off16 * sizeof(unsigned short);
    const size_t lds_a = ((size_t)BT * (T1 / 2) * NPF * FRAG + (T1 & 1 ? BT * NPF * 256 : 0)) * sizeof(unsigned short);
    const size_t lds_c = (size_t)T1 * BKS * NPB * FRAG * sizeof(unsigned short);
    if (lds_mid > 160 * 1024 || lds_a > 160 * 1024 || lds_c > 160 * 1024) return UMNN_EUNSUPPORTED;
    if (int rc = umnn_allow_lds((const void*)fv->fwd, lds_a)) return rc;
    if (fv->fwd16) { if (int rc = umnn_allow_lds((const void*)fv->fwd16, lds_a)) return rc; }
    if (int rc = umnn_allow_lds((const void*)mv->fn, lds_mid)) return rc;
    if (int rc = umnn_allow_lds((const void*)fv->bwd, lds_c)) return rc;
    // stage C with two waves per tile (two-piece build): UMNN_FRONT_BWD2=0 / option front_bwd2 = 0 keeps one wave per tile
    const bool c2 = fv->bwd2 != nullptr && umnn_options().front_bwd2 != 0;
    const size_t lds_c2 = lds_c + (size_t)2 * UMNN_WAVES_PER_BLOCK * 2 * (NPB * 16 * TRS) * sizeof(unsigned short);      // + the waves' operand tiles
    if (c2) { if (lds_c2 > 160 * 1024) return UMNN_EUNSUPPORTED; if (int rc = umnn_allow_lds((const void*)fv->bwd2, lds_c2)) return rc; }
    // ... and on fp16 pieces behind the fp16 middle stage (front_bwd2 = 2 keeps bf16 pieces there)
    const bool c16 = c2 && fv->bwd16 != nullptr && umnn_options().front_bwd2 == 1;
    if (c16) { if (int rc = umnn_allow_lds((const void*)fv->bwd16, lds_c2)) return rc; }
    const MidVariant* wv = nullptr;
    size_t lds_ws = 0;
#if UMNN_BWD_NPB == 2
    if (umnn_options().bwd_ws && LH == 4)
        for (const MidVariant& v : kMidWsVariants) if (v.nrl == nrl) { wv = &v; break; }
    lds_ws = (size_t)WS_LDS_USHORTS * sizeof(unsigned short);
    if (wv) { if (int rc = umnn_allow_lds((const void*)wv->fn, lds_ws)) return rc; }
#endif
    bool used_ws = false;
    // fp16 pieces for the middle stage: the rule of umnn_launch_backward_bf16 (bwd_ws16 = 1: large launches only; here from 2^21 node
    // evaluations, the middle stage recomputes one layer less than the whole-net pipeline; 2: whenever eligible), one chunk
    const char* hv = nullptr;
    {
        const int w16 = umnn_options().bwd_ws16;
        const bool size_ok = w16 == 2 || (w16 == 1 && base.NI * (long long)(n + 1) >= (1LL << 21));
        if (wv && size_ok && chunk == tiles && tiles >= 4LL * nblocks_max)
            if (int rc = umnn_ws16_front_eligible(base, nrl, &hv)) return rc;
    }

    FrontArgs fa;
    fa.b = base;
    fa.only_if = nullptr;
    fa.tz_only = saved ? 1 : 0;
    fa.b.ns = 1;
    mid.b.ns = 1;
    mid.b.l_lo = 1;
    fa.nl2 = mid.nl2 = nl2;
    // (the scalar memset + cotangent-scale pre-pass first: an error return must not leave a profiling bracket open)
    if (hv) { if (int rc = umnn_ws16_front_prepare(base, nblocks_max, stream)) return rc; }
    umnn_prof_begin(stream);
    for (long long t0 = 0; t0 < tiles; t0 += chunk) {
        const long long nt = tiles - t0 < chunk ? tiles - t0 : chunk;
        float* z2 = saved ? const_cast<float*>(base.z2_saved) + (size_t)t0 * (n + 1) * nl2 * 64 : (float*)scratch;
        float* d2 = saved ? (float*)scratch : z2 + (size_t)nt * (n + 1) * nl2 * 64;
        const bool run_a = !saved || base.gfx != nullptr;      // stage A: everything, or (z_2 saved) the tangent element of the g_fx term
        float* tz2 = base.gfx ? d2 + (size_t)nt * (n + 1) * nl2 * 64 : nullptr;
        int nblocks = nblocks_max;
        if ((long long)nblocks * UMNN_WAVES_PER_BLOCK > nt) nblocks = (int)((nt + UMNN_WAVES_PER_BLOCK - 1) / UMNN_WAVES_PER_BLOCK);
        if (nblocks < 1) nblocks = 1;
        fa.z2 = z2; fa.d2 = d2; fa.tz2 = tz2; fa.grp0 = (unsigned)t0; fa.b.ngroups = (unsigned)nt; fa.accumulate = t0 > 0;
        mid.z2 = z2; mid.d2 = d2; mid.tz2 = tz2; mid.grp0 = (unsigned)t0; mid.b.ngroups = (unsigned)nt; mid.accumulate = t0 > 0;
#if UMNN_BWD_NPB == 2
        if (hv) {
            // (stages A and B on fp16 pieces, then their bf16 builds behind them that only run if a piece overflowed -- the launch
            // flag, raised by the checks of stage B, which also see a non-finite z_2 from stage A: same outputs, rewritten)
            if (run_a) hipLaunchKernelGGL(fv->fwd16, dim3(2 * nblocks), dim3(UMNN_BLOCK), lds_a, stream, fa);      // (two workgroups per CU)
            mid.scal = base.scal; mid.only_if = nullptr;
            if (int rc = umnn_ws16_front_launch(mid, nrl, nblocks_max, stream)) { umnn_prof_end(stream, 0.0, UMNN_PROF_BACKWARD); return rc; }
            fa.only_if = mid.only_if = base.scal + 3;            // (Ws16Scal::flag)
            if (run_a) hipLaunchKernelGGL(fv->fwd, dim3(nblocks), dim3(UMNN_BLOCK), lds_a, stream, fa);
            hipLaunchKernelGGL(wv->fn, dim3(nblocks_max), dim3(64 * WS_WAVES), lds_ws, stream, mid);
            fa.only_if = nullptr;
            used_ws = true;
        } else if (wv && nt >= 4LL * nblocks_max) {
            if (run_a) hipLaunchKernelGGL(fv->fwd, dim3(nblocks), dim3(UMNN_BLOCK), lds_a, stream, fa);
            hipLaunchKernelGGL(wv->fn, dim3(nblocks_max), dim3(64 * WS_WAVES), lds_ws, stream, mid);
            used_ws = true;
        } else
#endif
        {
            if (run_a) hipLaunchKernelGGL(fv->fwd, dim3(nblocks), dim3(UMNN_BLOCK), lds_a, stream, fa);
            hipLaunchKernelGGL(mv->fn, dim3(nblocks * (UMNN_WAVES_PER_BLOCK / wpb_mid)), dim3(64 * wpb_mid), lds_mid, stream, mid);
        }
        if (c16 && hv && used_ws) {
            // (fp16 pieces, then the bf16 build that only runs if the launch flag is up -- raised by stage B's checks or by this stage's own)
            hipLaunchKernelGGL(fv->bwd16, dim3(nblocks), dim3(2 * UMNN_BLOCK), lds_c2, stream, fa);
            fa.only_if = base.scal + 3;            // (Ws16Scal::flag)
            hipLaunchKernelGGL(fv->bwd2, dim3(nblocks), dim3(2 * UMNN_BLOCK), lds_c2, stream, fa);
            fa.only_if = nullptr;
        } else if (c2) hipLaunchKernelGGL(fv->bwd2, dim3(nblocks), dim3(2 * UMNN_BLOCK), lds_c2, stream, fa);
        else hipLaunchKernelGGL(fv->bwd, dim3(nblocks), dim3(UMNN_BLOCK), lds_c, stream, fa);
    }
    umnn_prof_end(stream, 3.0 * umnn_cc_forward_flops_per_integral(net, n) * (double)base.NI, UMNN_PROF_BACKWARD);
    umnn_note_launch(hv ? hv : used_ws ? wv->name : mv->name);
    return umnn_check(hipGetLastError(), "cc_bwd front launch");
}
