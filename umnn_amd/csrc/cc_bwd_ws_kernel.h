// Weight-stationary one-pass backward on the bf16 matrix cores (round 3, second kernel): the four waves of a workgroup stop
// being four copies of the same program.  Same arithmetic as cc_bwd_swp_kernel.h / cc_bwd_bf16_kernel.h (reference lines
// ParallelNeuralIntegral.py:66-94,110-123): six-term recompute, three-term delta chain, dW on the matrix core -- but the
// measured wall of those kernels was the OPERAND STREAM: with one tile of 16 integrals per wave every GEMM MFMA needs a fresh
// 1 KB weight fragment out of LDS (168 fetches per tile-node, 34 % of the launch: DESIGN 4.2).  Here the weights never move:
//
//   wave G1, G2, G3   own ONE hidden->hidden layer each: W_l (3 bf16 pieces, 96 registers) and W_l^T (2 pieces, 64
//                     registers) live in the wave's registers for the whole launch.  Per step the wave runs the forward GEMM
//                     of one tile-node and the W^T GEMM of another (72 MFMAs, A operands = registers, B operands = 10 LDS
//                     reads) with the activation / split work of both behind them.  G3 also owns the output layer
//                     (f, dout, delta_L, d w_out), G1 the tail (dc, dW1[:,0]).
//   wave C            owns every dW accumulator (48 tiles, 192 registers) and layer 1: per step a_1 of a new tile-node, and
//                     dW_l += delta_{l+1} (x) a_l for three tile-nodes in flight (K = 32 form: [d_hi|d_hi] x [a_hi|a_lo] and
//                     [d_lo|d_lo] x [a_hi|a_lo], 32 MFMAs per layer, operands by transposing LDS reads).
//
// Tile-nodes flow through the workgroup as a systolic pipeline over LDS tiles (the [piece][point][slot] tile of the other
// kernels: own-lane b128 reads give B operands back, ds_read_b64_tr_b16 gives the transposed dW operands), one s_barrier per
// step.  Element u of the stream (a node of a tile; node order 0, [tangent of node 0], 1, .., n per tile, tiles one after the
// other) is at:
//   step u      C   : a_1[u]                      -> tile A1[u % 10]
//   step u+1/2  G1  : GEMM / activation, a_2[u]   -> tile A2[u % 7]          (GEMM in one step, its vector work in the next:
//   step u+3/4  G2  : a_3[u]                      -> tile A3[u % 4]           the wave always has TWO independent jobs)
//   step u+5/6  G3  : a_4[u], f, delta_4[u]       -> tile D4[u % 2]
//   step u+7    G3  : delta_3[u] = W_3^T delta_4 . act'(a_3) -> D3[u % 2];   C: dW_3 += delta_4[u] (x) a_3[u]
//   step u+8    G2  : delta_2[u] -> D2[u % 2];                                C: dW_2 += delta_3[u] (x) a_2[u]
//   step u+9    G1  : delta_1[u] -> dc, dW1[:,0];                             C: dW_1 += delta_2[u] (x) a_1[u]
// LDS: 21 activation tiles + 6 cotangent tiles of 4.5 KB, 6 third-piece tiles of 2.25 KB = 135 KB; no weight images.
// The tangent pass of the g_fx term (d f / d x at node 0) is one extra stream element per tile: C sends w1 . act'(z_1), the G
// waves multiply by act'(a_{l+1}(node 0)) -- the signs of the element before -- instead of applying the activation (a second
// instantiation of the step body, taken once per tile), G3 turns it into dfdt and a zero cotangent.
#pragma once
#include "cc_bwd_swp_kernel.h"

constexpr int WS_TILE = NPB * 16 * TRS;          // ushorts: the two leading bf16 pieces of a 16-point x 64-feature tile
constexpr int WS_P3 = 16 * TRS;                  // the third piece (lives one step)
constexpr int WS_NS1 = 10, WS_NS2 = 7, WS_NS3 = 4;
constexpr int WS_OFF_A1 = 0;
constexpr int WS_OFF_A2 = WS_OFF_A1 + WS_NS1 * WS_TILE;
constexpr int WS_OFF_A3 = WS_OFF_A2 + WS_NS2 * WS_TILE;
constexpr int WS_OFF_D = WS_OFF_A3 + WS_NS3 * WS_TILE;      // delta_l, l = 2..4: tile (l - 2) * 2 + (u & 1)
constexpr int WS_OFF_P3 = WS_OFF_D + 6 * WS_TILE;           // third piece of a_l, l = 1..3: tile (l - 1) * 2 + (u & 1)
constexpr int WS_LDS_USHORTS = WS_OFF_P3 + 6 * WS_P3;
constexpr int WS_DEPTH = 10;                     // steps between an element entering (C) and leaving (G1's tail)
constexpr int WS_BIAS = 140;                     // multiple of every slot count: keeps (step - delay) non-negative under %
constexpr int WS_IMGF = BT * BKS * NPF * FRAG, WS_IMGT = BT * BKS * NPB * FRAG;   // staging images (start of the launch only)

__host__ __device__ constexpr int ws_a_off(int l) { return l == 1 ? WS_OFF_A1 : (l == 2 ? WS_OFF_A2 : WS_OFF_A3); }
__host__ __device__ constexpr int ws_a_ns(int l) { return l == 1 ? WS_NS1 : (l == 2 ? WS_NS2 : WS_NS3); }

// position of a wave in the element stream of its workgroup: item j (the j-th tile of this workgroup: tile blockIdx.x +
// j * gridDim.x), element e of that tile.  This kernel runs un-split node ranges only (BwdArgs::ns <= 1: large batches), so
// every tile has the same elements: node 0, [tangent of node 0], nodes 1..n.
struct WsCursor {
    int j, e;
};
struct WsShape {
    int ne, tan0, nit;           // elements per tile, 1 if there is a tangent element, tiles of this workgroup
};
__device__ __forceinline__ unsigned ws_grp(const WsCursor& c) { return blockIdx.x + (unsigned)c.j * gridDim.x; }
__device__ __forceinline__ bool ws_is_tan(const WsShape& sh, const WsCursor& c) { return sh.tan0 && c.e == 1; }
__device__ __forceinline__ int ws_node(const WsShape& sh, const WsCursor& c) { return sh.tan0 ? (c.e <= 1 ? 0 : c.e - 1) : c.e; }
__device__ __forceinline__ WsCursor ws_next(const WsShape& sh, WsCursor c) {
    if (++c.e == sh.ne) { ++c.j; c.e = 0; }
    return c;
}
// ring of LDS tiles: ushort offset of the current tile, advanced once per step (no division in the step loop)
template <int NS, int STRIDE>
__device__ __forceinline__ void ws_adv(int& off) { off += STRIDE; if (off == NS * STRIDE) off = 0; }
template <int NS, int STRIDE>
__host__ __device__ constexpr int ws_ring0(int delay) { return ((WS_BIAS - delay) % NS) * STRIDE; }

#ifdef UMNN_WS_VGPR_ACC
#define WS_VGPR_HINT(x) asm volatile("" : "+v"(x))
#else
#define WS_VGPR_HINT(x) (void)0
#endif
typedef float ws_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ ws_f32x16 ws_mfma32(u32x4 a, u32x4 b, ws_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x2 ws_tr_read(const unsigned short* src) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4v __attribute__((address_space(3)))*)(src)));
}

// ============================================================================================================ wave C
template <int NRL>
__device__ __forceinline__ void ws_role_C(const BwdBf16Args& args, unsigned short* lds16, int S, const WsShape sh,
                                          unsigned wave_global) {
    constexpr int NLIVE = NRL > 0 ? NRL : 4 * BT;
    constexpr int NPAIR = (NLIVE + 1) / 2;
    const BwdArgs& a = args.b;
    const MlpDev& m = a.m;
    const int lane = threadIdx.x & 63, g = lane >> 4, p = lane & 15;
    const int H1 = m.width[1], E = a.E, d = a.d, n = a.n;
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    const int own = p * TRS + g * 16;
    // transposing reads for a 32 x 32 x 16 operand: lane group g = (hf = g >> 1, column half g & 1) reads the 4 x 16 blocks of
    // points 8 hf + 4 half + (0..3) x slots 32 tau + 16 (g & 1) + (0..15), half = 0, 1  (+ piece * 16 * TRS + 32 * tau + half * 4 * TRS)
    const int trb = (8 * (g >> 1) + (p >> 2)) * TRS + 16 * (g & 1) + 4 * (p & 3);

    float w1x[BT][4];
    {
        const float* __restrict__ W0 = m.W[0];
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = feat_of(t, r, g);
                w1x[t][r] = f < H1 ? W0[f * (1 + E)] : 0.f;
            }
    }
    // dW_l as 2 x 2 tiles of 32 x 32 (v_mfma_f32_32x32x16_bf16, K = the 16 points of the tile): rows = slots of delta_{l+1},
    // columns = slots of a_l.  12 matrix instructions per layer and node (3 cross terms x 4 tiles) instead of 32 of the
    // 16x16x32 form, each with a 32-cycle shadow for the vector work of layer 1.
    ws_f32x16 dW[3][2][2];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int to = 0; to < 2; ++to)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int v = 0; v < 16; ++v) dW[j][to][ti][v] = 0.f;

    const int nit = sh.nit;
    WsCursor cu{0, 0};
    float xv = 0.f, x0v = 0.f, dxv = 0.f;
    f32x4 c[BT];
#pragma unroll
    for (int t = 0; t < BT; ++t) c[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto new_item = [&]() __attribute__((always_inline)) {
        const long long q = (long long)ws_grp(cu) * 16 + p;
        const long long qq = q < a.NI ? q : a.NI - 1;
        xv = io_ld(a.x, qq, a.x_bf16);
        x0v = a.x0 ? io_ld(a.x0, qq, a.x_bf16) : 0.f;
        dxv = xv - x0v;
        const long long bi = qq / d;
        const IoView hb = IoView{a.h, a.h_bf16} + (bi * ((long long)E * d) + (qq - bi * d));
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
    };
    if (nit > 0) new_item();

    float actF[BT][4];
    unsigned qF[8][NPF];
    float rf0[8], rf1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        rf0[j] = rf1[j] = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < NPF; ++k2) qF[j][k2] = 0u;
    }
#pragma unroll
    for (int t = 0; t < BT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) actF[t][r] = 0.f;

#ifdef UMNN_WS_TIMING
    unsigned long long tt[4] = {0, 0, 0, 0};
#endif
    // rings of the tiles this wave touches (element s - delay)
    int rA3 = ws_ring0<WS_NS3, WS_TILE>(7), rA2 = ws_ring0<WS_NS2, WS_TILE>(8), rA1 = ws_ring0<WS_NS1, WS_TILE>(9);
    int rD4 = ws_ring0<2, WS_TILE>(7), rD3 = ws_ring0<2, WS_TILE>(8), rD2 = ws_ring0<2, WS_TILE>(9);
    int rO1 = ws_ring0<WS_NS1, WS_TILE>(0), rO3 = ws_ring0<2, WS_P3>(0);
    // node position of the current element (the table value of the NEXT element is fetched a step ahead)
    bool live = cu.j < nit;
    bool is_tan = live && ws_is_tan(sh, cu);
    float tk;
    {
        const int k = ws_node(sh, cu);
        const float uu = a.ccs[k] + 1.f;
        tk = (k == 0) ? xv : __fadd_rn(x0v, __fmul_rn(dxv, uu) * 0.5f);
    }
    for (int s = 0; s < S; ++s) {
#ifdef UMNN_WS_TIMING
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
        const WsCursor nx = live ? ws_next(sh, cu) : cu;
        const int kn = ws_node(sh, nx);
        const float ccs_n = a.ccs[kn];
        const unsigned short* A3 = lds16 + WS_OFF_A3 + rA3 + trb;
        const unsigned short* A2 = lds16 + WS_OFF_A2 + rA2 + trb;
        const unsigned short* A1 = lds16 + WS_OFF_A1 + rA1 + trb;
        const unsigned short* D4 = lds16 + WS_OFF_D + 4 * WS_TILE + rD4 + trb;
        const unsigned short* D3 = lds16 + WS_OFF_D + 2 * WS_TILE + rD3 + trb;
        const unsigned short* D2 = lds16 + WS_OFF_D + 0 * WS_TILE + rD2 + trb;
        unsigned short* const O1 = lds16 + WS_OFF_A1 + rO1 + own;
        unsigned short* const O1p3 = lds16 + WS_OFF_P3 + rO3 + own;

        u32x4 opA[2][2][NPB], opB[2][2][NPB];       // dW operands of a layer [32-slot block][piece], double-buffered by layer parity
        auto load_ops = [&](auto bc, const unsigned short* Dt, const unsigned short* At, auto ic) __attribute__((always_inline)) {
            constexpr int b = decltype(bc)::value, i = decltype(ic)::value;       // i = 0..7: one operand (two reads) per call
            constexpr int isB = i / 4, tau = (i % 4) / 2, piece = i % 2;
            const unsigned short* src = (isB ? At : Dt) + piece * 16 * TRS + 32 * tau;
            const u32x2 x = ws_tr_read(src);
            const u32x2 y = ws_tr_read(src + 4 * TRS);
            if constexpr (isB) opB[b][tau][piece] = u32x4{x[0], x[1], y[0], y[1]};
            else opA[b][tau][piece] = u32x4{x[0], x[1], y[0], y[1]};
        };
        // operands of the first layer (dW_3); then the layer-1 activations (registers only) while those fetches fly
        swp_static_for<8>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int o = i < 2 ? 2 * i : (i < 4 ? 4 + 2 * (i - 2) : (i < 6 ? 5 + 2 * (i - 4) : 1 + 2 * (i - 6)));   // A hi, B hi, B lo, A lo
            load_ops(std::integral_constant<int, 0>{}, D4, A3, std::integral_constant<int, o>{});
        });
        // layer 1 of element s (tangent element: w1 . act'(z_1) of node 0)
        auto layer1_reg = [&](auto ec, auto tanc) __attribute__((always_inline)) {
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4;
            constexpr bool TAN = decltype(tanc)::value;
            if constexpr (e < NLIVE) {
                const float z = fmaf(w1x[t][r], tk, c[t][r]);
                if constexpr (TAN) actF[t][r] = w1x[t][r] * (z > 0.f ? 1.f : slope);
                else actF[t][r] = hidden_act_f(z, slope);
            }
        };
        auto pairF = [&](auto jc, auto stc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, st = decltype(stc)::value, t = j / 2, r = 2 * (j & 1);
            if constexpr (j < NPAIR) {
                if constexpr (st == 0) { rf0[j] = actF[t][r]; rf1[j] = actF[t][r + 1]; qF[j][0] = split_stage(rf0[j], rf1[j]); }
                if constexpr (st == 1) qF[j][1] = split_stage(rf0[j], rf1[j]);
                if constexpr (st == 2) qF[j][2] = split_last(rf0[j], rf1[j]);
            }
        };
        auto store_a = [&](auto sc, auto kc) __attribute__((always_inline)) {
            constexpr int ks = decltype(sc)::value, k2 = decltype(kc)::value;
            const u32x4 v = u32x4{qF[4 * ks][k2], qF[4 * ks + 1][k2], qF[4 * ks + 2][k2], qF[4 * ks + 3][k2]};
            if constexpr (k2 < NPB) *reinterpret_cast<u32x4*>(O1 + k2 * 16 * TRS + ks * 8) = v;
            else *reinterpret_cast<u32x4*>(O1p3 + ks * 8) = v;
        };
#ifdef UMNN_WS_TIMING
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
        // (only these sixteen registers differ for a tangent element: the branch stays outside the matrix loop, whose 192
        // accumulators must not meet at a control-flow join)
        if (is_tan) swp_static_for<16>([&](auto ec) { layer1_reg(ec, std::true_type{}); });
        else swp_static_for<16>([&](auto ec) { layer1_reg(ec, std::false_type{}); });
        __builtin_amdgcn_sched_barrier(0);
        {
            swp_static_for<36>([&](auto nc) {
                constexpr int nn = decltype(nc)::value, li = nn / 12, idx = nn % 12;
                // (cross terms outermost: an accumulator is touched every fourth instruction)
                constexpr int term = idx / 4, to = (idx % 4) / 2, ti = idx % 2;
                constexpr int pa = term == 2 ? 1 : 0, pb = term == 1 ? 1 : 0;
                constexpr int b = li & 1;
                dW[2 - li][to][ti] = ws_mfma32(opA[b][to][pa], opB[b][ti][pb], dW[2 - li][to][ti]);
                // operands of the next layer, one per slot, into the buffer the layer before this one has finished with
                if constexpr (li < 2 && idx >= 2 && idx < 10) {
                    if constexpr (li == 0) load_ops(std::integral_constant<int, 1>{}, D3, A2, std::integral_constant<int, idx - 2>{});
                    else load_ops(std::integral_constant<int, 0>{}, D2, A1, std::integral_constant<int, idx - 2>{});
                }
                // the split of a_1 behind the first slots: pair j's three rounding stages at slots j, j + 1, j + 2; then the stores
                if constexpr (nn < 8) pairF(std::integral_constant<int, nn>{}, std::integral_constant<int, 0>{});
                if constexpr (nn >= 1 && nn < 9) pairF(std::integral_constant<int, nn - 1>{}, std::integral_constant<int, 1>{});
                if constexpr (nn >= 2 && nn < 10) pairF(std::integral_constant<int, nn - 2>{}, std::integral_constant<int, 2>{});
                if constexpr (nn >= 6 && nn < 9) store_a(std::integral_constant<int, 0>{}, std::integral_constant<int, nn - 6>{});
                if constexpr (nn >= 10 && nn < 13) store_a(std::integral_constant<int, 1>{}, std::integral_constant<int, nn - 10>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        }
#ifdef UMNN_WS_TIMING
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#endif
        // next element: its item data if it opens a new tile, its node position from the table value fetched above
        if (live) {
            const bool crossed = nx.j != cu.j;
            cu = nx;
            live = cu.j < nit;
            if (crossed && live) new_item();
            is_tan = live && ws_is_tan(sh, cu);
            const float uu = ccs_n + 1.f;
            tk = (kn == 0) ? xv : __fadd_rn(x0v, __fmul_rn(dxv, uu) * 0.5f);
        }
        ws_adv<WS_NS3, WS_TILE>(rA3); ws_adv<WS_NS2, WS_TILE>(rA2); ws_adv<WS_NS1, WS_TILE>(rA1);
        ws_adv<2, WS_TILE>(rD4); ws_adv<2, WS_TILE>(rD3); ws_adv<2, WS_TILE>(rD2);
        ws_adv<WS_NS1, WS_TILE>(rO1); ws_adv<2, WS_P3>(rO3);
#ifdef UMNN_WS_TIMING
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#ifdef UMNN_WS_TIMING
        const unsigned long long t4 = __builtin_amdgcn_s_memtime();
        tt[0] += t1 - t0; tt[1] += t2 - t1; tt[2] += t3 - t2; tt[3] += t4 - t3;
#endif
    }
#ifdef UMNN_WS_TIMING
    if (args.tz2 && lane == 0) {
        double* o = reinterpret_cast<double*>(const_cast<float*>(args.tz2)) + (size_t)wave_global * 6;
        for (int j = 0; j < 4; ++j) o[j] = (double)tt[j];
        o[4] = (double)S; o[5] = 0.0;
    }
#endif

    // ---- this wave's d_theta slice: the hidden layers' dW (rows / columns run over register slots, see slot_feature; 32 x 32
    // result layout: column lane & 31, register v = 4 i + r <-> row 8 i + 4 (lane >> 5) + r)
    float* part = a.partials + (size_t)wave_global * a.n_params;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int l = 1 + j;
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
                        if (idx >= 0) part[idx] = dW[j][to][ti][v];
                    }
                }
    }
}

// ============================================================================================================ waves G1..G3
template <int NRL, int LAYER>
__device__ __forceinline__ void ws_role_G(const BwdBf16Args& args, unsigned short* lds16, int S, const WsShape sh,
                                          unsigned wave_global, const u32x4 (&Wf)[BT][BKS][NPF], const u32x4 (&WT)[BT][BKS][NPB]) {
    constexpr int NLIVE = NRL > 0 ? NRL : 4 * BT;
    constexpr int NPAIR = (NLIVE + 1) / 2;
    constexpr int L = 4;
    constexpr int DF = 2 * LAYER - 1;                  // forward GEMM works on element s - DF, its vector work one step later
    constexpr int DP = 2 * LAYER;
    constexpr int DB = 10 - LAYER;                     // W^T GEMM and its vector work: element s - DB
    constexpr bool IS_OUT = LAYER == 3, IS_TAIL = LAYER == 1;
    const BwdArgs& a = args.b;
    const MlpDev& m = a.m;
    const int lane = threadIdx.x & 63, g = lane >> 4, p = lane & 15;
    const int H1 = m.width[1], HL = m.width[L], E = a.E, n = a.n;
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    const int own = p * TRS + g * 16;

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
    f32x4 dwo[BT], dW1x[BT], dcs[BT];
#pragma unroll
    for (int t = 0; t < BT; ++t) { dwo[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dW1x[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dcs[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // cursors: cp = element of the forward vector work (s - DP), cb = element of the backward vector work (s - DB; G1 only)
    const int nit = sh.nit;
    WsCursor cp{0, 0}, cb{0, 0};
    // per-item data: G3 (cotangents, Leibniz terms), G1 (node positions for dW1[:,0], dc)
    float xvP = 0.f, x0vP = 0.f, gvP = 0.f, gfxvP = 0.f, cotbase = 0.f;
    float xvB = 0.f, x0vB = 0.f, dxvB = 0.f;
    float fxv = 0.f, fx0v = 0.f, dfdt = 0.f, fp0 = 0.f;
    auto new_item_P = [&]() __attribute__((always_inline)) {
        if constexpr (IS_OUT) {
            const long long q = (long long)ws_grp(cp) * 16 + p;
            const bool ok = q < a.NI;
            const long long qq = ok ? q : a.NI - 1;
            xvP = io_ld(a.x, qq, a.x_bf16);
            x0vP = a.x0 ? io_ld(a.x0, qq, a.x_bf16) : 0.f;
            gvP = ok ? io_ld(a.g, qq, a.x_bf16) : 0.f;
            gfxvP = (ok && a.gfx) ? io_ld(a.gfx, qq, a.x_bf16) : 0.f;
            cotbase = gvP * (xvP - x0vP) * 0.5f;
            fxv = 0.f; fx0v = 0.f; dfdt = 0.f;
        }
    };
    auto new_item_B = [&]() __attribute__((always_inline)) {
        if constexpr (IS_TAIL) {
            const long long q = (long long)ws_grp(cb) * 16 + p;
            const long long qq = q < a.NI ? q : a.NI - 1;
            xvB = io_ld(a.x, qq, a.x_bf16);
            x0vB = a.x0 ? io_ld(a.x0, qq, a.x_bf16) : 0.f;
            dxvB = xvB - x0vB;
        }
    };
    if (nit > 0) { new_item_P(); new_item_B(); }

    f32x4 acc[BT], nd[BT];
    float actF[BT][4], delta[BT][4], tanv[BT][4];
#pragma unroll
    for (int t = 0; t < BT; ++t) {
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        nd[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) { actF[t][r] = 0.f; delta[t][r] = 0.f; tanv[t][r] = 0.f; }
    }
    unsigned qF[8][NPF], qB[8][NPB], q4[8][NPB];
    float rf0[8], rf1[8], rb0[8], rb1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        rf0[j] = rf1[j] = rb0[j] = rb1[j] = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < NPF; ++k2) qF[j][k2] = 0u;
#pragma unroll
        for (int k2 = 0; k2 < NPB; ++k2) { qB[j][k2] = 0u; q4[j][k2] = 0u; }
    }
    BFrag<NPB> bd;                                      // B operand of the W^T GEMM (G3: its own delta_4 of the step before)
#pragma unroll
    for (int s2 = 0; s2 < BKS; ++s2)
#pragma unroll
        for (int k2 = 0; k2 < NPB; ++k2) bd.v[s2][k2] = u32x4{0u, 0u, 0u, 0u};
    float sdot = 0.f, doutN = 0.f, fpN = 0.f;

#ifdef UMNN_WS_TIMING
    unsigned long long tt[4] = {0, 0, 0, 0};
#endif
    constexpr int LO = LAYER < 3 ? LAYER + 1 : 3;       // layer of the activation tile this wave writes (G3 writes none)
    int rAin = ws_ring0<ws_a_ns(LAYER), WS_TILE>(DF), rAin3 = ws_ring0<2, WS_P3>(DF), rAsg = ws_ring0<ws_a_ns(LAYER), WS_TILE>(DB);
    int rDin = ws_ring0<2, WS_TILE>(DB), rAout = ws_ring0<ws_a_ns(LO), WS_TILE>(DP), rAout3 = ws_ring0<2, WS_P3>(DP);
    int rD4 = ws_ring0<2, WS_TILE>(DP), rDout = ws_ring0<2, WS_TILE>(DB);
    // table values of the current elements (those of the next step's elements are fetched a step ahead, below)
    float ccwP = 0.f, tkB = 0.f;
    if constexpr (IS_OUT) ccwP = a.ccw[ws_node(sh, cp)];
    if constexpr (IS_TAIL) {
        const int kB = ws_node(sh, cb);
        const float uu = a.ccs[kB] + 1.f;
        tkB = (kB == 0) ? xvB : __fadd_rn(x0vB, __fmul_rn(dxvB, uu) * 0.5f);
    }
    for (int s = 0; s < S; ++s) {
#ifdef UMNN_WS_TIMING
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
        const bool liveP = s >= DP && cp.j < nit;
        const bool liveB = s >= DB && cb.j < nit;
        const bool tanP = liveP && ws_is_tan(sh, cp);
        const int kP = ws_node(sh, cp);
        const WsCursor nxP = liveP ? ws_next(sh, cp) : cp;
        WsCursor nxB = cb;
        float ccw_n = 0.f, ccs_n = 0.f;
        int kBn = 0;
        if constexpr (IS_OUT) ccw_n = a.ccw[ws_node(sh, nxP)];
        if constexpr (IS_TAIL) {
            if (liveB) nxB = ws_next(sh, cb);
            kBn = ws_node(sh, nxB);
            ccs_n = a.ccs[kBn];
        }
        if constexpr (IS_OUT) {
            if (liveP && cp.e == 0) new_item_P();
        }
        // ---- LDS tiles of this step
        const unsigned short* Ain = lds16 + ws_a_off(LAYER) + rAin + own;                                  // a_l[s - DF]
        const unsigned short* Ain3 = lds16 + WS_OFF_P3 + (LAYER - 1) * 2 * WS_P3 + rAin3 + own;
        const unsigned short* Asg = lds16 + ws_a_off(LAYER) + rAsg + own;                                  // a_l[s - DB]
        const unsigned short* Din = lds16 + WS_OFF_D + (LAYER + 1 - 2) * 2 * WS_TILE + rDin + own;         // delta_{l+1}[s - DB]
        unsigned short* const Aout = lds16 + ws_a_off(LO) + rAout + own;                                   // a_{l+1}[s - DP]
        unsigned short* const Aout3 = lds16 + WS_OFF_P3 + (LO - 1) * 2 * WS_P3 + rAout3 + own;
        unsigned short* const D4out = lds16 + WS_OFF_D + 4 * WS_TILE + rD4 + own;                          // G3: delta_4[s - 6]
        unsigned short* const Dout = lds16 + WS_OFF_D + (LAYER >= 2 ? LAYER - 2 : 0) * 2 * WS_TILE + rDout + own;   // delta_l[s - DB]

        // ---- operands of the two GEMMs and the sign piece of a_l
        BFrag<NPF> bf;
        u32x4 sg[BKS];
        if constexpr (!IS_OUT) {
#pragma unroll
            for (int s2 = 0; s2 < BKS; ++s2)
#pragma unroll
                for (int k2 = 0; k2 < NPB; ++k2) bd.v[s2][k2] = *reinterpret_cast<const u32x4*>(Din + k2 * 16 * TRS + s2 * 8);
        }
#pragma unroll
        for (int s2 = 0; s2 < BKS; ++s2) {
#pragma unroll
            for (int k2 = 0; k2 < NPB; ++k2) bf.v[s2][k2] = *reinterpret_cast<const u32x4*>(Ain + k2 * 16 * TRS + s2 * 8);
            bf.v[s2][2] = *reinterpret_cast<const u32x4*>(Ain3 + s2 * 8);
        }
#pragma unroll
        for (int s2 = 0; s2 < BKS; ++s2) sg[s2] = *reinterpret_cast<const u32x4*>(Asg + s2 * 8);

        // ---- micro-operations
        auto act_reg = [&](auto ec, auto tanc) __attribute__((always_inline)) {          // forward vector work, one register
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4, j = e / 2;
            constexpr bool TAN = decltype(tanc)::value;
            if constexpr (e < NLIVE) {
                if constexpr (!IS_OUT) {
                    if constexpr (TAN) {
                        // tangent element: times act'(a_{l+1}) of the element before (node 0), read off its leading bf16 piece
                        const unsigned u = qF[j][0];
                        const bool pos = (e & 1) ? ((int)u > 0xffff) : ((short)(u & 0xffffu) > 0);
                        actF[t][r] = acc[t][r] * (pos ? 1.f : slope);
                    } else {
                        actF[t][r] = hidden_act_f(acc[t][r], slope);
                    }
                } else {
                    if constexpr (TAN) tanv[t][r] = acc[t][r] * (actF[t][r] > 0.f ? 1.f : slope);
                    else actF[t][r] = hidden_act_f(acc[t][r], slope);
                }
            }
        };
        auto pairF = [&](auto jc, auto stc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, st = decltype(stc)::value, t = j / 2, r = 2 * (j & 1);
            if constexpr (j < NPAIR) {
                if constexpr (st == 0) { rf0[j] = actF[t][r]; rf1[j] = actF[t][r + 1]; qF[j][0] = split_stage(rf0[j], rf1[j]); }
                if constexpr (st == 1) qF[j][1] = split_stage(rf0[j], rf1[j]);
                if constexpr (st == 2) qF[j][2] = split_last(rf0[j], rf1[j]);
            }
        };
        auto store_a = [&](auto sc, auto kc) __attribute__((always_inline)) {
            constexpr int ks = decltype(sc)::value, k2 = decltype(kc)::value;
            const u32x4 v = u32x4{qF[4 * ks][k2], qF[4 * ks + 1][k2], qF[4 * ks + 2][k2], qF[4 * ks + 3][k2]};
            if constexpr (k2 < NPB) *reinterpret_cast<u32x4*>(Aout + k2 * 16 * TRS + ks * 8) = v;
            else *reinterpret_cast<u32x4*>(Aout3 + ks * 8) = v;
        };
        // G3: output layer of element s - 6 (same expressions as cc_bwd_swp_kernel.h)
        auto out_dot = [&](auto ec, auto tanc) __attribute__((always_inline)) {
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4;
            constexpr bool TAN = decltype(tanc)::value;
            if constexpr (e == 0) sdot = 0.f;
            if constexpr (e < NLIVE) sdot = fmaf(wout[t][r], TAN ? tanv[t][r] : actF[t][r], sdot);
        };
        auto out_scalar = [&]() __attribute__((always_inline)) {
            const float sd = group_allreduce(sdot);
            const bool sig = m.out_act != UMNN_OUT_ELU_PLUS_ONE;
            const float ex = __expf(sig ? -sd : sd);
            const float s1 = 1.f / (1.f + ex);
            const float f = sig ? s1 : (sd > 0.f ? sd + 1.f : ex);
            const float fpn = sig ? s1 * (1.f - s1) : (sd > 0.f ? 1.f : ex);
            const bool node = liveP && !tanP;
            if (node && kP == 0) { fxv = f; fp0 = fpn; }
            if (node && kP == n) fx0v = f;
            const float rinv = -__frcp_rn(f * f);
            const float invs = a.inv_f ? rinv : 1.f;
            const float cot = fmaf(cotbase * invs, ccwP, kP == 0 ? gfxvP : 0.f) * (node ? 1.f : 0.f);
            doutN = cot * fpn;
            // tangent element: sdot is w_out . d a_L / d t at node 0 -> d f / d t; no cotangent flows back (cot = 0 above)
            if (tanP) dfdt = fp0 * sd;
        };
        auto out_reg = [&](auto ec) __attribute__((always_inline)) {       // dwo += dout a_L ; delta_L = dout wout act'(a_L)
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4;
            if constexpr (e < NLIVE) {
                dwo[t][r] = fmaf(doutN, actF[t][r], dwo[t][r]);
                delta[t][r] = doutN * wout[t][r] * (actF[t][r] > 0.f ? 1.f : slope);
            }
        };
        auto pair4 = [&](auto jc, auto stc) __attribute__((always_inline)) {              // split of delta_4 (G3)
            constexpr int j = decltype(jc)::value, st = decltype(stc)::value, t = j / 2, r = 2 * (j & 1);
            if constexpr (j < NPAIR) {
                if constexpr (st == 0) { rb0[j] = delta[t][r]; rb1[j] = delta[t][r + 1]; q4[j][0] = split_stage(rb0[j], rb1[j]); }
                if constexpr (st == 1) q4[j][1] = split_last(rb0[j], rb1[j]);
            }
        };
        auto commit4 = [&](auto sc) __attribute__((always_inline)) {                      // delta_4 K-step: next step's operand, LDS tile
            constexpr int ks = decltype(sc)::value;
#pragma unroll
            for (int k2 = 0; k2 < NPB; ++k2) {
                bd.v[ks][k2] = u32x4{q4[4 * ks][k2], q4[4 * ks + 1][k2], q4[4 * ks + 2][k2], q4[4 * ks + 3][k2]};
                *reinterpret_cast<u32x4*>(D4out + k2 * 16 * TRS + ks * 8) = bd.v[ks][k2];
            }
        };
        // backward vector work of element s - DB: delta_l = (W_l^T delta_{l+1}) . act'(a_l)
        float dl[BT][4];
        auto s7_reg = [&](auto ec) __attribute__((always_inline)) {
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4;
            if constexpr (e < NLIVE) {
                dl[t][r] = nd[t][r] * act_grad_q(sg, t, r, slope);
                if constexpr (IS_TAIL) {
                    dcs[t][r] += dl[t][r];
                    dW1x[t][r] = fmaf(dl[t][r], tkB, dW1x[t][r]);
                }
            } else {
                dl[t][r] = 0.f;
            }
        };
        auto pairB = [&](auto jc, auto stc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, st = decltype(stc)::value, t = j / 2, r = 2 * (j & 1);
            if constexpr (j < NPAIR && !IS_TAIL) {
                if constexpr (st == 0) { rb0[j] = dl[t][r]; rb1[j] = dl[t][r + 1]; qB[j][0] = split_stage(rb0[j], rb1[j]); }
                if constexpr (st == 1) qB[j][1] = split_last(rb0[j], rb1[j]);
            }
        };
        auto store_d = [&](auto sc, auto kc) __attribute__((always_inline)) {
            constexpr int ks = decltype(sc)::value, k2 = decltype(kc)::value;
            if constexpr (!IS_TAIL)
                *reinterpret_cast<u32x4*>(Dout + k2 * 16 * TRS + ks * 8) = u32x4{qB[4 * ks][k2], qB[4 * ks + 1][k2], qB[4 * ks + 2][k2], qB[4 * ks + 3][k2]};
        };

#ifdef UMNN_WS_TIMING
        unsigned long long tmid = 0;
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
        // ---- the activations of element s - DP first (registers only: the operand fetches above are still in flight).  Only
        // these differ for a tangent element; the branch stays outside the matrix regions.
        if (tanP) {
            swp_static_for<16>([&](auto ec) { act_reg(ec, std::true_type{}); });
            if constexpr (IS_OUT) swp_static_for<16>([&](auto ec) { out_dot(ec, std::true_type{}); });
        } else {
            swp_static_for<16>([&](auto ec) { act_reg(ec, std::false_type{}); });
            if constexpr (IS_OUT) swp_static_for<16>([&](auto ec) { out_dot(ec, std::false_type{}); });
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            // ---- region B: W^T GEMM of element s - DB (24 MFMAs); behind it the rest of the forward vector work of element s - DP
            swp_static_for<24>([&](auto nc) {
                constexpr int nn = decltype(nc)::value, s2 = nn / 12, idx = nn % 12;
                if constexpr (idx < 8) {
                    constexpr int t = idx % 4, ba = idx / 4;
                    nd[t] = mfma_bf16(WT[t][s2][0], bd.v[s2][ba], (s2 == 0 && idx < 4) ? zero : nd[t]);
                } else {
                    constexpr int t = idx - 8;
                    nd[t] = mfma_bf16(WT[t][s2][1], bd.v[s2][0], nd[t]);
                }
                if constexpr (!IS_OUT) {
                    if constexpr (nn < 16 && (nn % 2) == 0) pairF(std::integral_constant<int, nn / 2>{}, std::integral_constant<int, 0>{});
                    if constexpr (nn >= 1 && nn < 17 && (nn % 2) == 1) pairF(std::integral_constant<int, (nn - 1) / 2>{}, std::integral_constant<int, 1>{});
                    if constexpr (nn >= 2 && nn < 18 && (nn % 2) == 0) pairF(std::integral_constant<int, (nn - 2) / 2>{}, std::integral_constant<int, 2>{});
                    if constexpr (nn >= 9 && nn < 12) store_a(std::integral_constant<int, 0>{}, std::integral_constant<int, nn - 9>{});
                    if constexpr (nn >= 17 && nn < 20) store_a(std::integral_constant<int, 1>{}, std::integral_constant<int, nn - 17>{});
                } else {
                    if constexpr (nn == 1) out_scalar();
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            // (results wanted in architectural registers: the vector work reads every one of them)
#pragma unroll
            for (int t = 0; t < BT; ++t) { WS_VGPR_HINT(nd[t]); }
#ifdef UMNN_WS_TIMING
            tmid = __builtin_amdgcn_s_memtime();
#endif
            // ---- region F: forward GEMM of element s - DF (48 MFMAs); behind it the backward vector work of element s - DB
            // (G3 first: cotangent of the output layer for element s - 6, the operand of the next step's W^T GEMM)
            swp_static_for<48>([&](auto nc) {
                constexpr int nn = decltype(nc)::value, s2 = nn / 24, idx = nn % 24;
                if constexpr (idx < 12) {
                    constexpr int t = idx % 4, ba = idx / 4;
                    acc[t] = mfma_bf16(Wf[t][s2][0], bf.v[s2][ba], (s2 == 0 && idx < 4) ? zero : acc[t]);
                } else if constexpr (idx < 20) {
                    constexpr int t = (idx - 12) % 4, ba = (idx - 12) / 4;
                    acc[t] = mfma_bf16(Wf[t][s2][1], bf.v[s2][ba], acc[t]);
                } else {
                    constexpr int t = idx - 20;
                    acc[t] = mfma_bf16(Wf[t][s2][2], bf.v[s2][0], acc[t]);
                }
                if constexpr (IS_OUT) {
                    if constexpr (nn < 16) out_reg(std::integral_constant<int, nn>{});
                    if constexpr (nn >= 2 && nn < 18 && (nn % 2) == 0) pair4(std::integral_constant<int, (nn - 2) / 2>{}, std::integral_constant<int, 0>{});
                    if constexpr (nn >= 3 && nn < 19 && (nn % 2) == 1) pair4(std::integral_constant<int, (nn - 3) / 2>{}, std::integral_constant<int, 1>{});
                    if constexpr (nn == 11) commit4(std::integral_constant<int, 0>{});
                    if constexpr (nn == 19) commit4(std::integral_constant<int, 1>{});
                    constexpr int o = 20;
                    if constexpr (nn >= o && nn < o + 16) s7_reg(std::integral_constant<int, nn - o>{});
                    if constexpr (nn >= o + 2 && nn < o + 18 && ((nn - o) % 2) == 0) pairB(std::integral_constant<int, (nn - o - 2) / 2>{}, std::integral_constant<int, 0>{});
                    if constexpr (nn >= o + 3 && nn < o + 19 && ((nn - o) % 2) == 1) pairB(std::integral_constant<int, (nn - o - 3) / 2>{}, std::integral_constant<int, 1>{});
                    if constexpr (nn == o + 11 || nn == o + 12) store_d(std::integral_constant<int, 0>{}, std::integral_constant<int, nn - o - 11>{});
                    if constexpr (nn == o + 19 || nn == o + 20) store_d(std::integral_constant<int, 1>{}, std::integral_constant<int, nn - o - 19>{});
                } else {
                    // spread over the first 40 slots: register e at slot 2e, pair j's stages at 4j + 4 / 4j + 6
                    if constexpr (nn < 32 && (nn % 2) == 0) s7_reg(std::integral_constant<int, nn / 2>{});
                    if constexpr (nn >= 4 && nn < 36 && (nn % 4) == 0) pairB(std::integral_constant<int, (nn - 4) / 4>{}, std::integral_constant<int, 0>{});
                    if constexpr (nn >= 6 && nn < 38 && (nn % 4) == 2) pairB(std::integral_constant<int, (nn - 6) / 4>{}, std::integral_constant<int, 1>{});
                    if constexpr (nn == 21 || nn == 23) store_d(std::integral_constant<int, 0>{}, std::integral_constant<int, (nn - 21) / 2>{});
                    if constexpr (nn == 37 || nn == 39) store_d(std::integral_constant<int, 1>{}, std::integral_constant<int, (nn - 37) / 2>{});
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
#pragma unroll
        for (int t = 0; t < BT; ++t) { WS_VGPR_HINT(acc[t]); }

#ifdef UMNN_WS_TIMING
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
#endif
        // ---- item boundaries (outside the scheduled regions: uniform branches)
        if constexpr (IS_OUT) {
            if (liveP && cp.e == sh.ne - 1) {
                const long long q = (long long)ws_grp(cp) * 16 + p;
                if (q < a.NI && g == 0) {
                    if (a.dx) io_st(a.dx, q, fmaf(gfxvP, dfdt, fxv * gvP), a.x_bf16);
                    if (a.dx0) io_st(a.dx0, q, -fx0v * gvP, a.x_bf16);
                }
            }
        }
        if constexpr (IS_TAIL) {
            if (liveB && cb.e == sh.ne - 1) {
                const long long q = (long long)ws_grp(cb) * 16 + p;
                if (q < a.NI) {
#pragma unroll
                    for (int t = 0; t < BT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int f = feat_of(t, r, g);
                            if (f < H1) a.dc[q * H1 + f] = dcs[t][r];
                        }
                }
#pragma unroll
                for (int t = 0; t < BT; ++t) dcs[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        cp = nxP;
        ccwP = ccw_n;
        if constexpr (IS_TAIL) {
            if (liveB) {
                const bool crossed = nxB.j != cb.j;
                cb = nxB;
                if (crossed && cb.j < nit) new_item_B();
                const float uu = ccs_n + 1.f;
                tkB = (kBn == 0) ? xvB : __fadd_rn(x0vB, __fmul_rn(dxvB, uu) * 0.5f);
            }
        }
        ws_adv<ws_a_ns(LAYER), WS_TILE>(rAin); ws_adv<2, WS_P3>(rAin3); ws_adv<ws_a_ns(LAYER), WS_TILE>(rAsg);
        ws_adv<2, WS_TILE>(rDin); ws_adv<ws_a_ns(LO), WS_TILE>(rAout); ws_adv<2, WS_P3>(rAout3);
        ws_adv<2, WS_TILE>(rD4); ws_adv<2, WS_TILE>(rDout);
#ifdef UMNN_WS_TIMING
        const unsigned long long t3b = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#ifdef UMNN_WS_TIMING
        const unsigned long long t4 = __builtin_amdgcn_s_memtime();
        tt[0] += t1 - t0; tt[1] += tmid - t1; tt[2] += t3 - tmid; tt[3] += t4 - t3b;
        (void)t3;
#endif
    }
#ifdef UMNN_WS_TIMING
    if (args.tz2 && lane == 0) {
        double* o = reinterpret_cast<double*>(const_cast<float*>(args.tz2)) + (size_t)wave_global * 6;
        for (int j = 0; j < 4; ++j) o[j] = (double)tt[j];
        o[4] = (double)S; o[5] = 0.0;
    }
#endif

    // ---- this wave's d_theta slice
    float* part = a.partials + (size_t)wave_global * a.n_params;
    if constexpr (IS_OUT || IS_TAIL) {
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = IS_OUT ? dwo[t][r] : dW1x[t][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
                const int f = feat_of(t, r, g);
                if (p == 0) {
                    if constexpr (IS_TAIL) {
                        if (f < H1) part[a.poffW[0] + f * (1 + E)] = v;
                    } else {
                        const int idx = f < HL ? a.poffW[L] + f : (f == HL ? a.poffb[L] : -1);
                        if (idx >= 0) part[idx] = v;
                    }
                }
            }
    }
}

template <int NRL>
__global__ __launch_bounds__(UMNN_BLOCK, 1) void cc_bwd_ws_kernel(const BwdBf16Args args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned short* lds16 = reinterpret_cast<unsigned short*>(lds);
    const BwdArgs& a = args.b;
    const MlpDev& m = a.m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- the weights: staged once as fragment images (the other kernels' staging code), then read into the owners' registers
    for (int l = 1; l <= 3; ++l) {
        stage_frag_image<false, NPF>(m, l, lds16 + (l - 1) * WS_IMGF, tid, blockDim.x);
        stage_frag_image<true, NPB>(m, l, lds16 + 3 * WS_IMGF + (l - 1) * WS_IMGT, tid, blockDim.x);
    }
    __syncthreads();
    u32x4 Wf[BT][BKS][NPF], WT[BT][BKS][NPB];
    {
        const int l = wid >= 1 ? wid : 1;
        const unsigned short* imf = lds16 + (l - 1) * WS_IMGF + lane * 8;
        const unsigned short* imt = lds16 + 3 * WS_IMGF + (l - 1) * WS_IMGT + lane * 8;
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int s = 0; s < BKS; ++s) {
#pragma unroll
                for (int k2 = 0; k2 < NPF; ++k2) Wf[t][s][k2] = *reinterpret_cast<const u32x4*>(imf + ((t * BKS + s) * NPF + k2) * FRAG);
#pragma unroll
                for (int k2 = 0; k2 < NPB; ++k2) WT[t][s][k2] = *reinterpret_cast<const u32x4*>(imt + ((t * BKS + s) * NPB + k2) * FRAG);
            }
    }
    // (the matrix instructions take A operands from accumulation registers as well: keep the 160 weight registers there and
    // the architectural half of the file for the vector work)
#pragma unroll
    for (int t = 0; t < BT; ++t)
#pragma unroll
        for (int s = 0; s < BKS; ++s) {
#pragma unroll
            for (int k2 = 0; k2 < NPF; ++k2) asm volatile("" : "+a"(Wf[t][s][k2]));
#pragma unroll
            for (int k2 = 0; k2 < NPB; ++k2) asm volatile("" : "+a"(WT[t][s][k2]));
        }
    __syncthreads();
    for (int i = tid; i < WS_LDS_USHORTS / 8; i += blockDim.x) reinterpret_cast<u32x4*>(lds16)[i] = u32x4{0u, 0u, 0u, 0u};
    WsShape sh;
    sh.tan0 = a.gfx != nullptr ? 1 : 0;
    sh.ne = a.n + 1 + sh.tan0;
    sh.nit = blockIdx.x < a.ngroups ? (int)((a.ngroups - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
    const int S = sh.nit * sh.ne + WS_DEPTH;
    const unsigned wave_global = blockIdx.x * (blockDim.x >> 6) + wid;
    __syncthreads();
    if (wid == 0) ws_role_C<NRL>(args, lds16, S, sh, wave_global);
    else if (wid == 1) ws_role_G<NRL, 1>(args, lds16, S, sh, wave_global, Wf, WT);
    else if (wid == 2) ws_role_G<NRL, 2>(args, lds16, S, sh, wave_global, Wf, WT);
    else ws_role_G<NRL, 3>(args, lds16, S, sh, wave_global, Wf, WT);
}
