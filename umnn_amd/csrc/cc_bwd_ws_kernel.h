// Weight-stationary one-pass backward on the bf16 matrix cores (round 3, second kernel): the waves of a workgroup stop being
// copies of the same program.  Same arithmetic as cc_bwd_swp_kernel.h / cc_bwd_bf16_kernel.h (reference lines
// ParallelNeuralIntegral.py:66-94,110-123): six-term recompute, three-term delta chain, three-term dW on the matrix core.  In
// those kernels every wave needs the whole register file (192 dW accumulators), so ONE wave per SIMD issues a serial stream
// of ~1430 instructions per tile-node (nothing fills its dependency stalls), and every GEMM MFMA pulls a fresh 1 KB weight
// fragment out of LDS.  Here the weights never move and no wave holds more than a quarter of the state: EIGHT
// waves per workgroup, two per SIMD, each <= 256 registers:
//
//   F1, F2, F3   forward GEMM of hidden layer l -> l+1: W_l (3 bf16 pieces, 96 registers) lives in the wave's registers for
//                the whole launch; per step 48 MFMAs on one tile-node (B operands: 6 LDS reads) with the activation / split
//                of the tile-node before behind them.  F3 ends in the output layer: f, dout, d w_out; it leaves dout and the
//                leading piece of a_4 (the sign is all delta_4 needs).
//   B1, B2, B3   W_l^T (2 pieces, 64 registers): per step 24 MFMAs, then delta_l = (W_l^T delta_{l+1}) . act'(a_l), split,
//                stored for the next wave down.  B1 ends in the tail (dc, dW1[:,0]).
//   Ca           layer 1 (a_1 of a new tile-node per step) and dW_3;
//   Cb           delta_4 = dout w_out act'(a_4) (split, stored), dW_2 and dW_1.
//                dW_l += delta_{l+1} (x) a_l as 2 x 2 tiles of 32 x 32 on v_mfma_f32_32x32x16_bf16 (K = the 16 points; 12 MFMAs
//                per layer and tile-node), operands by transposing LDS reads.
//
// Tile-nodes flow through the workgroup as a systolic pipeline over LDS tiles (the [piece][point][slot] tile of the other
// kernels: own-lane b128 reads give B operands back, ds_read_b64_tr_b16 gives the transposed dW operands), one s_barrier per
// step.  Element u of the stream (a node of a tile; node order 0, [tangent of node 0], 1, .., n per tile, tiles one after the
// other) is at:
//   step u      Ca  : a_1[u]                      -> tile A1[u % 11]
//   step u+1/2  F1  : GEMM / activation, a_2[u]   -> tile A2[u % 8]          (GEMM in one step, its vector work in the next:
//   step u+3/4  F2  : a_3[u]                      -> tile A3[u % 5]           the wave always has TWO independent jobs)
//   step u+5/6  F3  : a_4[u], f, dout[u]          -> tile S4[u % 2]
//   step u+7    Cb  : delta_4[u]                  -> tile D4[u % 2]
//   step u+8    B3  : delta_3[u] = W_3^T delta_4 . act'(a_3) -> D3[u % 2];   Ca: dW_3 += delta_4[u] (x) a_3[u]
//   step u+9    B2  : delta_2[u] -> D2[u % 2];                                Cb: dW_2 += delta_3[u] (x) a_2[u]
//   step u+10   B1  : delta_1[u] -> dc, dW1[:,0];                             Cb: dW_1 += delta_2[u] (x) a_1[u]
// LDS: 24 activation tiles + 6 cotangent tiles of 4.5 KB, 8 single-piece tiles of 2.25 KB = 153 KB; no weight images.
// The tangent pass of the g_fx term (d f / d x at node 0) is one extra stream element per tile: Ca sends w1 . act'(z_1), the F
// waves multiply by act'(a_{l+1}(node 0)) -- the signs of the element before -- instead of applying the activation, F3 turns
// it into dfdt and a zero cotangent.  Un-split node ranges only (BwdArgs::ns <= 1: large batches).
//
// Measured at C3 (8192 x 63, n = 100): 12.9-13.1 ms per launch against 13.4-13.6 for the software-pipelined loop.  Per tile-node
// 1135 vector + matrix + LDS instructions (there 1396) and 20 % fewer matrix cycles; what that saves is mostly lost to
// synchronisation: a role wave waits at the step barrier for 28 % of a step on average, and the four long roles (F1, F2, F3, Cb:
// one per SIMD) sit on the pair's 1152 matrix-pipe cycles plus their un-overlapped vector phases.  -DUMNN_WS_TIMING (an
// instrumentation build: results unchanged, s_memtime per role and step) times every role; EXPERIMENTS.md has the numbers and the
// twenty-odd variants that were measured -- their switches no longer live in this file.  Since round 4 this kernel is the
// FALLBACK of the fp16-piece pipeline (cc_bwd_ws16_kernel.h: same pipeline, three-term recompute), the kernel of 1/f launches,
// sigmoid outputs and mid-size batches, and the middle stage of the three-stage backward (FRONT).
#pragma once
#include "cc_bwd_swp_kernel.h"

constexpr int WS_TILE = NPB * 16 * TRS;          // ushorts: the two leading bf16 pieces of a 16-point x 64-feature tile
constexpr int WS_P3 = 16 * TRS;                  // the third piece (lives one step)
constexpr int WS_NS1 = 11, WS_NS2 = 8, WS_NS3 = 5;
constexpr int WS_OFF_A1 = 0;
constexpr int WS_OFF_A2 = WS_OFF_A1 + WS_NS1 * WS_TILE;
constexpr int WS_OFF_A3 = WS_OFF_A2 + WS_NS2 * WS_TILE;
constexpr int WS_OFF_D = WS_OFF_A3 + WS_NS3 * WS_TILE;      // delta_l, l = 2..4: tile (l - 2) * 2 + (u & 1)
constexpr int WS_OFF_P3 = WS_OFF_D + 6 * WS_TILE;           // third piece of a_l, l = 1..3: tile (l - 1) * 2 + (u & 1)
constexpr int WS_OFF_S4 = WS_OFF_P3 + 6 * WS_P3;            // leading piece of a_4 (its sign is all delta_4 needs) + dout in the row padding: tile (u & 1)
constexpr int WS_LDS_USHORTS = WS_OFF_S4 + 2 * WS_P3;
constexpr int WS_DEPTH = 11;                     // steps between an element entering (Ca) and leaving (B1's tail)
constexpr int WS_BIAS = 440;                     // multiple of every ring size: keeps (step - delay) non-negative under %
constexpr int WS_WAVES = 8;                     // waves per workgroup (two per SIMD)
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
__device__ __forceinline__ void ws_adv(int& off) {
    if constexpr (NS == 2) off ^= STRIDE;               // (two-tile rings: one scalar instruction)
    else { off += STRIDE; if (off == NS * STRIDE) off = 0; }
}
template <int NS, int STRIDE>
__host__ __device__ constexpr int ws_ring0(int delay) { return ((WS_BIAS - delay) % NS) * STRIDE; }

// companion work spread over a matrix loop: micro-operation i rides behind slot (3 i) / 2 (two of every three slots carry one)
__host__ __device__ constexpr int ws_op_of_slot(int nn) { return ((((2 * nn + 2) / 3) * 3) / 2 == nn) ? (2 * nn + 2) / 3 : -1; }
typedef float ws_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ ws_f32x16 ws_mfma32(u32x4 a, u32x4 b, ws_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x2 ws_tr_read(const unsigned short* src) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4v __attribute__((address_space(3)))*)(src)));
}

#define WS_STEP_SYNC() __syncthreads()
#ifdef UMNN_WS_TIMING
// per-role cycle sums (s_memtime): prep = operand fetches issued and waited for, work = the role's step (cut in two at slot
// UMNN_WS_TRACE_FRAC percent of its matrix loop when that is defined: one more sample per step, a different build per cut), wait =
// the step barrier
#ifndef UMNN_WS_TRACE_FRAC
#define UMNN_WS_TRACE_FRAC 50
#endif
#define WS_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define WS_TIMING_DECL unsigned long long tt[4] = {0, 0, 0, 0}, tmark = 0
#define WS_MARK(nn, nslots) do { if constexpr ((nn) == ((nslots) * UMNN_WS_TRACE_FRAC) / 100) tmark = __builtin_amdgcn_s_memtime(); } while (0)
#define WS_MARK_HERE() do { tmark = __builtin_amdgcn_s_memtime(); } while (0)
#define WS_TIMING_ACC(t0, t1, t2, t3) do { tt[0] += (t1) - (t0); tt[1] += tmark - (t1); tt[2] += (t2) - tmark; tt[3] += (t3) - (t2); } while (0)
#define WS_TIMING_OUT(S) do { if (args.tz2 && (threadIdx.x & 63) == 0) { \
        double* o = reinterpret_cast<double*>(const_cast<float*>(args.tz2)) + ((size_t)blockIdx.x * WS_WAVES + (threadIdx.x >> 6)) * 6; \
        o[0] = (double)tt[0]; o[1] = (double)tt[1]; o[2] = (double)tt[2]; o[3] = (double)tt[3]; o[4] = (double)(S); o[5] = 0.0; } } while (0)
#else
#define WS_T(var) (void)0
#define WS_TIMING_DECL (void)0
#define WS_MARK(nn, nslots) (void)0
#define WS_MARK_HERE() (void)0
#define WS_TIMING_ACC(t0, t1, t2, t3) (void)0
#define WS_TIMING_OUT(S) (void)0
#endif

// Start of every role: the weight images staged by the kernel's prologue occupy the tile area.  The GEMM waves read their
// fragments into registers first; then the whole workgroup clears the tiles (an element that has not arrived yet reads as zero
// activations / zero cotangents).
__device__ __forceinline__ void ws_clear_tiles(unsigned short* lds16) {
    __syncthreads();
    for (int i = threadIdx.x; i < WS_LDS_USHORTS / 8; i += blockDim.x) reinterpret_cast<u32x4*>(lds16)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
}

// ============================================================================================================ waves Ca, Cb
// dW operands of one layer and tile-node: [32-slot block][piece] of delta_{l+1}^T (A) and a_l (B), two transposing reads each
struct WsOps { u32x4 A[2][NPB], B[2][NPB]; };
template <int I>
__device__ __forceinline__ void ws_load_op(WsOps& o, const unsigned short* Dt, const unsigned short* At) {
    constexpr int isB = I / 4, tau = (I % 4) / 2, piece = I % 2;          // I = 0..7
    const unsigned short* src = (isB ? At : Dt) + piece * 16 * TRS + 32 * tau;
    const u32x2 x = ws_tr_read(src);
    const u32x2 y = ws_tr_read(src + 4 * TRS);
    if constexpr (isB) o.B[tau][piece] = u32x4{x[0], x[1], y[0], y[1]};
    else o.A[tau][piece] = u32x4{x[0], x[1], y[0], y[1]};
}
// the 12 matrix instructions of one layer: cross terms outermost (an accumulator is touched every fourth instruction)
template <int IDX>
__device__ __forceinline__ void ws_dw_mfma(ws_f32x16 (&dW)[2][2], const WsOps& o) {
    constexpr int term = IDX / 4, to = (IDX % 4) / 2, ti = IDX % 2;
    constexpr int pa = term == 2 ? 1 : 0, pb = term == 1 ? 1 : 0;
    dW[to][ti] = ws_mfma32(o.A[to][pa], o.B[ti][pb], dW[to][ti]);
}
// this workgroup's d_theta slice: rows / columns of the dW tiles run over register slots (slot_feature); 32 x 32 result
// layout: column lane & 31, register v = 4 i + r <-> row 8 i + 4 (lane >> 5) + r
__device__ __forceinline__ void ws_write_dw(const BwdArgs& a, float* part, int l, const ws_f32x16 (&dW)[2][2], int lane, bool accumulate) {
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
                    if (idx >= 0) part[idx] = (accumulate ? part[idx] : 0.f) + dW[to][ti][v];
                }
            }
}

template <int NRL, bool FRONT>
__device__ __forceinline__ void ws_role_Ca(const BwdBf16Args& args, unsigned short* lds16, int S, const WsShape sh, float* part) {
    constexpr int NLIVE = NRL > 0 ? NRL : 4 * BT;
    constexpr int NPAIR = (NLIVE + 1) / 2;
    const BwdArgs& a = args.b;
    const MlpDev& m = a.m;
    const int lane = threadIdx.x & 63, g = lane >> 4, p = lane & 15;
    const int H1 = m.width[1], E = a.E, d = a.d;
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    const int own = p * TRS + g * 16;
    // transposing reads for a 32 x 32 x 16 operand: lane group g = (hf = g >> 1, column half g & 1) reads the 4 x 16 blocks of
    // points 8 hf + 4 half + (0..3) x slots 32 tau + 16 (g & 1) + (0..15), half = 0, 1  (+ piece * 16 * TRS + 32 * tau + half * 4 * TRS)
    const int trb = (8 * (g >> 1) + (p >> 2)) * TRS + 16 * (g & 1) + 4 * (p & 3);
    const int nit = sh.nit;
    ws_clear_tiles(lds16);
    float wout[BT][4];                                  // (delta_4 is formed here: see the step loop)
#pragma unroll
    for (int t = 0; t < BT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = feat_of(t, r, g);
            wout[t][r] = f < m.width[4] ? m.W[4][f] : 0.f;       // (no cotangent through the constant feature's slot)
        }

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
#pragma unroll
    for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int v = 0; v < 16; ++v) dW[to][ti][v] = 0.f;

    WsCursor cu{0, 0};
    float xv = 0.f, x0v = 0.f, dxv = 0.f;
    f32x4 c[BT];
#pragma unroll
    for (int t = 0; t < BT; ++t) c[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // FRONT (middle stage of the three-stage backward, cc_backward_front.hip): "layer 1" is the front kernel's z_2 out of HBM,
    // [tile][node][register][lane].  z0 = z_2 of node 0 of the current tile (the tangent element -- d z_2 / d t at node 0 -- needs
    // its signs).  The elements alternate between two register sets, the step loop is unrolled by two, and the set an element has
    // just been read from is at once the target of the element two steps on -- loads in flight across the step barrier; see
    // ws16_role_Ca (cc_bwd_ws16_kernel.h) for the measurement behind this.
    const int nl2 = NRL > 0 ? NRL : args.nl2;
    float zs[2][BT][4], z0[BT][4];
#pragma unroll
    for (int t = 0; t < BT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { zs[0][t][r] = 0.f; zs[1][t][r] = 0.f; z0[t][r] = 0.f; }
    auto fetch_z = [&](const WsCursor& c2, float (&dst)[BT][4]) __attribute__((always_inline)) {
        const bool there = c2.j < nit;                                      // (past the last element: the start of the buffer)
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
    // Opening a tile: x, x0 and the tile's 16 x E embedding values from HBM, c = b_1 + W_1[:, 1:] h on the fp32 matrix pipe.  The
    // whole workgroup waits at the step barrier while this wave does that, so everything loaded at the opening is issued in ONE
    // batch of eight K-steps before the first product (item_embedding_gemm): one memory latency instead of eight, 12.86 -> 12.63 ms
    // per launch.  (Requesting the HBM part a step ahead costs ten registers this wave does not have: EXPERIMENTS.md.)
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

    WS_TIMING_DECL;
    int rA3 = ws_ring0<WS_NS3, WS_TILE>(8), rD4 = ws_ring0<2, WS_TILE>(8);
    int rS4 = ws_ring0<2, WS_P3>(7), rD4w = ws_ring0<2, WS_TILE>(7);
    int rO1 = ws_ring0<WS_NS1, WS_TILE>(0), rO3 = ws_ring0<2, WS_P3>(0);
    // node position of the current element (the table value of the NEXT element is fetched a step ahead)
    bool live = cu.j < nit;
    bool is_tan = live && ws_is_tan(sh, cu);
    float tk = 0.f;
    if constexpr (!FRONT) {
        const int k = ws_node(sh, cu);
        const float uu = a.ccs[k] + 1.f;
        tk = (k == 0) ? xv : __fadd_rn(x0v, __fmul_rn(dxv, uu) * 0.5f);
    }
    WsOps ops;
    {
        const unsigned short* A3n = lds16 + WS_OFF_A3 + rA3 + trb;
        ws_load_op<4>(ops, A3n, A3n); ws_load_op<6>(ops, A3n, A3n); ws_load_op<5>(ops, A3n, A3n); ws_load_op<7>(ops, A3n, A3n);
    }
    auto step = [&](auto parc) __attribute__((always_inline)) {
        constexpr int PAR = decltype(parc)::value;      // (FRONT) the register set of this step's element
        float (&zc)[BT][4] = zs[PAR];
        WS_T(t0);
        const WsCursor nx = live ? ws_next(sh, cu) : cu;
        const int kn = ws_node(sh, nx);
        float ccs_n = 0.f;
        if constexpr (!FRONT) ccs_n = a.ccs[kn];
        if constexpr (FRONT) {
            if (live && cu.e == 0) {
#pragma unroll
                for (int t = 0; t < BT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) z0[t][r] = zc[t][r];
            }
        }
        const unsigned short* A3 = lds16 + WS_OFF_A3 + rA3 + trb;
        const unsigned short* D4 = lds16 + WS_OFF_D + 4 * WS_TILE + rD4 + trb;
        unsigned short* const O1 = lds16 + WS_OFF_A1 + rO1 + own;
        unsigned short* const O1p3 = lds16 + WS_OFF_P3 + rO3 + own;
        // operands of dW_3: the a_3 half came in before the barrier, the delta_4 half (written last step) here
        ws_load_op<0>(ops, D4, A3); ws_load_op<2>(ops, D4, A3); ws_load_op<1>(ops, D4, A3); ws_load_op<3>(ops, D4, A3);
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
        WS_T(t1);
        // (only these sixteen registers differ for a tangent element: the branch stays outside the matrix loop)
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
            ws_dw_mfma<nn>(dW, ops);
            // the split of a_1: two pairs per slot, stage by stage; then the stores
            if constexpr (nn < 4) { pairF(std::integral_constant<int, 2 * nn>{}, std::integral_constant<int, 0>{}); pairF(std::integral_constant<int, 2 * nn + 1>{}, std::integral_constant<int, 0>{}); }
            if constexpr (nn >= 1 && nn < 5) { pairF(std::integral_constant<int, 2 * (nn - 1)>{}, std::integral_constant<int, 1>{}); pairF(std::integral_constant<int, 2 * (nn - 1) + 1>{}, std::integral_constant<int, 1>{}); }
            if constexpr (nn >= 2 && nn < 6) { pairF(std::integral_constant<int, 2 * (nn - 2)>{}, std::integral_constant<int, 2>{}); pairF(std::integral_constant<int, 2 * (nn - 2) + 1>{}, std::integral_constant<int, 2>{}); }
            if constexpr (nn >= 5 && nn < 8) store_a(std::integral_constant<int, 0>{}, std::integral_constant<int, nn - 5>{});
            if constexpr (nn >= 8 && nn < 11) store_a(std::integral_constant<int, 1>{}, std::integral_constant<int, nn - 8>{});
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
        ws_adv<WS_NS3, WS_TILE>(rA3); ws_adv<2, WS_TILE>(rD4);
        ws_adv<WS_NS1, WS_TILE>(rO1); ws_adv<2, WS_P3>(rO3);
        {   // next step's a_3 operand of dW_3 (a tile written four steps ago)
            const unsigned short* A3n = lds16 + WS_OFF_A3 + rA3 + trb;
            ws_load_op<4>(ops, A3n, A3n); ws_load_op<6>(ops, A3n, A3n); ws_load_op<5>(ops, A3n, A3n); ws_load_op<7>(ops, A3n, A3n);
        }
        ws_adv<2, WS_P3>(rS4); ws_adv<2, WS_TILE>(rD4w);
        WS_T(t2);
        WS_STEP_SYNC();
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
    ws_write_dw(a, part, 3, dW, lane, FRONT && args.accumulate);
}

template <int NRL, bool FRONT>
__device__ __forceinline__ void ws_role_Cb(const BwdBf16Args& args, unsigned short* lds16, int S, const WsShape sh, float* part) {
    constexpr int NLIVE = NRL > 0 ? NRL : 4 * BT;
    constexpr int L = 4;
    const BwdArgs& a = args.b;
    const MlpDev& m = a.m;
    const int lane = threadIdx.x & 63, g = lane >> 4, p = lane & 15;
    const int HL = m.width[L];
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    const int own = p * TRS + g * 16;
    const int trb = (8 * (g >> 1) + (p >> 2)) * TRS + 16 * (g & 1) + 4 * (p & 3);
    ws_clear_tiles(lds16);
    float wout[BT][4];
#pragma unroll
    for (int t = 0; t < BT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = feat_of(t, r, g);
            wout[t][r] = f < HL ? m.W[L][f] : 0.f;       // (no cotangent through the constant feature's slot)
        }
    ws_f32x16 dW2[2][2], dW1[2][2];
#pragma unroll
    for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int v = 0; v < 16; ++v) { dW2[to][ti][v] = 0.f; dW1[to][ti][v] = 0.f; }
    unsigned q4[8][NPB];
    float rb0[8], rb1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        rb0[j] = rb1[j] = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < NPB; ++k2) q4[j][k2] = 0u;
    }
    WS_TIMING_DECL;
    int rA2 = ws_ring0<WS_NS2, WS_TILE>(9), rA1 = ws_ring0<WS_NS1, WS_TILE>(10);
    int rD3 = ws_ring0<2, WS_TILE>(9), rD2 = ws_ring0<2, WS_TILE>(10);
    int rS4 = ws_ring0<2, WS_P3>(7), rD4 = ws_ring0<2, WS_TILE>(7);
    int rA3c = ws_ring0<WS_NS3, WS_TILE>(8), rD4c = ws_ring0<2, WS_TILE>(8);
    WsOps o2, o1;
    {   // (first step: the tiles are still zero)
        const unsigned short* A2n = lds16 + WS_OFF_A2 + rA2 + trb;
        const unsigned short* A1n = lds16 + WS_OFF_A1 + rA1 + trb;
        ws_load_op<4>(o2, A2n, A2n); ws_load_op<6>(o2, A2n, A2n); ws_load_op<5>(o2, A2n, A2n); ws_load_op<7>(o2, A2n, A2n);
        ws_load_op<4>(o1, A1n, A1n); ws_load_op<6>(o1, A1n, A1n); ws_load_op<5>(o1, A1n, A1n); ws_load_op<7>(o1, A1n, A1n);
    }
    ws_f32x16 dW3[2][2];
#pragma unroll
    for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int v = 0; v < 16; ++v) dW3[to][ti][v] = 0.f;
    for (int s = 0; s < S; ++s) {
        WS_T(t0);
        // delta_4 of element s - 7 from what F3 left a step ago: delta_L = dout w_out act'(a_L), split, stored for B3 and Ca --
        // as micro-operations behind the matrix instructions below (independent of them)
        const unsigned short* S4 = lds16 + WS_OFF_S4 + rS4;
        unsigned short* const D4o = lds16 + WS_OFF_D + 4 * WS_TILE + rD4 + own;
        u32x4 sg4[BKS];
        float dout = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < BKS; ++s2) sg4[s2] = *reinterpret_cast<const u32x4*>(S4 + own + s2 * 8);
        dout = *reinterpret_cast<const float*>(S4 + p * TRS + 64);
        const unsigned short* A2 = lds16 + WS_OFF_A2 + rA2 + trb;
        const unsigned short* A1 = lds16 + WS_OFF_A1 + rA1 + trb;
        const unsigned short* D3 = lds16 + WS_OFF_D + 2 * WS_TILE + rD3 + trb;
        const unsigned short* D2 = lds16 + WS_OFF_D + 0 * WS_TILE + rD2 + trb;
        // (the a_2 / a_1 halves of the operands came in before the barrier: only the cotangent halves, written last step, are fetched here)
        ws_load_op<0>(o2, D3, A2); ws_load_op<2>(o2, D3, A2); ws_load_op<1>(o2, D3, A2); ws_load_op<3>(o2, D3, A2);
        float d4[BT][4];
        auto d4_reg = [&](auto ec) __attribute__((always_inline)) {
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4;
            if constexpr (e < NLIVE) d4[t][r] = (dout * wout[t][r]) * act_grad_q(sg4, t, r, slope);
            else d4[t][r] = 0.f;
        };
        auto pair4 = [&](auto jc, auto stc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, st = decltype(stc)::value, t = j / 2, r = 2 * (j & 1);
            if constexpr (2 * j < NLIVE) {
                if constexpr (st == 0) { rb0[j] = d4[t][r]; rb1[j] = d4[t][r + 1]; q4[j][0] = split_stage(rb0[j], rb1[j]); }
                if constexpr (st == 1) q4[j][1] = split_last(rb0[j], rb1[j]);
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
            if constexpr (nn < 12) ws_dw_mfma<nn>(dW2, o2);
            else ws_dw_mfma<nn - 12>(dW1, o1);
            if constexpr (nn < 4) ws_load_op<nn>(o1, D2, A1);
            // delta_4: two registers per slot, pair j split at slots j + 1 / j + 2, K-steps stored at 6, 7 / 10, 11
            // one register per slot, pair j split at slots 2j + 2 / 2j + 3, K-steps stored at 10, 11 / 18, 19 (spread over the 24 slots)
            if constexpr (nn < 16) d4_reg(std::integral_constant<int, nn>{});
            if constexpr (nn >= 2 && nn < 18 && (nn % 2) == 0) pair4(std::integral_constant<int, (nn - 2) / 2>{}, std::integral_constant<int, 0>{});
            if constexpr (nn >= 3 && nn < 19 && (nn % 2) == 1) pair4(std::integral_constant<int, (nn - 3) / 2>{}, std::integral_constant<int, 1>{});
            if constexpr (nn == 10 || nn == 11) store_d4(std::integral_constant<int, 0>{}, std::integral_constant<int, nn - 10>{});
            if constexpr (nn == 18 || nn == 19) store_d4(std::integral_constant<int, 1>{}, std::integral_constant<int, nn - 18>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        ws_adv<WS_NS2, WS_TILE>(rA2); ws_adv<WS_NS1, WS_TILE>(rA1);
        ws_adv<2, WS_TILE>(rD3); ws_adv<2, WS_TILE>(rD2);
        ws_adv<2, WS_P3>(rS4); ws_adv<2, WS_TILE>(rD4);
        ws_adv<WS_NS3, WS_TILE>(rA3c); ws_adv<2, WS_TILE>(rD4c);
        {   // next step's a_2 / a_1 operands (tiles written six and ten steps ago)
            const unsigned short* A2n = lds16 + WS_OFF_A2 + rA2 + trb;
            const unsigned short* A1n = lds16 + WS_OFF_A1 + rA1 + trb;
            ws_load_op<4>(o2, A2n, A2n); ws_load_op<6>(o2, A2n, A2n); ws_load_op<5>(o2, A2n, A2n); ws_load_op<7>(o2, A2n, A2n);
            ws_load_op<4>(o1, A1n, A1n); ws_load_op<6>(o1, A1n, A1n); ws_load_op<5>(o1, A1n, A1n); ws_load_op<7>(o1, A1n, A1n);
        }
        WS_T(t2);
        WS_STEP_SYNC();
        WS_T(t3);
        WS_TIMING_ACC(t0, t1, t2, t3);
    }
    WS_TIMING_OUT(S);
    ws_write_dw(a, part, 2, dW2, lane, FRONT && args.accumulate);
    ws_write_dw(a, part, 1, dW1, lane, FRONT && args.accumulate);
}

// ============================================================================================================ waves F1..F3
template <int NRL, int LAYER, bool FRONT>
__device__ __forceinline__ void ws_role_F(const BwdBf16Args& args, unsigned short* lds16, int S, const WsShape sh, float* part) {
    constexpr int NLIVE = NRL > 0 ? NRL : 4 * BT;
    constexpr int NPAIR = (NLIVE + 1) / 2;
    constexpr int L = 4;
    constexpr int DF = 2 * LAYER - 1;                  // the GEMM works on element s - DF, its vector work runs one step later
    constexpr int DP = 2 * LAYER;
    constexpr bool IS_OUT = LAYER == 3;
    constexpr int LO = LAYER < 3 ? LAYER + 1 : 3;      // layer of the activation tile this wave writes (F3 writes delta_4 instead)
    const BwdArgs& a = args.b;
    const MlpDev& m = a.m;
    const int lane = threadIdx.x & 63, g = lane >> 4, p = lane & 15;
    const int HL = m.width[L], n = a.n;
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    const int own = p * TRS + g * 16;
    const int nit = sh.nit;
    u32x4 Wf[BT][BKS][NPF];                             // W_l, three bf16 pieces: this wave's A operands for the whole launch
    {
        const unsigned short* imf = lds16 + (LAYER - 1) * WS_IMGF + lane * 8;
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int s2 = 0; s2 < BKS; ++s2)
#pragma unroll
                for (int k2 = 0; k2 < NPF; ++k2) Wf[t][s2][k2] = *reinterpret_cast<const u32x4*>(imf + ((t * BKS + s2) * NPF + k2) * FRAG);
    }
    ws_clear_tiles(lds16);

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
    float xvP = 0.f, x0vP = 0.f, gvP = 0.f, gfxvP = 0.f, cotbase = 0.f;
    float fxv = 0.f, fx0v = 0.f, dfdt = 0.f, fp0 = 0.f;
    auto new_item_P = [&]() __attribute__((always_inline)) {
        if constexpr (IS_OUT) {
            const long long q = (long long)(args.grp0 + ws_grp(cp)) * 16 + p;
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
    if (nit > 0) new_item_P();

    f32x4 acc[BT], acc1[BT];
    float actF[BT][4], delta[BT][4];                   // (delta: the tangent values of a tangent element)
#pragma unroll
    for (int t = 0; t < BT; ++t) {
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) { actF[t][r] = 0.f; delta[t][r] = 0.f; }
    }
    unsigned qF[8][NPF], q4[8][NPB];
    float rf0[8], rf1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        rf0[j] = rf1[j] = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < NPF; ++k2) qF[j][k2] = 0u;
#pragma unroll
        for (int k2 = 0; k2 < NPB; ++k2) q4[j][k2] = 0u;
    }
    float sd4[4] = {0.f, 0.f, 0.f, 0.f}, doutN = 0.f;
    float sc_a = 0.f, sc_sd = 0.f, sc_ex = 0.f, sc_s1 = 0.f, sc_f = 0.f, sc_fp = 0.f;
    const bool sig = m.out_act != UMNN_OUT_ELU_PLUS_ONE;

    WS_TIMING_DECL;
    int rAin = ws_ring0<ws_a_ns(LAYER), WS_TILE>(DF), rAin3 = ws_ring0<2, WS_P3>(DF);
    int rAout = ws_ring0<ws_a_ns(LO), WS_TILE>(DP), rAout3 = ws_ring0<2, WS_P3>(DP), rD4 = ws_ring0<2, WS_P3>(DP);
    float ccwP = 0.f;
    if constexpr (IS_OUT) ccwP = a.ccw[ws_node(sh, cp)];
    for (int s = 0; s < S; ++s) {
        WS_T(t0);
        const bool liveP = s >= DP && cp.j < nit;
        const bool tanP = liveP && ws_is_tan(sh, cp);
        const int kP = ws_node(sh, cp);
        const WsCursor nxP = liveP ? ws_next(sh, cp) : cp;
        float ccw_n = 0.f;
        if constexpr (IS_OUT) {
            ccw_n = a.ccw[ws_node(sh, nxP)];
            if (liveP && cp.e == 0) new_item_P();
        }
        const unsigned short* Ain = lds16 + ws_a_off(LAYER) + rAin + own;                  // a_l[s - DF]
        const unsigned short* Ain3 = lds16 + WS_OFF_P3 + (LAYER - 1) * 2 * WS_P3 + rAin3 + own;
        unsigned short* const Aout = lds16 + ws_a_off(LO) + rAout + own;                   // a_{l+1}[s - DP]
        unsigned short* const Aout3 = lds16 + WS_OFF_P3 + (LO - 1) * 2 * WS_P3 + rAout3 + own;
        unsigned short* const S4out = lds16 + WS_OFF_S4 + rD4;                             // F3: leading piece of a_4[s - 6], dout
        BFrag<NPF> bf;
#pragma unroll
        for (int s2 = 0; s2 < BKS; ++s2) {
#pragma unroll
            for (int k2 = 0; k2 < NPB; ++k2) bf.v[s2][k2] = *reinterpret_cast<const u32x4*>(Ain + k2 * 16 * TRS + s2 * 8);
            bf.v[s2][2] = *reinterpret_cast<const u32x4*>(Ain3 + s2 * 8);
        }

        // ---- micro-operations of the vector work of element s - DP
        auto act_reg = [&](auto ec, auto tanc) __attribute__((always_inline)) {
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
                    if constexpr (TAN) delta[t][r] = acc[t][r] * (actF[t][r] > 0.f ? 1.f : slope);
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
        // F3: output layer of element s - 6 (same expressions as cc_bwd_swp_kernel.h)
        // (four partial sums: no thirteen-deep dependent chain in front of the matrix loop)
        auto out_dot = [&](auto ec, auto tanc) __attribute__((always_inline)) {
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4;
            constexpr bool TAN = decltype(tanc)::value;
            if constexpr (e < 4) sd4[e] = 0.f;
            if constexpr (e < NLIVE) sd4[r] = fmaf(wout[t][r], TAN ? delta[t][r] : actF[t][r], sd4[r]);
        };
        // the scalar part in stages, one per matrix-instruction slot: a serial chain (cross-lane sum, exp, reciprocal, selects)
        // whose every link would otherwise stall the in-order issue of this wave, matrix instructions included
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
            if constexpr (st == 3) sc_ex = __expf(sig ? -sc_sd : sc_sd);
            if constexpr (st == 4) sc_s1 = __builtin_amdgcn_rcpf(1.f + sc_ex);   // (sigmoid outputs only: 1 ulp; ELU+1 does not use it)
            if constexpr (st == 5) {
                sc_f = sig ? sc_s1 : (sc_sd > 0.f ? sc_sd + 1.f : sc_ex);
                sc_fp = sig ? sc_s1 * (1.f - sc_s1) : (sc_sd > 0.f ? 1.f : sc_ex);
            }
            if constexpr (st == 6) {
                const bool node = liveP && !tanP;
                if (node && kP == 0) { fxv = sc_f; fp0 = sc_fp; }
                if (node && kP == n) fx0v = sc_f;
                // tangent element: the sum is w_out . d a_L / d t at node 0 -> d f / d t
                if (tanP) dfdt = fp0 * sc_sd;
            }
            if constexpr (st == 7) {
                const bool node = liveP && !tanP;
                const float rinv = -__builtin_amdgcn_rcpf(sc_f * sc_f);           // (inv_f launches only)
                const float invs = a.inv_f ? rinv : 1.f;
                const float cot = fmaf(cotbase * invs, ccwP, kP == 0 ? gfxvP : 0.f) * (node ? 1.f : 0.f);
                doutN = cot * sc_fp;                                               // (no cotangent flows back from a tangent element)
            }
        };
        auto out_reg = [&](auto ec) __attribute__((always_inline)) {       // d w_out += dout a_L  (delta_L is formed by wave Cb)
            constexpr int e = decltype(ec)::value, t = e / 4, r = e % 4;
            if constexpr (e < NLIVE) dwo[t][r] = fmaf(doutN, actF[t][r], dwo[t][r]);
        };
        auto pair4 = [&](auto jc) __attribute__((always_inline)) {          // leading bf16 piece of a_L: carries its sign
            constexpr int j = decltype(jc)::value, t = j / 2, r = 2 * (j & 1);
            if constexpr (j < NPAIR) q4[j][0] = split_last(actF[t][r], actF[t][r + 1]);
        };
        auto store_s4 = [&](auto sc) __attribute__((always_inline)) {
            constexpr int ks = decltype(sc)::value;
            *reinterpret_cast<u32x4*>(S4out + own + ks * 8) = u32x4{q4[4 * ks][0], q4[4 * ks + 1][0], q4[4 * ks + 2][0], q4[4 * ks + 3][0]};
        };
        WS_T(t1);
        // ---- the activations of element s - DP first (registers only: the operand fetches above are still in flight).  Only
        // these differ for a tangent element; the branch stays outside the matrix loop.
        if (tanP) {
            swp_static_for<16>([&](auto ec) { act_reg(ec, std::true_type{}); });
            if constexpr (IS_OUT) swp_static_for<16>([&](auto ec) { out_dot(ec, std::true_type{}); });
        } else {
            swp_static_for<16>([&](auto ec) { act_reg(ec, std::false_type{}); });
            if constexpr (IS_OUT) swp_static_for<16>([&](auto ec) { out_dot(ec, std::false_type{}); });
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- the GEMM of element s - DF (48 MFMAs, A operands = this wave's registers); behind it the rest of the vector work
        {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            swp_static_for<48>([&](auto nc) {
                constexpr int nn = decltype(nc)::value;
                WS_MARK(nn, 48);
                constexpr int s2 = nn / 24, idx = nn % 24;
                f32x4 (&ac)[BT] = acc;
                constexpr bool first = s2 == 0 && idx < 4;
                if constexpr (idx < 12) {
                    constexpr int t = idx % 4, ba = idx / 4;
                    ac[t] = mfma_bf16(Wf[t][s2][0], bf.v[s2][ba], first ? zero : ac[t]);
                } else if constexpr (idx < 20) {
                    constexpr int t = (idx - 12) % 4, ba = (idx - 12) / 4;
                    ac[t] = mfma_bf16(Wf[t][s2][1], bf.v[s2][ba], ac[t]);
                } else {
                    constexpr int t = idx - 20;
                    ac[t] = mfma_bf16(Wf[t][s2][2], bf.v[s2][0], ac[t]);
                }
                if constexpr (!IS_OUT) {
                    // the split of a_{l+1} spread over the whole loop (measured: front-loaded companions made the first twelve
                    // slots take 56 cycles each, the bare ones at the end 17): 30 micro-operations, one behind two of every three
                    // slots -- per K-step the four pairs stage by stage, then its three stores
                    constexpr int op = ws_op_of_slot(nn);
                    if constexpr (op >= 0 && op < 30) {
                        constexpr int ks = op / 15, w = op % 15;
                        if constexpr (w < 12) pairF(std::integral_constant<int, 4 * ks + w % 4>{}, std::integral_constant<int, w / 4>{});
                        else store_a(std::integral_constant<int, ks>{}, std::integral_constant<int, w - 12>{});
                    }
                } else {
                    // the leading piece of a_L does not wait for the scalar chain; dout and d w_out follow it
                    if constexpr (nn < 8) out_scalar(std::integral_constant<int, nn>{});
                    if constexpr (nn < 8) pair4(std::integral_constant<int, nn>{});
                    if constexpr (nn == 8 || nn == 9) store_s4(std::integral_constant<int, nn - 8>{});
                    if constexpr (nn == 9) *reinterpret_cast<float*>(S4out + p * TRS + 64) = doutN;
                    if constexpr (nn >= 10 && nn < 18) { out_reg(std::integral_constant<int, 2 * (nn - 10)>{}); out_reg(std::integral_constant<int, 2 * (nn - 10) + 1>{}); }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // ---- item boundary (outside the scheduled region: uniform branch)
        if constexpr (IS_OUT) {
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
        ws_adv<ws_a_ns(LAYER), WS_TILE>(rAin); ws_adv<2, WS_P3>(rAin3);
        ws_adv<ws_a_ns(LO), WS_TILE>(rAout); ws_adv<2, WS_P3>(rAout3); ws_adv<2, WS_P3>(rD4);
        WS_T(t2);
        WS_STEP_SYNC();
        WS_T(t3);
        WS_TIMING_ACC(t0, t1, t2, t3);
    }
    WS_TIMING_OUT(S);
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
                    if (idx >= 0) part[idx] = (FRONT && args.accumulate ? part[idx] : 0.f) + v;
                }
            }
    }
}

// ============================================================================================================ waves B1..B3
template <int NRL, int LAYER, bool FRONT>
__device__ __forceinline__ void ws_role_B(const BwdBf16Args& args, unsigned short* lds16, int S, const WsShape sh, float* part) {
    constexpr int NLIVE = NRL > 0 ? NRL : 4 * BT;
    constexpr int DB = 11 - LAYER;                     // element s - DB: W_l^T GEMM, then its vector work, in the same step
    constexpr bool IS_TAIL = LAYER == 1;
    const BwdArgs& a = args.b;
    const MlpDev& m = a.m;
    const int lane = threadIdx.x & 63, g = lane >> 4, p = lane & 15;
    const int H1 = m.width[1], E = a.E;
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    const int own = p * TRS + g * 16;
    const int nit = sh.nit;
    u32x4 WT[BT][BKS][NPB];                             // W_l^T, two bf16 pieces
    {
        const unsigned short* imt = lds16 + 3 * WS_IMGF + (LAYER - 1) * WS_IMGT + lane * 8;
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int s2 = 0; s2 < BKS; ++s2)
#pragma unroll
                for (int k2 = 0; k2 < NPB; ++k2) WT[t][s2][k2] = *reinterpret_cast<const u32x4*>(imt + ((t * BKS + s2) * NPB + k2) * FRAG);
    }
    ws_clear_tiles(lds16);

    f32x4 dW1x[BT], dcs[BT];
#pragma unroll
    for (int t = 0; t < BT; ++t) { dW1x[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dcs[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    WsCursor cb{0, 0};
    float xvB = 0.f, x0vB = 0.f, dxvB = 0.f;
    auto new_item_B = [&]() __attribute__((always_inline)) {
        if constexpr (IS_TAIL && !FRONT) {
            const long long q = (long long)(args.grp0 + ws_grp(cb)) * 16 + p;
            const long long qq = q < a.NI ? q : a.NI - 1;
            xvB = io_ld(a.x, qq, a.x_bf16);
            x0vB = a.x0 ? io_ld(a.x0, qq, a.x_bf16) : 0.f;
            dxvB = xvB - x0vB;
        }
    };
    if (nit > 0) new_item_B();

    WS_TIMING_DECL;
    int rAsg = ws_ring0<ws_a_ns(LAYER), WS_TILE>(DB), rDin = ws_ring0<2, WS_TILE>(DB), rDout = ws_ring0<2, WS_TILE>(DB);
    float tkB = 0.f;
    if constexpr (IS_TAIL && !FRONT) {
        const int kB = ws_node(sh, cb);
        const float uu = a.ccs[kB] + 1.f;
        tkB = (kB == 0) ? xvB : __fadd_rn(x0vB, __fmul_rn(dxvB, uu) * 0.5f);
    }
    for (int s = 0; s < S; ++s) {
        WS_T(t0);
        const bool liveB = s >= DB && cb.j < nit;
        WsCursor nxB = cb;
        float ccs_n = 0.f;
        int kBn = 0;
        if constexpr (IS_TAIL) {
            if (liveB) nxB = ws_next(sh, cb);
            kBn = ws_node(sh, nxB);
            if constexpr (!FRONT) ccs_n = a.ccs[kBn];
        }
        const unsigned short* Asg = lds16 + ws_a_off(LAYER) + rAsg + own;                                  // a_l[s - DB]
        const unsigned short* Din = lds16 + WS_OFF_D + (LAYER + 1 - 2) * 2 * WS_TILE + rDin + own;         // delta_{l+1}[s - DB]
        unsigned short* const Dout = lds16 + WS_OFF_D + (LAYER >= 2 ? LAYER - 2 : 0) * 2 * WS_TILE + rDout + own;   // delta_l[s - DB]
        BFrag<NPB> bd;
        u32x4 sg[BKS];
#pragma unroll
        for (int s2 = 0; s2 < BKS; ++s2)
#pragma unroll
            for (int k2 = 0; k2 < NPB; ++k2) bd.v[s2][k2] = *reinterpret_cast<const u32x4*>(Din + k2 * 16 * TRS + s2 * 8);
#pragma unroll
        for (int s2 = 0; s2 < BKS; ++s2) sg[s2] = *reinterpret_cast<const u32x4*>(Asg + s2 * 8);
        WS_T(t1);
        // ---- W_l^T GEMM (24 MFMAs, A operands = this wave's registers)
        f32x4 nd[BT];
        {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < BKS; ++s2) {
#pragma unroll
                for (int ba = 0; ba < NPB; ++ba)
#pragma unroll
                    for (int t = 0; t < BT; ++t) nd[t] = mfma_bf16(WT[t][s2][0], bd.v[s2][ba], (s2 == 0 && ba == 0) ? zero : nd[t]);
#pragma unroll
                for (int t = 0; t < BT; ++t) nd[t] = mfma_bf16(WT[t][s2][1], bd.v[s2][0], nd[t]);
            }
        }
        WS_MARK_HERE();                                   // (B waves: the cut is always after the GEMM)
        // ---- delta_l = (W_l^T delta_{l+1}) . act'(a_l); B1: the tail of the node; B2, B3: split and store for the wave below
        f32x4 dl[BT];
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (4 * t + r < NLIVE) {
                    dl[t][r] = nd[t][r] * act_grad_q(sg, t, r, slope);
                    if constexpr (IS_TAIL && !FRONT) {
                        dcs[t][r] += dl[t][r];
                        dW1x[t][r] = fmaf(dl[t][r], tkB, dW1x[t][r]);
                    }
                } else {
                    dl[t][r] = 0.f;
                }
            }
        if constexpr (!IS_TAIL) {
            BFrag<NPB> q;
            split_regs<NRL, NPB>(dl, q);
#pragma unroll
            for (int s2 = 0; s2 < BKS; ++s2)
#pragma unroll
                for (int k2 = 0; k2 < NPB; ++k2) *reinterpret_cast<u32x4*>(Dout + k2 * 16 * TRS + s2 * 8) = q.v[s2][k2];
        }
        if constexpr (IS_TAIL && FRONT) {
            // middle stage: delta_2 = dL/dz_2 of this node goes back to HBM for the front-backward kernel (not for the tangent element)
            if (liveB && !ws_is_tan(sh, cb)) {
                const int nl2 = NRL > 0 ? NRL : args.nl2;
                const size_t base = ((size_t)ws_grp(cb) * (size_t)(a.n + 1) + (size_t)ws_node(sh, cb)) * nl2 * 64 + lane;
#pragma unroll
                for (int t = 0; t < BT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * t + r < NLIVE && 4 * t + r < nl2) args.d2[base + (size_t)(4 * t + r) * 64] = dl[t][r];
            }
            if (liveB) cb = nxB;
        }
        if constexpr (IS_TAIL && !FRONT) {
            if (liveB && cb.e == sh.ne - 1) {
                const long long q = (long long)(args.grp0 + ws_grp(cb)) * 16 + p;
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
            if (liveB) {
                const bool crossed = nxB.j != cb.j;
                cb = nxB;
                if (crossed && cb.j < nit) new_item_B();
                const float uu = ccs_n + 1.f;
                tkB = (kBn == 0) ? xvB : __fadd_rn(x0vB, __fmul_rn(dxvB, uu) * 0.5f);
            }
        }
        ws_adv<ws_a_ns(LAYER), WS_TILE>(rAsg); ws_adv<2, WS_TILE>(rDin); ws_adv<2, WS_TILE>(rDout);
        WS_T(t2);
        WS_STEP_SYNC();
        WS_T(t3);
        WS_TIMING_ACC(t0, t1, t2, t3);
    }
    WS_TIMING_OUT(S);
    if constexpr (IS_TAIL && !FRONT) {
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = dW1x[t][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
                const int f = feat_of(t, r, g);
                if (p == 0 && f < H1) part[a.poffW[0] + f * (1 + E)] = v;
            }
    }
}

// wave -> role.  Waves w and w + 4 share a SIMD.  Default pairing: Ca + Cb, F1 + B1, F2 + B2, F3 + B3 -- every SIMD gets the same
// matrix-pipe time (1152 cycles per step); the other pairings measured (EXPERIMENTS.md) balance instruction counts better and lose 1-2 ms to
// matrix-pipe contention.
template <int NRL, bool FRONT = false>
__global__ __launch_bounds__(64 * WS_WAVES, 1) void cc_bwd_ws_kernel(const BwdBf16Args args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned short* lds16 = reinterpret_cast<unsigned short*>(lds);
    const BwdArgs& a = args.b;
    const MlpDev& m = a.m;
    const int tid = threadIdx.x;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (args.only_if && *args.only_if == 0) return;      // queued as the fallback of the fp16-piece pipeline: nothing overflowed
    // ---- the weights: staged once as fragment images (the other kernels' staging code), then read into the owners' registers
    for (int l = 1; l <= 3; ++l) {
        stage_frag_image<false, NPF>(m, l, lds16 + (l - 1) * WS_IMGF, tid, blockDim.x);
        stage_frag_image<true, NPB>(m, l, lds16 + 3 * WS_IMGF + (l - 1) * WS_IMGT, tid, blockDim.x);
    }
    __syncthreads();
    const int role = wid & 3, upper = wid >> 2;
    WsShape sh;
    sh.tan0 = a.gfx != nullptr ? 1 : 0;
    sh.ne = a.n + 1 + sh.tan0;
    sh.nit = blockIdx.x < a.ngroups ? (int)((a.ngroups - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
    const int S = sh.nit * sh.ne + WS_DEPTH;
    // one d_theta slice per workgroup: every parameter is owned by exactly one of its waves.  One-pass launches use slices
    // 0 .. gridDim.x - 1 (the reduction then reads a quarter of what the four-slices-per-workgroup kernels make it read); as the
    // middle stage of the three-stage backward the slice is the first of the workgroup's four (the front kernels use all four)
    float* part = a.partials + (size_t)blockIdx.x * (FRONT ? UMNN_WAVES_PER_BLOCK : 1) * a.n_params;
    if (!upper) {
        if (role == 0) ws_role_Ca<NRL, FRONT>(args, lds16, S, sh, part);
        else if (role == 1) ws_role_F<NRL, 1, FRONT>(args, lds16, S, sh, part);
        else if (role == 2) ws_role_F<NRL, 2, FRONT>(args, lds16, S, sh, part);
        else ws_role_F<NRL, 3, FRONT>(args, lds16, S, sh, part);
    } else {
        if (role == 0) ws_role_Cb<NRL, FRONT>(args, lds16, S, sh, part);
        else if (role == 1) ws_role_B<NRL, 1, FRONT>(args, lds16, S, sh, part);
        else if (role == 2) ws_role_B<NRL, 2, FRONT>(args, lds16, S, sh, part);
        else ws_role_B<NRL, 3, FRONT>(args, lds16, S, sh, part);
    }
}
