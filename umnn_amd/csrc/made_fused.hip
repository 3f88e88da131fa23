// The MADE conditioner of one flow block as ONE launch (the optional half; reference
// models/UMNN/made.py:16-27 MaskedLinear, :113-119 MADE.forward, :165-168 ConditionnalMADE.forward, called once per block by
// UMNNMAF.py:79):   h = W_L relu( ... relu(W_1 x + b_1) ... ) + b_L   with the masks folded into the weights by the host.
//
// Why: at the launch-bound shapes (2-D toy flow, POWER d = 6, the VAE's prior flow) a step was 60-80 % quadrature and the rest
// two launches per masked linear (operand split + hipBLASLt GEMM): 6-10 short launches per block.  Here the whole masked MLP
// runs in one kernel; the large shapes (BSDS300: 8192 x 512 x 1890 output layer) keep the library GEMMs.
//
// Arithmetic = the inference fast path's: x W^T ~= xh Wh + xl Wh + xh Wl with bf16 pieces and fp32 accumulation on
// v_mfma_f32_16x16x32_bf16 (3e-6 of the output range; masked entries are exact zeros in both pieces, so the autoregressive
// property holds bit for bit); the bias is added in fp32.
//
// Layout.  D[out feature][row] = A[out feature][k] B[k][row]: a workgroup (4 waves) owns RT tiles of 16 rows; wave w owns the
// output tiles t = w, w+4, ... of every layer, eight at a time, for ALL the workgroup's rows -- a weight fragment fetched
// from L2 is used for RT row tiles x 3 cross terms.  The activations of the current layer live in LDS as ready-made B operands
// [row tile][K-step][piece][lane][8 bf16]: after MFMA a lane (g, p) holds features 16t + 4g + r of row p, so with the K order
//       k-slot (s, g, j)  <->  feature 32 s + 16 (j >> 2) + 4 g + (j & 3)
// every lane packs the next layer's operand out of its own accumulators (no cross-lane movement); the host packs the weight
// fragments [tile][K-step][piece][lane][8 bf16] in that K order (umnn_amd/made.py: pack_fragments).  Two barriers per layer.
#include <hip/hip_runtime.h>
#include "cc_bf16.h"
#include "cc_host.h"
#include "../../include/umnn_cc.h"

namespace {
constexpr int MF_WAVES = 4;
constexpr int MF_TPC = 8;              // output tiles per wave and pass
constexpr int MF_SMAX = 16;            // K-steps of 32: widths up to 512
constexpr int MF_FRAG = 512;           // ushorts per fragment (64 lanes x 8)

struct MadeArgs {
    const unsigned short* W[UMNN_MADE_MAX_LAYERS];     // packed fragments of layer l
    const float* b[UMNN_MADE_MAX_LAYERS];
    int width[UMNN_MADE_MAX_LAYERS + 1];               // [K0, N1, ..., NL]
    int n_layers;
    const float* x;                                     // [B, K0]
    void* out;                                          // [B, NL] fp32 or bf16
    int out_bf16;                                       // 0 fp32 h, 1 bf16 h, 2 split3 operand of relu(h) for a following library GEMM
    int out_ld;                                         // (mode 2) bf16 elements per operand row
    int relu_in;                                        // (made_linear_kernel) x is a pre-activation: ReLU in the operand load
    int x_vec;                                          // (made_linear_kernel) K % 4 == 0 and x 16-byte aligned: b128 loads
    const float* x2;                                    // (made_linear_kernel) second input block [B, K - K1] or null
    int K1;
    long long B;
};

__device__ __forceinline__ int kfeat(int s, int g, int j) { return 32 * s + 16 * (j >> 2) + 4 * g + (j & 3); }

template <int RT>
__global__ __launch_bounds__(64 * MF_WAVES, 1) void made_fused_kernel(const MadeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short act[];       // [RT][MF_SMAX][2][64][8]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, p = lane & 15;
    const long long row0 = (long long)blockIdx.x * (16 * RT);
    auto act_at = [&](int rt, int s, int piece) { return act + (((rt * MF_SMAX + s) * 2 + piece) * 64 + lane) * 8; };

    // ---- layer-0 operands straight from x: every wave stages its share of the (row tile, K-step) pairs
    {
        const int K0 = a.width[0], S0 = (K0 + 31) / 32;
        for (int it = wid; it < RT * S0; it += MF_WAVES) {
            const int rt = it / S0, s = it - rt * S0;
            const long long row = row0 + 16 * rt + p;
            const float* xr = a.x + (row < a.B ? row : a.B - 1) * K0;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = kfeat(s, g, j);
                v[j] = (k < K0 && row < a.B) ? xr[k] : 0.f;
            }
            unsigned q0[2], q1[2], q2[2], q3[2];
            split_pair<2>(v[0], v[1], q0); split_pair<2>(v[2], v[3], q1); split_pair<2>(v[4], v[5], q2); split_pair<2>(v[6], v[7], q3);
#pragma unroll
            for (int piece = 0; piece < 2; ++piece)
                *reinterpret_cast<u32x4*>(act_at(rt, s, piece)) = u32x4{q0[piece], q1[piece], q2[piece], q3[piece]};
        }
    }
    __syncthreads();

    for (int l = 0; l < a.n_layers; ++l) {
        const int K = a.width[l], N = a.width[l + 1];
        const int S = (K + 31) / 32, T = (N + 15) / 16;
        const bool last = l + 1 == a.n_layers;
        const unsigned short* __restrict__ Wl = a.W[l];
        const float* __restrict__ bl = a.b[l];
        // hidden layers have at most 4 * MF_TPC tiles (one pass per wave); the output layer may take several passes, whose
        // results go to global memory, so no barrier is needed between them
        const int npass = (T + MF_WAVES * MF_TPC - 1) / (MF_WAVES * MF_TPC);     // (uniform: every wave runs every pass and barrier)
        for (int pass = 0; pass < npass; ++pass) {
            const int t0 = wid + pass * MF_WAVES * MF_TPC;
            const int nc = t0 < T ? (T - t0 + MF_WAVES - 1) / MF_WAVES : 0;      // valid tiles of this wave in this pass (<= MF_TPC)
            f32x4 acc[MF_TPC][RT];
#pragma unroll
            for (int c = 0; c < MF_TPC; ++c)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[c][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
            // fragments of K-step s for this wave's tiles t0 + 4c: [tile][K-step][piece]; one K-step is fetched ahead
            u32x4 fr[2][MF_TPC][2];
            auto fetch = [&](int buf, int s) __attribute__((always_inline)) {
#pragma unroll
                for (int c = 0; c < MF_TPC; ++c) {
                    if (c < nc) {
                        const int t = t0 + MF_WAVES * c;
#pragma unroll
                        for (int piece = 0; piece < 2; ++piece)
                            fr[buf][c][piece] = *reinterpret_cast<const u32x4*>(Wl + ((size_t)(t * S + s) * 2 + piece) * MF_FRAG + lane * 8);
                    }
                }
            };
            if (nc > 0) fetch(0, 0);
            if (nc > 0)
            for (int s = 0; s < S; s += 2) {
                // two K-steps per trip so that the fragment buffers are statically indexed
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int ss = s + half;
                    if (ss < S) {
                        if (ss + 1 < S) fetch(half ^ 1, ss + 1);
                        u32x4 bh[RT], bo[RT];
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            bh[rt] = *reinterpret_cast<const u32x4*>(act_at(rt, ss, 0));
                            bo[rt] = *reinterpret_cast<const u32x4*>(act_at(rt, ss, 1));
                        }
                        // (term outermost: consecutive MFMAs write different accumulators)
#pragma unroll
                        for (int term = 0; term < 3; ++term)
#pragma unroll
                            for (int c = 0; c < MF_TPC; ++c)
                                if (c < nc) {
#pragma unroll
                                    for (int rt = 0; rt < RT; ++rt)
                                        acc[c][rt] = mfma_bf16(fr[half][c][term == 2 ? 1 : 0], term == 1 ? bo[rt] : bh[rt], acc[c][rt]);   // Wh xh, Wh xl, Wl xh
                                }
                    }
                }
            }
            if (!last) __syncthreads();          // every wave has read the current activations (hidden layers: one pass each)
            // ---- epilogue of the pass: bias (+ ReLU, split, next layer's operand) or the store of h
#pragma unroll
            for (int c = 0; c < MF_TPC; ++c) {
                if (c >= nc) continue;
                const int t = t0 + MF_WAVES * c;
                const int f0 = 16 * t + 4 * g;                        // this lane's four features
                float bv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[r] = f0 + r < N ? bl[f0 + r] : 0.f;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc[c][rt][r] + bv[r];
                    if (!last) {
                        // features 16t + 4g + r = k-slots j = 4 (t & 1) + r of K-step t >> 1: half of this lane's operand quad
                        unsigned q0[2], q1[2];
                        split_pair<2>(f0 + 0 < N ? fmaxf(v[0], 0.f) : 0.f, f0 + 1 < N ? fmaxf(v[1], 0.f) : 0.f, q0);
                        split_pair<2>(f0 + 2 < N ? fmaxf(v[2], 0.f) : 0.f, f0 + 3 < N ? fmaxf(v[3], 0.f) : 0.f, q1);
#pragma unroll
                        for (int piece = 0; piece < 2; ++piece)
                            *reinterpret_cast<u32x2*>(act_at(rt, t >> 1, piece) + 4 * (t & 1)) = u32x2{q0[piece], q1[piece]};
                    } else {
                        const long long row = row0 + 16 * rt + p;
                        if (row < a.B) {
                            if (a.out_bf16 == 2) {
                                // the next layer is a LIBRARY GEMM: write its left operand [hi | lo | hi | 1 | 1 | 0..] of
                                // a = relu(v) (umnn_made_split3's format, ld = a.out_ld bf16 per row)
                                unsigned short* o = reinterpret_cast<unsigned short*>(a.out) + row * a.out_ld;
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    if (f0 + r < N) {
                                        const float av = fmaxf(v[r], 0.f);
                                        const unsigned short hb = bf16_rn_bits(av);
                                        const unsigned short lb = bf16_rn_bits(av - bf16_bits_to_f32(hb));
                                        o[f0 + r] = hb; o[N + f0 + r] = lb; o[2 * N + f0 + r] = hb;
                                    }
                                if (t == 0 && g == 0) {
                                    o[3 * N] = 0x3f80; o[3 * N + 1] = 0x3f80;
                                    for (int k = 3 * N + 2; k < a.out_ld; ++k) o[k] = 0;
                                }
                            } else if (a.out_bf16) {
                                unsigned short* o = reinterpret_cast<unsigned short*>(a.out) + row * N + f0;
#pragma unroll
                                for (int r = 0; r < 4; ++r) if (f0 + r < N) o[r] = f32_to_bf16_rn(v[r]);
                            } else {
                                float* o = reinterpret_cast<float*>(a.out) + row * N + f0;
                                if (f0 + 3 < N && (N & 3) == 0) *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
                                else
#pragma unroll
                                    for (int r = 0; r < 4; ++r) if (f0 + r < N) o[r] = v[r];
                            }
                        }
                    }
                }
            }
        }
        if (!last) {
            // zero the operand halves of an odd tile count / of K-steps the next layer reads but this layer did not write
            const int Tn = T, Sn = (N + 31) / 32;
            if ((Tn & 1) && wid == 0) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int piece = 0; piece < 2; ++piece)
                        *reinterpret_cast<u32x2*>(act_at(rt, Sn - 1, piece) + 4) = u32x2{0u, 0u};
            }
            __syncthreads();                     // next layer's operands complete
        }
    }
}

// ONE masked linear layer per launch (umnn_made_linear_forward): the conditioners with a wide OUTPUT layer (the VAE prior flow's
// 1920, BSDS300's 1890 columns), where a grid over row groups alone cannot fill the chip.  Grid = (groups of 16 RT rows) x (G
// groups of output tiles: workgroup y owns tiles y, y + G, ...); same operand layout, fragments and arithmetic as the kernel above.
// A launch is a few microseconds of work, so it is organised around memory LATENCY: the first PD K-steps of the wave's weight
// fragments are requested before anything else, all of the workgroup's x rows are loaded in one batch (ReLU of the previous
// layer and the bf16 split applied on the way into LDS), and the K loop keeps PD K-steps of fragments in flight in a register ring.
template <int RT, int TPC, int PD, int NW>
__global__ __launch_bounds__(64 * NW, 1) void made_linear_kernel(const MadeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short act[];       // [RT][MF_SMAX][2][64][8]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, p = lane & 15;
    const long long row0 = (long long)blockIdx.x * (16 * RT);
    auto act_at = [&](int rt, int s, int piece) { return act + (((rt * MF_SMAX + s) * 2 + piece) * 64 + lane) * 8; };
    const int K = a.width[0], N = a.width[1], K1 = a.x2 ? a.K1 : a.width[0];
    const int S = (K + 31) / 32, T = (N + 15) / 16;
    const unsigned short* __restrict__ Wl = a.W[0];
    const float* __restrict__ bl = a.b[0];
    const int G = gridDim.y, fg = blockIdx.y;
    const int TG = fg < T ? (T - fg + G - 1) / G : 0;                           // this workgroup's tiles
    const int npass = (TG + NW * TPC - 1) / (NW * TPC);

    u32x4 fr[PD][TPC][2];                                                       // ring: K-step s lives in buffer s % PD
    auto fetch = [&](int buf, int s, int t0, int nc) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < TPC; ++c)
            if (c < nc) {
                const int t = fg + G * (t0 + NW * c);
#pragma unroll
                for (int piece = 0; piece < 2; ++piece)
                    fr[buf][c][piece] = *reinterpret_cast<const u32x4*>(Wl + ((size_t)(t * S + s) * 2 + piece) * MF_FRAG + lane * 8);
            }
    };
    auto tiles_of = [&](int pass, int& t0, int& nc) __attribute__((always_inline)) {
        t0 = wid + pass * NW * TPC;                                       // (index among the workgroup's TG tiles)
        nc = t0 < TG ? (TG - t0 + NW - 1) / NW : 0;
        nc = nc > TPC ? TPC : nc;
    };
    int t0, nc;
    tiles_of(0, t0, nc);
#pragma unroll
    for (int d = 0; d < PD; ++d)
        if (d < S) fetch(d, d, t0, nc);

    // ---- operands from x: every wave stages its share of the (row tile, K-step) pairs, all loads in flight together
    {
        constexpr int NST = RT * MF_SMAX / NW;
        const bool vec = a.x_vec != 0;
        float v[NST][8];
#pragma unroll
        for (int c = 0; c < NST; ++c) {
            const int it = wid + NW * c;
            const int rt = it / S, s = it - rt * S;
            const long long row = row0 + 16 * rt + p;
            const bool ok = it < RT * S && row < a.B;
            // x = [x1 | x2] (the conditional MADE's cat((context, x), 1), made.py:167): columns < K1 from x1, the rest from x2
            const long long rr = row < a.B ? row : a.B - 1;
            const float* xr1 = a.x + rr * K1;
            const float* xr2 = a.x2 ? a.x2 + rr * (K - K1) - K1 : xr1;
            if (vec) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = kfeat(s, g, 4 * h);
                    f32x4 q = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (ok && k < K) q = *reinterpret_cast<const f32x4*>((k < K1 ? xr1 : xr2) + k);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[c][4 * h + r] = q[r];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = kfeat(s, g, j);
                    v[c][j] = (ok && k < K) ? (k < K1 ? xr1 : xr2)[k] : 0.f;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NST; ++c) {
            const int it = wid + NW * c;
            if (it < RT * S) {
                const int rt = it / S, s = it - rt * S;
                if (a.relu_in) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[c][j] = fmaxf(v[c][j], 0.f);
                }
                unsigned q0[2], q1[2], q2[2], q3[2];
                split_pair<2>(v[c][0], v[c][1], q0); split_pair<2>(v[c][2], v[c][3], q1);
                split_pair<2>(v[c][4], v[c][5], q2); split_pair<2>(v[c][6], v[c][7], q3);
#pragma unroll
                for (int piece = 0; piece < 2; ++piece)
                    *reinterpret_cast<u32x4*>(act_at(rt, s, piece)) = u32x4{q0[piece], q1[piece], q2[piece], q3[piece]};
            }
        }
    }
    __syncthreads();

    // one accumulator per cross term when a wave owns a single tile (three independent MFMA chains per row tile instead of one
    // chain of 3 S dependent instructions); the bias enters through the accumulator init, fetched before the K loop
    constexpr int NT = TPC == 1 ? 3 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        f32x4 acc[NT][TPC][RT];
#pragma unroll
        for (int c = 0; c < TPC; ++c) {
            const int f0 = 16 * (fg + G * (t0 + NW * c)) + 4 * g;               // this lane's four features of tile c
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < nc) {
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[r] = f0 + r < N ? bl[f0 + r] : 0.f;
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                acc[0][c][rt] = bv;
#pragma unroll
                for (int ti = 1; ti < NT; ++ti) acc[ti][c][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        u32x4 bh[2][RT], bo[2][RT];                                             // B operands of K-step ss in buffer ss & 1, read one step ahead
        auto ldb = [&](int buf, int ss) __attribute__((always_inline)) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                bh[buf][rt] = *reinterpret_cast<const u32x4*>(act_at(rt, ss, 0));
                bo[buf][rt] = *reinterpret_cast<const u32x4*>(act_at(rt, ss, 1));
            }
        };
        if (nc > 0) {
            ldb(0, 0);
            for (int s = 0; s < S; s += PD) {
#pragma unroll
                for (int d = 0; d < PD; ++d) {
                    const int ss = s + d;
                    if (ss < S) {
                        if (ss + 1 < S) ldb((d + 1) & 1, ss + 1);
#pragma unroll
                        for (int term = 0; term < 3; ++term)
#pragma unroll
                            for (int c = 0; c < TPC; ++c)
                                if (c < nc) {
#pragma unroll
                                    for (int rt = 0; rt < RT; ++rt)
                                        acc[NT == 3 ? term : 0][c][rt] = mfma_bf16(fr[d][c][term == 2 ? 1 : 0], term == 1 ? bo[d & 1][rt] : bh[d & 1][rt],
                                                                                  acc[NT == 3 ? term : 0][c][rt]);   // Wh xh, Wh xl, Wl xh
                                }
                        if (ss + PD < S) fetch(d, ss + PD, t0, nc);          // the buffer is free again
                    }
                }
            }
        }
        const int t0e = t0, nce = nc;
        if (pass + 1 < npass) {                                                 // next pass: its first fragments behind this epilogue
            tiles_of(pass + 1, t0, nc);
#pragma unroll
            for (int d = 0; d < PD; ++d)
                if (d < S) fetch(d, d, t0, nc);
        }
#pragma unroll
        for (int c = 0; c < TPC; ++c) {
            if (c >= nce) continue;
            const int t = fg + G * (t0e + NW * c);
            const int f0 = 16 * t + 4 * g;                                      // this lane's four features
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const long long row = row0 + 16 * rt + p;
                if (row >= a.B) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[0][c][rt][r];
                    if constexpr (NT == 3) v[r] += acc[1][c][rt][r] + acc[2][c][rt][r];
                }
                if (a.out_bf16) {
                    unsigned short* o = reinterpret_cast<unsigned short*>(a.out) + row * N + f0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (f0 + r < N) o[r] = f32_to_bf16_rn(v[r]);
                } else {
                    float* o = reinterpret_cast<float*>(a.out) + row * N + f0;
                    if (f0 + 3 < N && (N & 3) == 0) *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
                    else
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (f0 + r < N) o[r] = v[r];
                }
            }
        }
    }
}
}  // namespace

extern "C" int umnn_made_mlp_forward(const umnn_made_net* net, const float* x, long long B, void* h_out, int out_bf16,
                                     void* stream_) {
    return umnn_made_mlp_forward_ex(net, x, B, h_out, out_bf16 != 0 ? 1 : 0, 0, stream_);
}

extern "C" int umnn_made_mlp_forward_ex(const umnn_made_net* net, const float* x, long long B, void* h_out, int out_mode,
                                        int out_ld, void* stream_) {
    if (!net) return umnn_fail(UMNN_EINVAL, "made_mlp: net is null");
    const int L = net->n_layers;
    if (L < 1 || L > UMNN_MADE_MAX_LAYERS) return umnn_fail(UMNN_EUNSUPPORTED, "made_mlp: 1..UMNN_MADE_MAX_LAYERS linear layers");
    if (B < 0) return umnn_fail(UMNN_EINVAL, "made_mlp: B < 0");
    MadeArgs a = {};
    for (int l = 0; l <= L; ++l) {
        a.width[l] = net->widths[l];
        if (a.width[l] < 1) return umnn_fail(UMNN_EINVAL, "made_mlp: widths must be >= 1");
        if (l < L && a.width[l] > 32 * MF_SMAX)
            return umnn_fail(UMNN_EUNSUPPORTED, "made_mlp: input / hidden widths up to 512 (wider conditioners keep the library GEMMs)");
    }
    if (B == 0) return 0;
    if (!x || !h_out) return umnn_fail(UMNN_EINVAL, "made_mlp: null pointer");
    for (int l = 0; l < L; ++l) {
        if (!net->W[l] || !net->b[l]) return umnn_fail(UMNN_EINVAL, "made_mlp: null weight / bias pointer");
        a.W[l] = reinterpret_cast<const unsigned short*>(net->W[l]);
        a.b[l] = net->b[l];
    }
    if (out_mode < 0 || out_mode > 2) return umnn_fail(UMNN_EINVAL, "made_mlp: out_mode is 0 (fp32), 1 (bf16) or 2 (split3 operand)");
    if (out_mode == 2 && (out_ld < 3 * net->widths[L] + 2 || net->widths[L] > 32 * MF_SMAX))
        return umnn_fail(UMNN_EINVAL, "made_mlp: operand output needs ld >= 3 * width + 2 and width <= 512");
    a.n_layers = L; a.x = x; a.out = h_out; a.out_bf16 = out_mode; a.out_ld = out_ld; a.B = B;
    hipStream_t stream = (hipStream_t)stream_;
    // row tiles per workgroup: 4 (a fragment serves 64 rows) once that still gives every CU a workgroup, else 1
    const int cus = umnn_num_cus();
    const int RT = (B + 63) / 64 >= cus / 2 ? 4 : 1;
    const size_t lds = (size_t)RT * MF_SMAX * 2 * 64 * 8 * sizeof(unsigned short);
    const unsigned grid = (unsigned)((B + 16 * RT - 1) / (16 * RT));
    if (RT == 4) {
        if (int rc = umnn_allow_lds((const void*)made_fused_kernel<4>, lds)) return rc;
        hipLaunchKernelGGL(made_fused_kernel<4>, dim3(grid), dim3(64 * MF_WAVES), lds, stream, a);
        umnn_note_made_launch("made_fused<RT=4>");
    } else {
        if (int rc = umnn_allow_lds((const void*)made_fused_kernel<1>, lds)) return rc;
        hipLaunchKernelGGL(made_fused_kernel<1>, dim3(grid), dim3(64 * MF_WAVES), lds, stream, a);
        umnn_note_made_launch("made_fused<RT=1>");
    }
    return umnn_check(hipGetLastError(), "made_fused launch");
}

extern "C" int umnn_made_linear_forward(const void* W_frag, const float* bias, int K, int N, const float* x, const float* x2, int K1,
                                        long long B, int relu_in, void* out, int out_bf16, int row_tiles, int feature_groups,
                                        void* stream_) {
    if (K < 1 || N < 1) return umnn_fail(UMNN_EINVAL, "made_linear: widths must be >= 1");
    if (K > 32 * MF_SMAX) return umnn_fail(UMNN_EUNSUPPORTED, "made_linear: input width up to 512 (wider layers keep the library GEMMs)");
    if (B < 0) return umnn_fail(UMNN_EINVAL, "made_linear: B < 0");
    if (x2 && (K1 < 1 || K1 >= K)) return umnn_fail(UMNN_EINVAL, "made_linear: two input blocks need 1 <= K1 < K");
    if (row_tiles != 0 && row_tiles != 1 && row_tiles != 2 && row_tiles != 4) return umnn_fail(UMNN_EINVAL, "made_linear: row_tiles is 0 (auto), 1, 2 or 4");
    if (feature_groups < 0 || feature_groups > 65535) return umnn_fail(UMNN_EINVAL, "made_linear: feature_groups is 0 (auto) .. 65535");
    if (B == 0) return 0;
    if (!W_frag || !bias || !x || !out) return umnn_fail(UMNN_EINVAL, "made_linear: null pointer");
    MadeArgs a = {};
    a.width[0] = K; a.width[1] = N; a.n_layers = 1;
    a.W[0] = reinterpret_cast<const unsigned short*>(W_frag); a.b[0] = bias;
    a.x = x; a.x2 = x2; a.K1 = x2 ? K1 : K; a.out = out; a.out_bf16 = out_bf16 ? 1 : 0; a.out_ld = 0; a.relu_in = relu_in ? 1 : 0; a.B = B;
    a.x_vec = (a.K1 % 4 == 0 && (K - a.K1) % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(x2) & 15) == 0) ? 1 : 0;
    // grid: (row groups of 16 RT rows) x (G groups of output tiles).  A weight fragment fetched from L2 serves RT row tiles, so RT
    // as large as still fills the chip with at least four tiles per workgroup; then one tile per wave (four or eight waves) with
    // the tile's whole K extent of fragments in flight, or two tiles per wave and half of it
    const int cus = umnn_num_cus(), T = (N + 15) / 16;
    const int gmax = T >= 4 ? T / 4 : 1;
    int RT = row_tiles, G = feature_groups;
    if (RT == 0) {
        RT = 1;
        for (int rt = 4; rt >= 1; rt >>= 1) {
            const long long rows = (B + 16 * rt - 1) / (16 * rt);
            if (rows * gmax >= cus || rt == 1) { RT = rt; break; }
        }
    }
    if (G == 0) {
        const long long rows = (B + 16 * RT - 1) / (16 * RT);
        const long long want = (cus + rows - 1) / rows;
        G = (int)(want < 1 ? 1 : (want > gmax ? gmax : want));
    }
    if (G > T) G = T;
    hipStream_t stream = (hipStream_t)stream_;
    const size_t lds = (size_t)RT * MF_SMAX * 2 * 64 * 8 * sizeof(unsigned short);
    const dim3 grid((unsigned)((B + 16 * RT - 1) / (16 * RT)), (unsigned)G);
    const int TG = (T + G - 1) / G;                       // tiles of the fullest workgroup
#define UMNN_ML_LAUNCH(RT_, TPC_, PD_, NW_)                                                                                \
    do {                                                                                                                   \
        if (int rc = umnn_allow_lds((const void*)made_linear_kernel<RT_, TPC_, PD_, NW_>, lds)) return rc;                 \
        hipLaunchKernelGGL((made_linear_kernel<RT_, TPC_, PD_, NW_>), grid, dim3(64 * NW_), lds, stream, a);               \
        umnn_note_made_launch("made_linear<RT=" #RT_ ",TPC=" #TPC_ ",PD=" #PD_ ",NW=" #NW_ ">");                          \
    } while (0)
// (register use of the variants, -Rpass-analysis=kernel-resource-usage at hipcc 7.2: 165..245 VGPRs, none with scratch -- the largest is
// <RT=4,TPC=2,PD=8,NW=8> at 245, two waves per SIMD)
#define UMNN_ML_PICK(RT_, PD4_, PD8_)                                                                                            \
    do {                                                                                                                   \
        if (TG <= 4) UMNN_ML_LAUNCH(RT_, 1, PD4_, 4);                                                                      \
        else if (TG <= 8) UMNN_ML_LAUNCH(RT_, 1, PD8_, 8);                                                                   \
        else UMNN_ML_LAUNCH(RT_, 2, 8, 8);                                                                                 \
    } while (0)
    if (RT == 4) UMNN_ML_PICK(4, 8, 8);                      // (four waves stage 16 operand pairs each: 128 registers beside the ring)
    else if (RT == 2) UMNN_ML_PICK(2, 16, 16);
    else UMNN_ML_PICK(1, 16, 16);
#undef UMNN_ML_PICK
#undef UMNN_ML_LAUNCH
    return umnn_check(hipGetLastError(), "made_linear launch");
}
