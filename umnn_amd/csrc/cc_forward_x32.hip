// Forward quadrature, large-batch variant on the 32x32x16 bf16 MFMA (bf16x3 arithmetic, nets of up to 63 units per
// hidden layer, at least two hidden layers).  Same reference lines as cc_forward.hip.
//
// Why a second layout: the quadrature loop is bound by VALU issue, and an MFMA only hides the vector instructions the
// same wave issues right behind it -- about two per 16-cycle 16x16x32, five to six per 32-cycle 32x32x16
// (tools/ubench/fill.hip).  The 32x32 shape does the same MACs per cycle but gives a third more issue shadow per unit of
// matrix time, enough to cover (almost) all of the activation / split / packing work (~3.4 vector instructions per
// 32-cycle MFMA here).  A wave owns two groups of 32 integrals that run half a layer out of phase, exactly like the
// two 16-point tiles of cc_fwd_bf16_kernel<PIPE>: while the matrix pipe multiplies group A by layer l, the VALU
// activates / splits / packs group B's layer l-1 output, one slice per MFMA slot.  64 points per wave need ~370
// registers, so one wave per SIMD.
//
// Layout (D = W x act, 32 features x 32 points per MFMA): lane l holds point n = l & 31 of its group in every operand;
// hg = l >> 5.  Accumulator register i of M-tile m (features 32m..32m+31) is feature 32m + 8(i>>2) + 4hg + (i&3) (the
// hardware's row order).  K-step s (16 features) of the next layer takes registers 8(s&1)..8(s&1)+7 of M-tile s>>1:
// lane half hg supplies k-slots 8hg..8hg+7, i.e. exactly its own registers -- the accumulators of one layer are, after
// activation, splitting and packing, the B operands of the next, with no cross-lane movement.  The weight fragments are
// staged with their K order permuted to match (x32_kfeat).  Biases ride on the constant-one feature at index H_l.
// The split's remainders a - bf16(a) are taken on the matrix pipe (0/-1 selection fragment, C = a), see
// cc_forward_bf16.hip.
#include <type_traits>
#include <utility>

#include "cc_bf16.h"
#include "cc_fwd_shared.h"
#include "cc_host.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FwdX32Args {
    FwdArgs f;
    int off16[UMNN_MAX_LINEAR];     // ushort offset of the fragment image of hidden layer l -> l+1
    unsigned nitems;                // work items of 64 integrals
};

// feature held in accumulator register i of M-tile m by lane half hg
__device__ __forceinline__ constexpr int x32_feat(int m, int i, int hg) { return 32 * m + 8 * (i >> 2) + 4 * hg + (i & 3); }
// feature at K index k (0..15) of K-step s (0..3): k-slot j = k & 7 of lane half k >> 3
__device__ __forceinline__ constexpr int x32_kfeat(int s, int k) {
    return 32 * (s >> 1) + 8 * (2 * (s & 1) + ((k & 7) >> 2)) + 4 * (k >> 3) + (k & 3);
}

__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int... I, class F>
__device__ __forceinline__ void x32_static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}

// fragment (output M-tile mo, K-step s, piece): 64 lanes x 8 bf16 at ((mo*4 + s)*2 + piece)*512 + lane*8 + j
__device__ __forceinline__ void x32_stage_images(const MlpDev& m, const int* off16, unsigned short* lds16, int tid, int nthreads) {
    const int L = m.n_linear - 1;
    for (int l = 1; l < L; ++l) {
        const int Hin = m.width[l], Hout = m.width[l + 1];
        const float* __restrict__ W = m.W[l];
        const float* __restrict__ b = m.b[l];
        unsigned short* img = lds16 + off16[l];
#pragma unroll 4
        for (int idx = tid; idx < 2 * 4 * 512; idx += nthreads) {
            const int j = idx & 7, ln = (idx >> 3) & 63, ms = idx >> 9;
            const int s = ms & 3, mo = ms >> 2;
            const int fo = 32 * mo + (ln & 31);
            const int fi = x32_kfeat(s, 8 * (ln >> 5) + j);
            float v = 0.f;
            if (fo < Hout) {
                if (fi < Hin) v = W[fo * Hin + fi];
                else if (fi == Hin) v = b[fo];
            } else if (fo == Hout && fi == Hin) {
                v = 1.f;
            }
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const unsigned short hb = bf16_rn_bits(v);
                img[(ms * 2 + part) * 512 + ln * 8 + j] = hb;
                v -= bf16_bits_to_f32(hb);
            }
        }
    }
}

// LIVE1: live registers per lane in M-tile 1 (features 32..63): 11 covers hidden widths up to 50, 16 everything to 63.
template <int LIVE1>
__global__ __launch_bounds__(UMNN_BLOCK, 1) void cc_fwd_x32_kernel(const FwdX32Args args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const FwdArgs& a = args.f;
    const MlpDev& m = a.m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pn = lane & 31, hg = lane >> 5;
    const int L = m.n_linear - 1;
    const int H1 = m.width[1], HL = m.width[L];
    const int E = a.E, d = a.d, n = a.n;
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    unsigned short* lds16 = reinterpret_cast<unsigned short*>(lds);
    constexpr int NSLOT = 24;
    using Slots = std::make_integer_sequence<int, NSLOT>;
    constexpr std::true_type kFirst{};
    constexpr std::false_type kLater{};

    x32_stage_images(m, args.off16, lds16, tid, UMNN_BLOCK);
    __syncthreads();

    const unsigned item = blockIdx.x * UMNN_WAVES_PER_BLOCK + wid;
    if (item >= args.nitems) return;

    // per-lane constants: first-layer column, output row (features of the registers this lane holds)
    float w1x[2][16], wout[2][16];
    {
        const float* __restrict__ W0 = m.W[0];
        const float* __restrict__ WL = m.W[L];
        const float bL = m.b[L][0];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int f = x32_feat(mt, i, hg);
                w1x[mt][i] = f < H1 ? W0[f * (1 + E)] : 0.f;
                wout[mt][i] = f < HL ? WL[f] : (f == HL ? bL : 0.f);
            }
    }
    // 0/-1 selection fragments of the split: row rho of an M-tile picks, in K-step parity q = (rho >> 4), the k-slot
    // in which lane half (rho >> 2) & 1 packed that feature: j = ((rho >> 3) & 1) * 4 + (rho & 3)
    u32x4 sel[2];
    {
        const unsigned rho = lane & 31, kg = lane >> 5;
        const unsigned j = ((rho >> 3) & 1) * 4 + (rho & 3);
        const bool mine = kg == ((rho >> 2) & 1);
        const unsigned v = 0xBF80u << (16 * (j & 1));                    // bf16(-1) in the half of its dword
        u32x4 e = {0u, 0u, 0u, 0u};
        if (mine) e[j >> 1] = v;
        sel[0] = (rho >> 4) == 0 ? e : u32x4{0u, 0u, 0u, 0u};
        sel[1] = (rho >> 4) == 1 ? e : u32x4{0u, 0u, 0u, 0u};
    }

    // the two groups of 32 integrals of this wave
    float xv[2], x0v[2], dxv[2], Facc[2] = {0.f, 0.f}, fxv[2] = {0.f, 0.f}, fx0v[2] = {0.f, 0.f};
    bool ok[2];
    long long qv[2];
    f32x16 c[2][2];                                        // hoisted first-layer term [group][M-tile]
#pragma unroll
    for (int gr = 0; gr < 2; ++gr) {
        const long long q = ((long long)item * 2 + gr) * 32 + pn;
        ok[gr] = q < a.NI;
        const long long qq = ok[gr] ? q : a.NI - 1;
        qv[gr] = qq;
        xv[gr] = a.x[qq];
        x0v[gr] = a.x0 ? a.x0[qq] : 0.f;
        dxv[gr] = xv[gr] - x0v[gr];
        const long long bi = qq / d;
        const float* hb = a.h + bi * ((long long)E * d) + (qq - bi * d);
        const float* __restrict__ W0 = m.W[0];
        const float* __restrict__ b0 = m.b[0];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int f = x32_feat(mt, i, hg);
                c[gr][mt][i] = f < H1 ? b0[f] : (f == H1 ? 1.f : 0.f);
            }
        // c += W1[:, 1:] h  on the fp32 32x32x2 MFMA: lane (row pn | col pn, k = hg) supplies one A and one B value
        for (int e0 = 0; e0 < E; e0 += 16) {
            float hv[8], Av[2][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = e0 + 2 * j + hg;
                const bool in = e < E;
                hv[j] = in ? hb[(long long)e * d] : 0.f;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int fo = 32 * mt + pn;
                    Av[mt][j] = (in && fo < H1) ? W0[fo * (1 + E) + 1 + e] : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (e0 + 2 * j < E) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        c[gr][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Av[mt][j], hv[j], c[gr][mt], 0, 0, 0);
                }
        }
    }

    u32x4 wf[2][4][2];                                     // [M-tile][K-step][piece] of the layer in flight
    u32x4 bf[2][4][2];                                     // [group][K-step][piece]
#pragma unroll
    for (int gr = 0; gr < 2; ++gr)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) bf[gr][s][k2] = u32x4{0u, 0u, 0u, 0u};
    auto frag = [&](int l, int mo, int s, int k2) {
        return *reinterpret_cast<const u32x4*>(lds16 + args.off16[l] + lane * 8 + ((mo * 4 + s) * 2 + k2) * 512);
    };
#pragma unroll
    for (int mo = 0; mo < 2; ++mo)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) wf[mo][s][k2] = frag(1, mo, s, k2);

    // GEMM slot i: K-step i/6, cross term (i/2)%3 = (W piece, activation piece) (0,0),(0,1),(1,0), M-tile i%2
    auto mfma_slot = [&](auto ic, const u32x4 (&bfin)[4][2], f32x16 (&acc)[2]) {
        constexpr int i = decltype(ic)::value;
        constexpr int s = i / 6, term = (i / 2) % 3, mo = i % 2;
        constexpr int wa = term == 2 ? 1 : 0, ba = term == 1 ? 1 : 0;
        if constexpr (i < 2) {
            f32x16 zero;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero[r] = 0.f;
            acc[mo] = mfma32(wf[mo][s][wa], bfin[s][ba], zero);
        } else {
            acc[mo] = mfma32(wf[mo][s][wa], bfin[s][ba], acc[mo]);
        }
    };
    // second group's section: fetch the next layer's fragment into registers whose last use has passed
    auto reload_slot = [&](auto ic, int lnext) {
        constexpr int i = decltype(ic)::value;
        constexpr int s = i / 6, term = (i / 2) % 3, mo = i % 2;
        if constexpr (term == 1) wf[mo][s][0] = frag(lnext, mo, s, 0);
        if constexpr (term == 2) wf[mo][s][1] = frag(lnext, mo, s, 1);
    };
    // Packing of one group, in place on its raw pre-activations z[2] (f32x16 per M-tile), one slice per MFMA slot:
    //   A(mt, i)  register i of M-tile mt: z <- LeakyReLU(z)  (FIRST: z = w1x * t_k + c first)
    //   H(s)      leading pieces of K-step s (registers 8(s&1).. of M-tile s>>1) packed into bf[s][0]
    //   R(s)      z[s>>1] <- z[s>>1] - bf16(.) for that K-step's 16 features, on the matrix pipe
    //   Lo(s)     second pieces into bf[s][1] (after both R of the M-tile)
    auto act_reg = [&](auto first, f32x16& z, const float (&w)[16], const f32x16& cv, float tkv, int i) {
        constexpr bool FIRST = decltype(first)::value;
        if constexpr (FIRST) z[i] = fmaf(w[i], tkv, cv[i]);
        z[i] = hidden_act_f(z[i], slope);
    };
    auto cvt4 = [&](const f32x16& z, int base, int nlive) {      // registers base..base+7 -> 4 packed dwords
        u32x4 o = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const int i0 = base + 2 * pr;
            if (i0 < nlive) {
                const bf16x2 hh = __builtin_convertvector(f32x2{z[i0], i0 + 1 < nlive ? z[i0 + 1] : 0.f}, bf16x2);
                o[pr] = __builtin_bit_cast(unsigned, hh);
            }
        }
        return o;
    };
    auto pack_slot = [&](auto ic, auto first, f32x16 (&z)[2], u32x4 (&bfo)[4][2], float tkv, const f32x16 (&cv)[2]) {
        constexpr int i = decltype(ic)::value;
        // M-tile 0: registers 0..15 in slots 0-3 and 5-8 (two per slot)
        if constexpr (i <= 3) { act_reg(first, z[0], w1x[0], cv[0], tkv, 2 * i); act_reg(first, z[0], w1x[0], cv[0], tkv, 2 * i + 1); }
        if constexpr (i >= 5 && i <= 8) { act_reg(first, z[0], w1x[0], cv[0], tkv, 2 * i - 2); act_reg(first, z[0], w1x[0], cv[0], tkv, 2 * i - 1); }
        if constexpr (i == 4) bfo[0][0] = cvt4(z[0], 0, 16);
        if constexpr (i == 9) bfo[1][0] = cvt4(z[0], 8, 16);
        if constexpr (i == 10) z[0] = mfma32(sel[0], bfo[0][0], z[0]);
        if constexpr (i == 12) z[0] = mfma32(sel[1], bfo[1][0], z[0]);
        if constexpr (i == 14) bfo[0][1] = cvt4(z[0], 0, 16);
        if constexpr (i == 16) bfo[1][1] = cvt4(z[0], 8, 16);
        // M-tile 1: registers 0..7 in slots 10-13, the rest in 17 (and 18)
        if constexpr (i >= 10 && i <= 13) { act_reg(first, z[1], w1x[1], cv[1], tkv, 2 * i - 20); act_reg(first, z[1], w1x[1], cv[1], tkv, 2 * i - 19); }
        if constexpr (i == 15) bfo[2][0] = cvt4(z[1], 0, LIVE1);
        if constexpr (i == 17) {
#pragma unroll
            for (int r = 8; r < (LIVE1 < 12 ? LIVE1 : 12); ++r) act_reg(first, z[1], w1x[1], cv[1], tkv, r);
        }
        if constexpr (i == 18) {
#pragma unroll
            for (int r = 12; r < LIVE1; ++r) act_reg(first, z[1], w1x[1], cv[1], tkv, r);
        }
        if constexpr (i == 19) {
            bfo[3][0] = cvt4(z[1], 8, LIVE1);
            z[1] = mfma32(sel[0], bfo[2][0], z[1]);
        }
        if constexpr (i == 21) z[1] = mfma32(sel[1], bfo[3][0], z[1]);
        if constexpr (i == 23) { bfo[2][1] = cvt4(z[1], 0, LIVE1); bfo[3][1] = cvt4(z[1], 8, LIVE1); }
    };
    // output dot product of one group, spread over the slots of the last section
    auto dot_slot = [&](auto ic, const f32x16 (&z)[2], float& sd) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < 16) sd = fmaf(wout[0][i], hidden_act_f(z[0][i], slope), sd);
        if constexpr (i >= 16 && i - 16 < LIVE1) sd = fmaf(wout[1][i - 16], hidden_act_f(z[1][i - 16], slope), sd);
        if constexpr (i + 8 < LIVE1) sd = fmaf(wout[1][i + 8], hidden_act_f(z[1][i + 8], slope), sd);
    };

    f32x16 acc0[2], acc1[2];                               // raw layer outputs of the two groups (dead rows stay zero)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[mt][r] = 0.f; acc1[mt][r] = 0.f; }
    for (int k = 0; k <= n; ++k) {
        const float u = a.ccs[k] + 1.f;
        const float wk = a.ccw[k];
        float tk[2];
#pragma unroll
        for (int gr = 0; gr < 2; ++gr) tk[gr] = k == 0 ? xv[gr] : __fadd_rn(x0v[gr], __fmul_rn(dxv[gr], u) * 0.5f);
        // first group: layer 1 on the VALU, nothing to hide behind yet
        x32_static_for(Slots{}, [&](auto ic) { pack_slot(ic, kFirst, acc0, bf[0], tk[0], c[0]); });
        __builtin_amdgcn_sched_barrier(0);
        // section A of layer 1
        x32_static_for(Slots{}, [&](auto ic) {
            mfma_slot(ic, bf[0], acc0);
            pack_slot(ic, kFirst, acc1, bf[1], tk[1], c[1]);
            __builtin_amdgcn_sched_barrier(0);
        });
        for (int l = 1; l + 1 < L; ++l) {
            x32_static_for(Slots{}, [&](auto ic) {                     // section B of layer l
                mfma_slot(ic, bf[1], acc1);
                pack_slot(ic, kLater, acc0, bf[0], 0.f, c[0]);
                reload_slot(ic, l + 1);
                __builtin_amdgcn_sched_barrier(0);
            });
            x32_static_for(Slots{}, [&](auto ic) {                     // section A of layer l+1
                mfma_slot(ic, bf[0], acc0);
                pack_slot(ic, kLater, acc1, bf[1], 0.f, c[1]);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        float sd0 = 0.f, sd1 = 0.f;
        x32_static_for(Slots{}, [&](auto ic) {                         // section B of the last hidden layer
            mfma_slot(ic, bf[1], acc1);
            dot_slot(ic, acc0, sd0);
            reload_slot(ic, 1);
            __builtin_amdgcn_sched_barrier(0);
        });
        x32_static_for(Slots{}, [&](auto ic) { dot_slot(ic, acc1, sd1); });
#pragma unroll
        for (int gr = 0; gr < 2; ++gr) {
            const float part = gr == 0 ? sd0 : sd1;
            const unsigned ub = __float_as_uint(part);
            auto sw = __builtin_amdgcn_permlane32_swap(ub, ub, false, false);   // {lane half 0's, lane half 1's} value
            const float sr = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            const float f = out_act_f(sr, m.out_act);
            Facc[gr] = fmaf(wk, a.inv_f ? 1.f / f : f, Facc[gr]);
            if (k == 0) fxv[gr] = f;
            if (k == n) fx0v[gr] = f;
        }
    }

    if (hg == 0) {
#pragma unroll
        for (int gr = 0; gr < 2; ++gr) {
            if (!ok[gr]) continue;
            const long long q = qv[gr];
            const float Fv = Facc[gr] * dxv[gr] * 0.5f;
            if (a.F) a.F[q] = Fv;
            if (a.fx) a.fx[q] = fxv[gr];
            if (a.fx0) a.fx0[q] = fx0v[gr];
            if (a.scaling) {
                const long long bi = q / d;
                const int i = (int)(q - bi * d);
                const float sc = a.scaling[i];
                const float z0 = a.h[bi * ((long long)E * d) + i];
                a.z[a.reverse_z ? bi * d + (d - 1 - i) : q] = __expf(sc) * (Fv + z0);
                const float lj = __logf(fxv[gr] + 1e-10f) + sc;
                a.logjac[q] = a.logjac_in ? a.logjac_in[q] + lj : lj;
            }
        }
    }
}

// Launches the 32x32 variant when asked to (UMNN_FWD_X32=1) and the shape is in its family (bf16x3; 2..7 hidden layers,
// none wider than 63); UMNN_EUNSUPPORTED otherwise (the caller continues with the 16x16 kernels).
// MEASURED (round 1, C3): 3.21 ms per launch against 2.95 ms for cc_fwd_bf16_kernel<PIPE> -- 64 points per wave cost
// ~320 registers = one wave per SIMD, and with no partner wave every MFMA -> VALU dependency at a section boundary is
// exposed (63 % matrix-pipe busy against 70 %).  Kept as an opt-in, parity-tested variant; not the default.
int umnn_launch_forward_x32(FwdArgs& a, const umnn_mlp* net, int nb_steps, hipStream_t stream) {
    const char* ev = getenv("UMNN_FWD_X32");
    const int env = ev ? atoi(ev) : 0;
    if (env == 0) return UMNN_EUNSUPPORTED;
    const int L = a.m.n_linear - 1;
    if (L < 2) return UMNN_EUNSUPPORTED;
    int hmax = 0;
    for (int l = 1; l <= L; ++l) {
        if (a.m.t_out[l] > 4) return UMNN_EUNSUPPORTED;
        hmax = a.m.width[l] > hmax ? a.m.width[l] : hmax;
    }
    const unsigned nitems = (unsigned)((a.NI + 63) / 64);
    FwdX32Args args;
    args.f = a;
    args.nitems = nitems;
    int off16 = 0;
    for (int l = 1; l < L; ++l) { args.off16[l] = off16; off16 += 2 * 4 * 2 * 512; }
    const size_t lds_bytes = (size_t)off16 * sizeof(unsigned short);
    if (lds_bytes > 160 * 1024) return UMNN_EUNSUPPORTED;
    const bool small = hmax <= 50;
    void (*kfn)(const FwdX32Args) = small ? cc_fwd_x32_kernel<11> : cc_fwd_x32_kernel<16>;
    const char* kname = small ? "cc_fwd_x32<LIVE1=11>" : "cc_fwd_x32<LIVE1=16>";
    if (int rc = umnn_allow_lds((const void*)kfn, lds_bytes)) return rc;
    const unsigned nblk = (nitems + UMNN_WAVES_PER_BLOCK - 1) / UMNN_WAVES_PER_BLOCK;
    umnn_prof_begin(stream);
    hipLaunchKernelGGL(kfn, dim3(nblk), dim3(UMNN_BLOCK), lds_bytes, stream, args);
    umnn_prof_end(stream, umnn_cc_forward_flops_per_integral(net, nb_steps) * (double)a.NI);
    umnn_note_launch(kname);
    return umnn_check(hipGetLastError(), "cc_fwd_x32 launch");
}
