// Forward Clenshaw-Curtis quadrature of the integrand MLP, fused: one pass emits
//   F(x) = (x-x0)/2 * sum_k w_k f(t_k; h),   f(x; h) (node 0)   and   f(x0; h) (node n)
// and optionally the UMNNMAF block epilogue z = exp(s)(F + h_0), log_jac = log(f(x)+1e-10) + s.
//
// Replaces (reference, never materialising the node axis):
//   models/UMNN/ParallelNeuralIntegral.py:49-65  integrate(compute_grad=False)
//   models/UMNN/NeuralIntegral.py:53-66          sequential variant (same arithmetic)
//   models/UMNN/UMNNMAF.py:263-284               IntegrandNetwork.forward on every node
//   models/UMNN/UMNNMAF.py:80-83,134,138-139     block epilogue (flow entry point)
//
// Work decomposition: a wave owns P tiles of 16 integrals (lane&15 = integral within the tile) and
// walks a contiguous range of quadrature nodes; the node sum is a per-lane register accumulation.
// NS in {1,2,4} waves of a workgroup may share one tile group and split the node range (small
// problems), their partial sums meeting in LDS.  Per node: layer 1 is one FMA per feature (the
// node-invariant part W1[:,1:]*h + b1 is hoisted and computed once per integral, itself on MFMA);
// hidden layers are 16x16x4 fp32 MFMAs against LDS-resident weight images; the scalar output layer
// is a per-lane dot product plus a 4-lane-group all-reduce.  See cc_common.h for the layouts.
#ifndef UMNN_ASM_TIED
#define UMNN_ASM_TIED 1      // cc_common.h: inline-assembly outputs tied to inputs in the forward translation units
#endif
#include "cc_fwd_shared.h"

// TAIL = 1 (exact variants only): the last tile holds at most 4 features (<= 3 real ones + the constant), all
// in component r = 0.  Giving that tile 13 MFMAs (416 cycles) to produce 2-3 useful rows costs more than computing
// them on the VALU (~45 instructions, ~130 cycles; MFMA and VALU time add on this hardware): each
// lane dots its own K-slice of the tail rows (weights broadcast from LDS as b128) and the four lane groups are
// summed with row/half swaps.
template <int TMAX, int KSC, int P, int TAIL, int NT>
__global__ __launch_bounds__(UMNN_BLOCK) void cc_fwd_kernel(const FwdArgs a) {
    constexpr int TT = TMAX - TAIL;            // tiles produced by MFMA; NT = real features of the VALU tail tile
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const MlpDev& m = a.m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, p = lane & 15;
    const int L = m.n_linear - 1;             // hidden layers
    const int H1 = m.width[1], HL = m.width[L];
    const int E = a.E, d = a.d, n = a.n;
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;

    stage_hidden_images(m, lds, tid, UMNN_BLOCK);
    __syncthreads();

    // ---- which tile group / node range does this wave own?
    const int ns = a.ns;                                      // 1, 2 or 4
    const int sub = wid / ns, part = wid % ns;
    const unsigned gpb = UMNN_WAVES_PER_BLOCK / ns;           // tile groups per workgroup
    const unsigned grp = xcd_remap(blockIdx.x, gridDim.x) * gpb + sub;
    const bool live = grp < a.ngroups;
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

        // ---- per-lane constants: first-layer x-column and output-layer row (+ its bias)
        float w1x[TMAX][4], wout[TMAX][4];
        {
            const float* __restrict__ W0 = m.W[0];
            const float* __restrict__ WL = m.W[L];
            const float bL = m.b[L][0];
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = feat_of(t, r, g);
                    w1x[t][r] = f < H1 ? W0[f * (1 + E)] : 0.f;
                    wout[t][r] = f < HL ? WL[f] : (f == HL ? bL : 0.f);
                }
        }

        // ---- hoisted first-layer term c = W1[:,1:] h + b1 (and the constant-one feature), on MFMA
        f32x4 c[P][TMAX];
        {
            const float* __restrict__ W0 = m.W[0];
            const float* __restrict__ b0 = m.b[0];
            const int t1 = KSC ? TMAX : m.t_out[1];
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                f32x4 init;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = feat_of(t, r, g);
                    init[r] = f < H1 ? b0[f] : (f == H1 ? 1.f : 0.f);
                }
#pragma unroll
                for (int pt = 0; pt < P; ++pt) c[pt][t] = init;
            }
            for (int se = 0; se < (E + 3) / 4; ++se) {
                const int e = 4 * se + g;
                float hv[P];
#pragma unroll
                for (int pt = 0; pt < P; ++pt) hv[pt] = e < E ? hb[pt][(long long)e * d] : 0.f;
#pragma unroll
                for (int t = 0; t < TMAX; ++t) {
                    if (t < t1) {
                        const int fo = fout_of(t, p);
                        const float A = (fo < H1 && e < E) ? W0[fo * (1 + E) + 1 + e] : 0.f;
#pragma unroll
                        for (int pt = 0; pt < P; ++pt) c[pt][t] = mfma16(A, hv[pt], c[pt][t]);
                    }
                }
            }
        }

        // ---- walk the quadrature nodes
        for (int k = k_lo; k < k_hi; ++k) {
            const float u = a.ccs[k] + 1.f;
            const float wk = a.ccw[k];
            f32x4 act[P][TMAX];
#pragma unroll
            for (int pt = 0; pt < P; ++pt) {
                // t_k = x0 + (x-x0)*(s_k+1)/2 in the reference's rounding order; node 0 is x itself
                const float tk = k == 0 ? xv[pt] : __fadd_rn(x0v[pt], __fmul_rn(dxv[pt], u) * 0.5f);
#pragma unroll
                for (int t = 0; t < TMAX; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        act[pt][t][r] = (t < TT || r == 0) ? hidden_act_f(fmaf(w1x[t][r], tk, c[pt][t][r]), slope) : 0.f;
            }

            for (int l = 1; l < L; ++l) {
                const int ks = KSC ? KSC : m.ks_in[l];
                const int to = KSC ? TT : m.t_out[l + 1];
                const float* img = lds + m.lds_off[l] + lane;
                // tail rows first (VALU, reads the layer's INPUT activations; independent of the MFMAs below)
                float tailv[P];
                if constexpr (TAIL) {
                    const float* wt = lds + m.tail_off + (l - 1) * (3 * 4 * 16) + g * 16;
#pragma unroll
                    for (int pt = 0; pt < P; ++pt) tailv[pt] = g == NT ? 1.f : 0.f;   // the constant feature
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        f32x4 w[TT];
#pragma unroll
                        for (int t = 0; t < TT; ++t) w[t] = *reinterpret_cast<const f32x4*>(wt + j * 64 + 4 * t);
                        const float wl = wt[j * 64 + 4 * TT];
#pragma unroll
                        for (int pt = 0; pt < P; ++pt) {
                            float dsum = wl * act[pt][TT][0];
#pragma unroll
                            for (int t = 0; t < TT; ++t)
#pragma unroll
                                for (int r = 0; r < 4; ++r) dsum = fmaf(w[t][r], act[pt][t][r], dsum);
                            dsum = group_allreduce(dsum);
                            tailv[pt] = g == j ? hidden_act_f(dsum, slope) : tailv[pt];
                        }
                    }
                }
                f32x4 acc[P][TMAX];
#pragma unroll
                for (int pt = 0; pt < P; ++pt)
#pragma unroll
                    for (int t = 0; t < TMAX; ++t) acc[pt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4 * TMAX; ++s) {
                    if (KSC ? (s < KSC) : (s < ks)) {
#pragma unroll
                        for (int t = 0; t < TT; ++t) {
                            if (KSC || t < to) {
                                const float A = img[(t * ks + s) * 64];
#pragma unroll
                                for (int pt = 0; pt < P; ++pt)
                                    acc[pt][t] = mfma16(A, act[pt][s >> 2][s & 3], acc[pt][t]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int pt = 0; pt < P; ++pt)
#pragma unroll
                    for (int t = 0; t < TMAX; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (t < TT) act[pt][t][r] = (KSC || t < to) ? hidden_act_f(acc[pt][t][r], slope) : 0.f;
                            else act[pt][t][r] = r == 0 ? tailv[pt] : 0.f;
                        }
            }

#pragma unroll
            for (int pt = 0; pt < P; ++pt) {
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < TMAX; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (t < TT || r == 0) s = fmaf(wout[t][r], act[pt][t][r], s);
                s = group_allreduce(s);
                const float f = out_act_f(s, m.out_act);
                Facc[pt] = fmaf(wk, maybe_inverse(f, a.inv_f), Facc[pt]);
                if (k == 0) fxv[pt] = f;
                if (k == n) fx0v[pt] = f;
            }
        }
    }

    fwd_epilogue<P>(a, lds, Facc, fxv, fx0v, ok, qv, dxv, live, part, ns, wid, g, p);
}


// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
#include <cstring>

#include "cc_host.h"

typedef void (*fwd_kernel_t)(const FwdArgs);

struct FwdVariant { int tmax, ksc, p, tail, nt; fwd_kernel_t fn; const char* name; };

#define FWD_VARIANT(T, K, PP, TL, NT) { T, K, PP, TL, NT, cc_fwd_kernel<T, K, PP, TL, NT>, "cc_fwd<T=" #T ",KS=" #K ",P=" #PP ",TAIL=" #TL ">" }
static const FwdVariant kFwdVariants[] = {
    FWD_VARIANT(4, 13, 1, 1, 2), FWD_VARIANT(4, 13, 2, 1, 2),   // hidden width 50 (UCI / VAE nets): VALU tail for 48,49
    FWD_VARIANT(4, 13, 1, 0, 0), FWD_VARIANT(4, 13, 2, 0, 0),   // widths 48..51, all four tiles on MFMA
    FWD_VARIANT(7, 26, 1, 0, 0), FWD_VARIANT(7, 26, 2, 0, 0),   // hidden width 100 (toy / MonotonicMLP nets)
    FWD_VARIANT(2, 0, 1, 0, 0),  FWD_VARIANT(2, 0, 2, 0, 0),    // generic, hidden widths <= 31
    FWD_VARIANT(4, 0, 1, 0, 0),  FWD_VARIANT(4, 0, 2, 0, 0),    // generic, <= 63
    FWD_VARIANT(8, 0, 1, 0, 0),  FWD_VARIANT(8, 0, 2, 0, 0),    // generic, <= 127
};

struct FwdOvfPlan;        // cc_forward_bf16.hip: the queued-fallback plan of an fp16-piece launch (null = a plain launch)
int umnn_launch_forward_bf16(FwdArgs& a, const umnn_mlp* net, int nparts, int P, int ns, int nb_steps, hipStream_t stream, const FwdOvfPlan* ovf);
int umnn_launch_forward_f16(FwdArgs& a, const umnn_mlp* net, int nparts, int P, int ns, int nb_steps, hipStream_t stream, const FwdOvfPlan* ovf);     // cc_forward_f16.hip

static int launch_forward(const umnn_mlp* net, const float* x0, const float* x, const float* h,
                          const float* scaling, const float* cc_w, const float* cc_s, int nb_steps,
                          long long B, int d, int E, int inv_f,
                          float* F, float* f_x, float* f_x0, float* z, float* log_jac, hipStream_t stream,
                          int reverse_z = 0, const float* log_jac_in = nullptr,
                          float* ll = nullptr, unsigned* row_cnt = nullptr, int ll_first = 0, int ll_last = 0,
                          const umnn_io* io = nullptr, float* z2_save = nullptr) {
    if (int rc = umnn_check_io(io)) return rc;
    FwdArgs a;
    int tmax = 0, ksu = 0;
    if (int rc = umnn_prepare_mlp(net, E, &a.m, &tmax, &ksu)) return rc;
    if (nb_steps < 1) return umnn_fail(UMNN_EINVAL, "forward: nb_steps must be >= 1");
    if (B < 0 || d < 1) return umnn_fail(UMNN_EINVAL, "forward: B must be >= 0 and d >= 1");
    if (B == 0) return 0;      // empty batch: nothing to read or write (zero-size buffers may be null)
    if (!x || !h || !cc_w || !cc_s) return umnn_fail(UMNN_EINVAL, "forward: x, h, cc_w, cc_s must be non-null");
    if (!scaling && !F) return umnn_fail(UMNN_EINVAL, "forward: F must be non-null");
    if (scaling && (!z || !log_jac)) return umnn_fail(UMNN_EINVAL, "flow forward: z and log_jac must be non-null");

    a.x0 = x0; a.x = x; a.h = h; a.ccw = cc_w; a.ccs = cc_s;
    a.F = F; a.fx = f_x; a.fx0 = f_x0; a.scaling = scaling; a.z = z; a.logjac = log_jac;
    a.logjac_in = log_jac_in; a.reverse_z = reverse_z;
    a.ll = ll; a.row_cnt = row_cnt; a.ll_first = ll_first; a.ll_last = ll_last;
    a.x_bf16 = io && io->x_dtype == UMNN_DTYPE_BF16; a.h_bf16 = io && io->h_dtype == UMNN_DTYPE_BF16;
    a.inv_z = nullptr; a.inv_x = nullptr; a.inv_j = 0; a.inv_iters = 0;
    if (ll && a.x_bf16) return umnn_fail(UMNN_EINVAL, "flow ll forward: z / log_jac scratch must be fp32");
    a.NI = B * (long long)d; a.d = d; a.E = E; a.n = nb_steps; a.inv_f = inv_f;
    a.ovf_mode = 0; a.ovf_flag = nullptr; a.ovf_gen = 0;
    a.z2_save = z2_save; a.z2_nl2 = z2_save ? (net->widths[2] + 1 + 3) / 4 : 0;

    // ---- choose the variant: exact (compile-time K-steps) when all hidden layers share a width we
    // instantiated, otherwise the smallest generic tile count that fits
    const long long tiles16 = (a.NI + 15) / 16;
    const int simd_slots = umnn_num_cus() * 4;
    int P = tiles16 >= 8LL * simd_slots ? 2 : 1;
    int ns = 1;
    if (tiles16 < 2LL * simd_slots) ns = tiles16 * 2 <= 2LL * simd_slots ? 4 : 2;
    if (ns > nb_steps + 1) ns = 1;
    const UmnnOptions& opt = umnn_options();
    const int opt_p = opt.fwd_p, opt_ns = opt.fwd_ns, opt_tail = opt.fwd_tail;
    if (opt_p > 0) P = opt_p;
    if (opt_ns > 0) ns = opt_ns;

    if (z2_save && (opt.fwd_precision == UMNN_PRECISION_FP32 || opt.fwd_precision == UMNN_PRECISION_BF16X6))
        return umnn_fail(UMNN_EUNSUPPORTED, "forward (z_2 saved): two-piece arithmetic modes only");
    // bf16-split kernels (default): hidden GEMMs on the bf16 matrix cores; falls through to fp32 MFMA when the
    // shape does not fit them (a single hidden layer has no hidden->hidden GEMM at all)
    const int prec = opt.fwd_precision;
    if (prec != UMNN_PRECISION_FP32 && a.m.n_linear - 1 >= 2) {
        a.ns = ns;
        // f16x3 (default): two fp16 pieces, three cross terms (fp32-level), the bf16x3 build queued behind it for tile groups whose
        // pieces overflow (cc_forward_bf16.hip).  The deferred groups are marked IN the output (F, or z), so a launch whose marker
        // output aliases one of its inputs runs bf16x3 outright -- the arithmetic its overflowing groups would get anyway.  Shapes
        // that family does not cover fall through to the three-piece bf16 kernels (the same accuracy class), then to fp32 MFMA.
        if (prec == UMNN_PRECISION_F16X3) {
            const float* marker = F ? F : z;
            if (marker == x || marker == x0 || marker == h) {
                const int rc2 = umnn_launch_forward_bf16(a, net, 2, P, ns, nb_steps, stream, nullptr);
                if (rc2 != UMNN_EUNSUPPORTED) return rc2;
            } else {
                const int rc16 = umnn_launch_forward_f16(a, net, 2, P, ns, nb_steps, stream, nullptr);
                if (rc16 != UMNN_EUNSUPPORTED) return rc16;
            }
        }
        const int rc = umnn_launch_forward_bf16(a, net, prec == UMNN_PRECISION_BF16X3 ? 2 : 3, P, ns, nb_steps, stream, nullptr);
        if (rc != UMNN_EUNSUPPORTED) return rc;
    }
    if (z2_save) return umnn_fail(UMNN_EUNSUPPORTED, "forward (z_2 saved): this net / launch is not served by the wide-first kernels");
    // TAIL is possible when every hidden layer has the same width H with 16(T-1) <= H <= 16(T-1)+3
    const int H1w = net->widths[1];
    int want_tail = (ksu && ksu == 4 * (tmax - 1) + 1 && H1w - 16 * (tmax - 1) >= 0 && H1w - 16 * (tmax - 1) <= 3) ? 1 : 0;
    if (opt_tail >= 0) want_tail = want_tail && opt_tail != 0;
    const FwdVariant* pick = nullptr;
    for (int tl = want_tail; tl >= 0 && !pick; --tl)
        for (const FwdVariant& v : kFwdVariants)
            if (v.ksc && v.ksc == ksu && v.tmax == tmax && v.p == P && v.tail == tl &&
                (!tl || v.nt == H1w - 16 * (tmax - 1))) { pick = &v; break; }
    if (!pick)
        for (const FwdVariant& v : kFwdVariants)
            if (!v.ksc && v.tmax >= tmax && v.p == P) { pick = &v; break; }
    if (!pick) return umnn_fail(UMNN_EUNSUPPORTED, "forward: hidden width above UMNN_MAX_HIDDEN_WIDTH");
    const fwd_kernel_t kfn = pick->fn;
    const char* kname = pick->name;

    const int L = a.m.n_linear - 1;
    if (pick->tail) {       // re-lay the LDS plan: T-1 tiles per image, then the tail rows
        int off = 0;
        for (int l = 1; l <= L; ++l) a.m.t_mfma[l] = a.m.t_out[l] - 1;
        for (int l = 1; l < L; ++l) { a.m.lds_off[l] = off; off += a.m.t_mfma[l + 1] * a.m.ks_in[l] * 64; }
        a.m.tail_off = off;
        a.m.n_tail = H1w - 16 * (tmax - 1);
        off += (L - 1) * 3 * 4 * 16;
        a.m.lds_off[L] = off;
    }
    const size_t lds_floats = (size_t)a.m.lds_off[L] + (ns > 1 ? UMNN_WAVES_PER_BLOCK * 3 * P * 16 : 0);
    const size_t lds_bytes = lds_floats * sizeof(float);
    if (lds_bytes > 160 * 1024) return umnn_fail(UMNN_EUNSUPPORTED, "forward: weight images exceed 160 KiB of LDS");
    if (int rc = umnn_allow_lds((const void*)kfn, lds_bytes)) return rc;

    a.ns = ns;
    a.ngroups = (unsigned)((a.NI + 16 * P - 1) / (16 * P));
    const unsigned gpb = UMNN_WAVES_PER_BLOCK / ns;
    const unsigned nblk = (a.ngroups + gpb - 1) / gpb;
    umnn_prof_begin(stream);
    hipLaunchKernelGGL(kfn, dim3(nblk), dim3(UMNN_BLOCK), lds_bytes, stream, a);
    umnn_prof_end(stream, umnn_cc_forward_flops_per_integral(net, nb_steps) * (double)a.NI);
    umnn_note_launch(kname);
    return umnn_check(hipGetLastError(), "cc_fwd launch");
}

extern "C" int umnn_cc_forward(const umnn_mlp* net, const float* x0, const float* x, const float* h,
                               const float* cc_w, const float* cc_s, int nb_steps,
                               long long B, int d, int E, int inv_f,
                               float* F, float* f_x, float* f_x0, void* stream) {
    return launch_forward(net, x0, x, h, nullptr, cc_w, cc_s, nb_steps, B, d, E, inv_f, F, f_x, f_x0,
                          nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int umnn_flow_block_forward(const umnn_mlp* net, const float* x, const float* h, const float* scaling,
                                       const float* cc_w, const float* cc_s, int nb_steps,
                                       long long B, int d, int E,
                                       float* z, float* log_jac, float* f_x, float* f_x0, void* stream) {
    if (!scaling) return umnn_fail(UMNN_EINVAL, "flow forward: scaling must be non-null");
    return launch_forward(net, nullptr, x, h, scaling, cc_w, cc_s, nb_steps, B, d, E, 0, nullptr, f_x, f_x0,
                          z, log_jac, (hipStream_t)stream);
}

extern "C" int umnn_flow_stack_block_forward(const umnn_mlp* net, const float* x, const float* h, const float* scaling,
                                             const float* cc_w, const float* cc_s, int nb_steps,
                                             long long B, int d, int E, int reverse_z, const float* log_jac_in,
                                             float* z, float* log_jac, float* f_x, float* f_x0, void* stream) {
    if (!scaling) return umnn_fail(UMNN_EINVAL, "flow forward: scaling must be non-null");
    if (z == x && reverse_z) return umnn_fail(UMNN_EINVAL, "flow forward: z must not alias x when reverse_z is set");
    return launch_forward(net, nullptr, x, h, scaling, cc_w, cc_s, nb_steps, B, d, E, 0, nullptr, f_x, f_x0,
                          z, log_jac, (hipStream_t)stream, reverse_z != 0, log_jac_in);
}

// The training forward of a block whose backward will be the three-stage family (cc_backward_front.hip): as
// umnn_flow_stack_block_forward, and the pre-activations of hidden layer 2 at every node are left in z2_save for umnn_cc_backward_saved.
extern "C" long long umnn_cc_forward_z2_floats(const umnn_mlp* net, long long B, int d, int E, int nb_steps) {
    MlpDev m;
    int tmax = 0, ksu = 0;
    if (umnn_prepare_mlp(net, E, &m, &tmax, &ksu) || B <= 0 || d < 1 || nb_steps < 1) return 0;
    const int L = m.n_linear - 1;
    if (L < 3 || L > 5 || m.t_out[1] < 5 || m.t_out[1] > 8) return 0;
    for (int l = 2; l <= L; ++l) if (m.t_out[l] > 4) return 0;
    const UmnnOptions& opt = umnn_options();
    if (opt.fwd_precision == UMNN_PRECISION_FP32 || opt.fwd_precision == UMNN_PRECISION_BF16X6 || opt.bwd_precision != UMNN_PRECISION_BF16X3) return 0;
    const long long tiles = (B * (long long)d + 15) / 16, nl2 = (net->widths[2] + 1 + 3) / 4;
    return tiles * (nb_steps + 1) * nl2 * 64;
}
extern "C" int umnn_flow_stack_block_forward_save(const umnn_mlp* net, const float* x, const float* h, const float* scaling,
                                                  const float* cc_w, const float* cc_s, int nb_steps,
                                                  long long B, int d, int E, int reverse_z, const float* log_jac_in,
                                                  float* z, float* log_jac, float* f_x, float* f_x0,
                                                  float* z2_save, long long z2_floats, void* stream) {
    if (!scaling) return umnn_fail(UMNN_EINVAL, "flow forward: scaling must be non-null");
    if (z == x && reverse_z) return umnn_fail(UMNN_EINVAL, "flow forward: z must not alias x when reverse_z is set");
    const long long need = umnn_cc_forward_z2_floats(net, B, d, E, nb_steps);
    if (need == 0) return umnn_fail(UMNN_EUNSUPPORTED, "flow forward (z_2 saved): not the wide-first family / arithmetic mode");
    if (!z2_save || z2_floats < need) return umnn_fail(UMNN_EINVAL, "flow forward (z_2 saved): buffer smaller than umnn_cc_forward_z2_floats()");
    return launch_forward(net, nullptr, x, h, scaling, cc_w, cc_s, nb_steps, B, d, E, 0, nullptr, f_x, f_x0,
                          z, log_jac, (hipStream_t)stream, reverse_z != 0, log_jac_in, nullptr, nullptr, 0, 0, nullptr, z2_save);
}

// ... with bf16 or fp32 activation storage (configuration C4's embedding in bf16: z_2 itself stays fp32 -- it is scratch between two kernels)
extern "C" int umnn_flow_stack_block_forward_save_io(const umnn_mlp* net, const umnn_io* io, const void* x, const void* h, const float* scaling,
                                                     const float* cc_w, const float* cc_s, int nb_steps,
                                                     long long B, int d, int E, int reverse_z, const void* log_jac_in,
                                                     void* z, void* log_jac, void* f_x, void* f_x0,
                                                     float* z2_save, long long z2_floats, void* stream) {
    if (!scaling) return umnn_fail(UMNN_EINVAL, "flow forward: scaling must be non-null");
    if (z == x && reverse_z) return umnn_fail(UMNN_EINVAL, "flow forward: z must not alias x when reverse_z is set");
    const long long need = umnn_cc_forward_z2_floats(net, B, d, E, nb_steps);
    if (need == 0) return umnn_fail(UMNN_EUNSUPPORTED, "flow forward (z_2 saved): not the wide-first family / arithmetic mode");
    if (!z2_save || z2_floats < need) return umnn_fail(UMNN_EINVAL, "flow forward (z_2 saved): buffer smaller than umnn_cc_forward_z2_floats()");
    return launch_forward(net, nullptr, (const float*)x, (const float*)h, scaling, cc_w, cc_s, nb_steps, B, d, E, 0, nullptr,
                          (float*)f_x, (float*)f_x0, (float*)z, (float*)log_jac, (hipStream_t)stream, reverse_z != 0,
                          (const float*)log_jac_in, nullptr, nullptr, 0, 0, io, z2_save);
}

extern "C" int umnn_cc_forward_io(const umnn_mlp* net, const umnn_io* io, const void* x0, const void* x, const void* h,
                                  const float* cc_w, const float* cc_s, int nb_steps,
                                  long long B, int d, int E, int inv_f, void* F, void* f_x, void* f_x0, void* stream) {
    return launch_forward(net, (const float*)x0, (const float*)x, (const float*)h, nullptr, cc_w, cc_s, nb_steps, B, d, E, inv_f,
                          (float*)F, (float*)f_x, (float*)f_x0, nullptr, nullptr, (hipStream_t)stream, 0, nullptr,
                          nullptr, nullptr, 0, 0, io);
}

extern "C" int umnn_flow_stack_block_forward_io(const umnn_mlp* net, const umnn_io* io, const void* x, const void* h,
                                                const float* scaling, const float* cc_w, const float* cc_s, int nb_steps,
                                                long long B, int d, int E, int reverse_z, const void* log_jac_in,
                                                void* z, void* log_jac, void* f_x, void* f_x0, void* stream) {
    if (!scaling) return umnn_fail(UMNN_EINVAL, "flow forward: scaling must be non-null");
    if (z == x && reverse_z) return umnn_fail(UMNN_EINVAL, "flow forward: z must not alias x when reverse_z is set");
    return launch_forward(net, nullptr, (const float*)x, (const float*)h, scaling, cc_w, cc_s, nb_steps, B, d, E, 0, nullptr,
                          (float*)f_x, (float*)f_x0, (float*)z, (float*)log_jac, (hipStream_t)stream, reverse_z != 0,
                          (const float*)log_jac_in, nullptr, nullptr, 0, 0, io);
}

extern "C" int umnn_flow_ll_block_forward(const umnn_mlp* net, const float* x, const float* h, const float* scaling,
                                          const float* cc_w, const float* cc_s, int nb_steps,
                                          long long B, int d, int E, int reverse_z, int first, int last,
                                          float* z, float* log_jac_scratch, float* ll, unsigned* row_counters,
                                          void* stream) {
    if (!scaling) return umnn_fail(UMNN_EINVAL, "flow ll forward: scaling must be non-null");
    if (B > 0 && (!ll || !row_counters)) return umnn_fail(UMNN_EINVAL, "flow ll forward: ll and row_counters must be non-null");
    if (z == x) return umnn_fail(UMNN_EINVAL, "flow ll forward: z must not alias x");
    return launch_forward(net, nullptr, x, h, scaling, cc_w, cc_s, nb_steps, B, d, E, 0, nullptr, nullptr, nullptr,
                          z, log_jac_scratch, (hipStream_t)stream, reverse_z != 0, nullptr, ll, row_counters,
                          first != 0, last != 0);
}

extern "C" int umnn_cc_forward_timed(const umnn_mlp* net, const float* x0, const float* x, const float* h,
                                     const float* cc_w, const float* cc_s, int nb_steps,
                                     long long B, int d, int E, float* F, float* f_x, float* f_x0,
                                     int reps, float* ms, void* stream) {
    if (reps < 1 || !ms) return umnn_fail(UMNN_EINVAL, "forward_timed: reps >= 1 and ms non-null");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    if (int rc = umnn_check(hipEventCreate(&e0), "hipEventCreate")) return rc;
    if (int rc = umnn_check(hipEventCreate(&e1), "hipEventCreate")) return rc;
    int rc = umnn_cc_forward(net, x0, x, h, cc_w, cc_s, nb_steps, B, d, E, 0, F, f_x, f_x0, stream);   // warm
    if (!rc) rc = umnn_check(hipEventRecord(e0, st), "hipEventRecord");
    for (int i = 0; i < reps && !rc; ++i)
        rc = umnn_cc_forward(net, x0, x, h, cc_w, cc_s, nb_steps, B, d, E, 0, F, f_x, f_x0, stream);
    if (!rc) rc = umnn_check(hipEventRecord(e1, st), "hipEventRecord");
    if (!rc) rc = umnn_check(hipEventSynchronize(e1), "hipEventSynchronize");
    float t = 0.f;
    if (!rc) rc = umnn_check(hipEventElapsedTime(&t, e0, e1), "hipEventElapsedTime");
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *ms = t / reps;
    return rc;
}
