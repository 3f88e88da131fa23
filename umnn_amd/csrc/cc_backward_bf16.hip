// Backward of the quadrature on the bf16 matrix cores: variant table and launcher of the one-pass kernels (the kernel
// template, with its layout notes, lives in cc_bwd_bf16_kernel.h and is shared with the staged backward of
// cc_backward_front.hip).
#include "cc_bwd_bf16_kernel.h"
#include "cc_bwd_swp_kernel.h"
#include "cc_bwd_ws_kernel.h"
#include "cc_bwd_ws16_kernel.h"

// ------------------------------------------------------------------------------------------
typedef void (*bwd_bf16_kernel_t)(const BwdBf16Args);
struct BwdBf16Variant { int lh, edge, nrl; bwd_bf16_kernel_t fn; const char* name; };
#define BWD_BF16_VARIANT(LHH, E, NR) { LHH, E, NR, cc_bwd_bf16_kernel<LHH, (E) != 0, NR>, "cc_bwd_bf16<L=" #LHH ",EDGE=" #E ",LIVE=" #NR ">" }
// LIVE=13: every hidden layer 48..51 wide (dead registers cost no VALU).  LIVE=0 (all 16 registers): every other net of
// three or four tiles per layer, i.e. hidden widths 32..63 in any combination -- narrower layers are zero-padded to four
// tiles by the staged images (fits the register file since the dW operands became K=16 fragments; 4 spills at L=4).
static const BwdBf16Variant kBwdBf16Variants[] = {
    BWD_BF16_VARIANT(4, 1, 13), BWD_BF16_VARIANT(3, 1, 13), BWD_BF16_VARIANT(2, 1, 13),
    BWD_BF16_VARIANT(4, 1, 0), BWD_BF16_VARIANT(3, 1, 0), BWD_BF16_VARIANT(2, 1, 0),
};

// software-pipelined loop (cc_bwd_swp_kernel.h): same variants, same results, one shared W / W^T image
struct BwdSwpVariant { int lh, nrl; bwd_bf16_kernel_t fn; const char* name; };
#define BWD_SWP_VARIANT(LHH, NR) { LHH, NR, cc_bwd_swp_kernel<LHH, NR>, "cc_bwd_bf16<L=" #LHH ",LIVE=" #NR ",SWP>" }
static const BwdSwpVariant kBwdSwpVariants[] = {
    BWD_SWP_VARIANT(4, 13), BWD_SWP_VARIANT(3, 13), BWD_SWP_VARIANT(2, 13),
    BWD_SWP_VARIANT(4, 0), BWD_SWP_VARIANT(3, 0), BWD_SWP_VARIANT(2, 0),
};

// weight-stationary workgroup pipeline (cc_bwd_ws_kernel.h): four hidden layers, enough tiles to keep every workgroup's pipeline full
struct BwdWsVariant { int nrl; bwd_bf16_kernel_t fn; const char* name; };
static const BwdWsVariant kBwdWsVariants[] = {
    { 13, cc_bwd_ws_kernel<13>, "cc_bwd_bf16<L=4,LIVE=13,WS>" },
    { 0, cc_bwd_ws_kernel<0>, "cc_bwd_bf16<L=4,LIVE=0,WS>" },
};
// the same pipeline on fp16 pieces (cc_bwd_ws16_kernel.h): three-term recompute, scaled cotangents, overflow flag + queued fallback
static const BwdWsVariant kBwdWs16Variants[] = {
    { 13, cc_bwd_ws16_kernel<13>, "cc_bwd_f16<L=4,LIVE=13,WS>" },
    { 0, cc_bwd_ws16_kernel<0>, "cc_bwd_f16<L=4,LIVE=0,WS>" },
};

// ---- the fp16 pipeline as the middle stage of the three-stage backward (cc_backward_front.hip calls these three)
static const BwdWsVariant kBwdWs16FrontVariants[] = {
    { 13, cc_bwd_ws16_kernel<13, true>, "cc_bwd_f16<L=4,LIVE=13,WS,FRONT>" },
    { 0, cc_bwd_ws16_kernel<0, true>, "cc_bwd_f16<L=4,LIVE=0,WS,FRONT>" },
};
static_assert(offsetof(Ws16Scal, flag) == 3 * sizeof(unsigned), "cc_backward_front.hip addresses the flag as scal + 3");
// *name = the variant's kernel name if the launch can use it (nullptr otherwise); the size rule is the caller's
int umnn_ws16_front_eligible(const BwdArgs& base, int nrl, const char** name) {
    *name = nullptr;
    if (base.inv_f || !base.scal || base.m.out_act != UMNN_OUT_ELU_PLUS_ONE || base.n + 1 > W16_MAX_NODES) return 0;
    for (const BwdWsVariant& c : kBwdWs16FrontVariants)
        if (c.nrl == nrl) {
            if (int rc = umnn_allow_lds((const void*)c.fn, (size_t)W16_LDS_USHORTS * sizeof(unsigned short))) return rc;
            *name = c.name;
            break;
        }
    return 0;
}
// the launch scalars (cotangent scale) of a call, once, ahead of its stages
int umnn_ws16_front_prepare(const BwdArgs& base, int nblocks_max, hipStream_t stream) {
    if (int rc = umnn_check(hipMemsetAsync(base.scal, 0, sizeof(Ws16Scal), stream), "memset launch scalars")) return rc;
    const long long want = (base.NI + 1023) / 1024;
    const unsigned nbm = (unsigned)(want < (long long)nblocks_max ? want : (long long)nblocks_max);
    hipLaunchKernelGGL(cc_bwd_cotmax_kernel<>, dim3(nbm), dim3(256), 0, stream, base, reinterpret_cast<Ws16Scal*>(base.scal));
    return 0;
}
int umnn_ws16_front_launch(const BwdBf16Args& mid, int nrl, int nblocks, hipStream_t stream) {
    for (const BwdWsVariant& c : kBwdWs16FrontVariants)
        if (c.nrl == nrl) {
            hipLaunchKernelGGL(c.fn, dim3(nblocks), dim3(64 * WS_WAVES), (size_t)W16_LDS_USHORTS * sizeof(unsigned short), stream, mid);
            return 0;
        }
    return UMNN_EUNSUPPORTED;
}

// Plans and launches the main backward pass with the bf16 kernels.  Returns UMNN_EUNSUPPORTED when the shape is
// outside this family (caller falls back to the fp32 kernels).
int umnn_launch_backward_bf16(const BwdArgs& base, const umnn_mlp* net, int nblocks_max, int* nwaves_out,
                              hipStream_t stream) {
    BwdBf16Args args;
    args.b = base;
    args.z2 = nullptr; args.d2 = nullptr; args.tz2 = nullptr; args.grp0 = 0; args.nl2 = 0; args.accumulate = 0;
    args.scal = nullptr; args.only_if = nullptr;
    BwdArgs& a = args.b;
    const int L = a.m.n_linear - 1;
    if (L < 2 || L - 1 > 3) return UMNN_EUNSUPPORTED;
    int nrl = a.m.ks_in[1], tmax = 0;
    for (int l = 1; l <= L; ++l) {
        if (a.m.t_out[l] > BT) return UMNN_EUNSUPPORTED;
        tmax = a.m.t_out[l] > tmax ? a.m.t_out[l] : tmax;
        if (a.m.ks_in[l] != nrl) nrl = 0;
    }
    if (tmax < 3) return UMNN_EUNSUPPORTED;        // one or two tiles per layer: the fp32 kernels waste less
    if (nrl != 13) nrl = 0;
    a.ngroups = (unsigned)((a.NI + 15) / 16);
    if (a.ns > a.n + 1) a.ns = a.n + 1;
    if (umnn_options().bwd_ws && L == 4) {
        const long long items = (long long)a.ngroups * (a.ns > 1 ? a.ns : 1);
        const BwdWsVariant* wv = nullptr;
        for (const BwdWsVariant& c : kBwdWsVariants)
            if (c.nrl == nrl) { wv = &c; break; }
        if (wv && a.ns <= 1 && items >= 4LL * nblocks_max) {
            const size_t lds_ws = (size_t)WS_LDS_USHORTS * sizeof(unsigned short);
            const int nblocks = nblocks_max;
            *nwaves_out = nblocks;                  // one d_theta slice per workgroup
            a.l_lo = 1;
            if (int rc = umnn_allow_lds((const void*)wv->fn, lds_ws)) return rc;
            // fp16 pieces (default): cotangent scale first, then the pipeline, then the bf16 pipeline queued behind it as the
            // fallback that only runs if a piece overflowed (same outputs, rewritten).  1/f launches, sigmoid outputs and quadratures
            // whose tables do not fit the kernel's LDS copy keep the bf16 pipeline.
            const BwdWsVariant* hv = nullptr;
            // bwd_ws16 = 1 (default): only for launches of at least 2^21 node evaluations (2^22 until the whole GPU suite had passed at 2^21).  Its recompute carries ~3e-7 of relative noise
            // on a pre-activation (fp32: ~1e-7), i.e. 3-4 times as many LeakyReLU kink decisions differ from an exact evaluation; one such
            // decision moves d_theta by ~4e-5 of its largest entry at 4e5 node evaluations (tools/bwd_truth64.py) and by 1 / N of
            // that beyond, so below the threshold the six-term bf16 pipeline keeps mid-size batches inside the 1e-4 parity band.
            // bwd_ws16 = 2: whenever the workgroup pipeline is eligible (tests, measurements).
            const int w16 = umnn_options().bwd_ws16;
            const bool w16_size_ok = w16 == 2 || (w16 == 1 && a.NI * (long long)(a.n + 1) >= (1LL << 21));
            if (w16_size_ok && !a.inv_f && a.scal && a.m.out_act == UMNN_OUT_ELU_PLUS_ONE && a.n + 1 <= W16_MAX_NODES)
                for (const BwdWsVariant& c : kBwdWs16Variants)
                    if (c.nrl == nrl) { hv = &c; break; }
            if (hv) {
                const size_t lds16b = (size_t)W16_LDS_USHORTS * sizeof(unsigned short);
                if (int rc = umnn_allow_lds((const void*)hv->fn, lds16b)) return rc;
                args.scal = a.scal;
                if (int rc = umnn_check(hipMemsetAsync(a.scal, 0, sizeof(Ws16Scal), stream), "memset launch scalars")) return rc;
                umnn_prof_begin(stream);       // (after the memset: its error return must not leave the bracket open)
                const long long want = (a.NI + 1023) / 1024;             // (>= 4 integrals per thread; at most one workgroup per CU)
                const unsigned nbm = (unsigned)(want < (long long)nblocks_max ? want : (long long)nblocks_max);
                hipLaunchKernelGGL(cc_bwd_cotmax_kernel<>, dim3(nbm), dim3(256), 0, stream, a, reinterpret_cast<Ws16Scal*>(a.scal));
#ifdef UMNN_WS_TIMING
                static double* tbuf16 = nullptr;
                const int nw16 = nblocks * WS_WAVES;
                if (!tbuf16) hipMalloc(&tbuf16, sizeof(double) * 6 * 8192);
                hipMemsetAsync(tbuf16, 0, sizeof(double) * 6 * nw16, stream);
                args.tz2 = reinterpret_cast<const float*>(tbuf16);
#endif
                hipLaunchKernelGGL(hv->fn, dim3(nblocks), dim3(64 * WS_WAVES), lds16b, stream, args);
#ifdef UMNN_WS_TIMING
                {
                    hipStreamSynchronize(stream);
                    static double host[6 * 8192];
                    hipMemcpy(host, tbuf16, sizeof(double) * 6 * nw16, hipMemcpyDeviceToHost);
                    const char* role[8] = {"Ca", "F1", "F2", "F3", "Cb", "B1", "B2", "B3"};
                    for (int r = 0; r < WS_WAVES; ++r) {
                        double sm[6] = {0};
                        for (int w = r; w < nw16; w += WS_WAVES) for (int j = 0; j < 6; ++j) sm[j] += host[6 * w + j];
                        const double st = sm[4] > 0 ? sm[4] : 1;
                        fprintf(stderr, "WS16_TIMING %s per step (s_memtime ticks): prep %.0f | work to the %d %% mark %.0f | rest of the work %.0f | barrier wait %.0f   (steps per wave %.0f)\n",
                                role[r], sm[0] / st, UMNN_WS_TRACE_FRAC, sm[1] / st, sm[2] / st, sm[3] / st, sm[4] / (nw16 / WS_WAVES));
                    }
                    args.tz2 = nullptr;
                }
#endif
                args.only_if = &reinterpret_cast<Ws16Scal*>(a.scal)->flag;
                hipLaunchKernelGGL(wv->fn, dim3(nblocks), dim3(64 * WS_WAVES), lds_ws, stream, args);
                umnn_prof_end(stream, 3.0 * umnn_cc_forward_flops_per_integral(net, a.n) * (double)a.NI, UMNN_PROF_BACKWARD);
                umnn_note_launch(hv->name);
                return umnn_check(hipGetLastError(), "cc_bwd_f16 (ws) launch");
            }
#ifdef UMNN_WS_TIMING
            static double* tbuf = nullptr;
            const int nw = nblocks * WS_WAVES;
            if (!tbuf) hipMalloc(&tbuf, sizeof(double) * 6 * 8192);
            hipMemsetAsync(tbuf, 0, sizeof(double) * 6 * nw, stream);
            args.tz2 = reinterpret_cast<const float*>(tbuf);
#endif
            umnn_prof_begin(stream);
            hipLaunchKernelGGL(wv->fn, dim3(nblocks), dim3(64 * WS_WAVES), lds_ws, stream, args);
            umnn_prof_end(stream, 3.0 * umnn_cc_forward_flops_per_integral(net, a.n) * (double)a.NI, UMNN_PROF_BACKWARD);
#ifdef UMNN_WS_TIMING
            {
                hipStreamSynchronize(stream);
                static double host[6 * 8192];
                hipMemcpy(host, tbuf, sizeof(double) * 6 * nw, hipMemcpyDeviceToHost);
                const char* role[8] = {"Ca", "F1", "F2", "F3", "Cb", "B1", "B2", "B3"};
                for (int r = 0; r < WS_WAVES; ++r) {
                    double sm[6] = {0};
                    for (int w = r; w < nw; w += WS_WAVES) for (int j = 0; j < 6; ++j) sm[j] += host[6 * w + j];
                    const double st = sm[4] > 0 ? sm[4] : 1;
                    fprintf(stderr, "WS_TIMING %s per step (s_memtime ticks): prep %.0f | work to the %d %% mark %.0f | rest of the work %.0f | barrier wait %.0f   (steps per wave %.0f)\n",
                            role[r], sm[0] / st, UMNN_WS_TRACE_FRAC, sm[1] / st, sm[2] / st, sm[3] / st, sm[4] / (nw / WS_WAVES));
                }
            }
#endif
            umnn_note_launch(wv->name);
            return umnn_check(hipGetLastError(), "cc_bwd_bf16 (ws) launch");
        }
    }
    if (umnn_options().bwd_swp) {
        const BwdSwpVariant* sv = nullptr;
        for (const BwdSwpVariant& c : kBwdSwpVariants)
            if (c.nrl == nrl && c.lh == L) { sv = &c; break; }
        const size_t lds_swp = ((size_t)(L - 1) * BT * BKS * NPF * FRAG + (size_t)UMNN_WAVES_PER_BLOCK * L * NPB * 16 * TRS) * sizeof(unsigned short);
        if (sv && lds_swp <= 160 * 1024) {
            const long long items = (long long)a.ngroups * (a.ns > 1 ? a.ns : 1);
            int nblocks = nblocks_max;
            if ((long long)nblocks * UMNN_WAVES_PER_BLOCK > items)
                nblocks = (int)((items + UMNN_WAVES_PER_BLOCK - 1) / UMNN_WAVES_PER_BLOCK);
            if (nblocks < 1) nblocks = 1;
            *nwaves_out = nblocks * UMNN_WAVES_PER_BLOCK;
            a.l_lo = 1;
            if (int rc = umnn_allow_lds((const void*)sv->fn, lds_swp)) return rc;
            umnn_prof_begin(stream);
            hipLaunchKernelGGL(sv->fn, dim3(nblocks), dim3(UMNN_BLOCK), lds_swp, stream, args);
            umnn_prof_end(stream, 3.0 * umnn_cc_forward_flops_per_integral(net, a.n) * (double)a.NI, UMNN_PROF_BACKWARD);
            umnn_note_launch(sv->name);
            return umnn_check(hipGetLastError(), "cc_bwd_bf16 (swp) launch");
        }
    }
    int off16 = 0;
    for (int l = 1; l < L; ++l) { args.off_fwd[l] = off16; off16 += BT * BKS * NPF * FRAG; }
    for (int l = 1; l < L; ++l) { args.off_tr[l] = off16; off16 += BT * BKS * NPB * FRAG; }
    args.off_trtile = off16;
    off16 += UMNN_WAVES_PER_BLOCK * NPB * 16 * TRS;
    const size_t lds_bytes = (size_t)off16 * sizeof(unsigned short);
    if (lds_bytes > 160 * 1024) return UMNN_EUNSUPPORTED;
    const long long items = (long long)a.ngroups * (a.ns > 1 ? a.ns : 1);
    int nblocks = nblocks_max;
    if ((long long)nblocks * UMNN_WAVES_PER_BLOCK > items)
        nblocks = (int)((items + UMNN_WAVES_PER_BLOCK - 1) / UMNN_WAVES_PER_BLOCK);
    if (nblocks < 1) nblocks = 1;
    *nwaves_out = nblocks * UMNN_WAVES_PER_BLOCK;
    const BwdBf16Variant* v = nullptr;
    for (const BwdBf16Variant& c : kBwdBf16Variants)
        if (c.nrl == nrl && c.lh == L) { v = &c; break; }
    if (!v) return UMNN_EUNSUPPORTED;
    a.l_lo = 1;
    if (int rc = umnn_allow_lds((const void*)v->fn, lds_bytes)) return rc;
    umnn_prof_begin(stream);
    hipLaunchKernelGGL(v->fn, dim3(nblocks), dim3(UMNN_BLOCK), lds_bytes, stream, args);
    umnn_prof_end(stream, 3.0 * umnn_cc_forward_flops_per_integral(net, a.n) * (double)a.NI, UMNN_PROF_BACKWARD);
    umnn_note_launch(v->name);
    return umnn_check(hipGetLastError(), "cc_bwd_bf16 launch");
}
