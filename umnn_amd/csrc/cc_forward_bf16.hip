// Forward quadrature on the bf16 matrix cores: variant table and launcher (the kernel template, with its layout notes,
// lives in cc_fwd_bf16_kernel.h and is shared with the inversion variants of cc_invert.hip).
#ifndef UMNN_ASM_TIED
#define UMNN_ASM_TIED 1      // cc_common.h: inline-assembly outputs tied to inputs in the forward translation units
#endif
#include "cc_fwd_bf16_kernel.h"
using namespace UMNN_FWD_NS;
// (this file is compiled twice: as is -- bf16 pieces, umnn_launch_forward_bf16 -- and through cc_forward_f16.hip with
// -DUMNN_FWD_PIECE_F16 -- fp16 pieces, umnn_launch_forward_f16, kernel names cc_fwd_f16<...>)
#ifdef UMNN_FWD_PIECE_F16
#define FWD_KNAME "cc_fwd_f16"
#define FWD_LAUNCH umnn_launch_forward_f16
#else
#define FWD_KNAME "cc_fwd_bf16"
#define FWD_LAUNCH umnn_launch_forward_bf16
#endif

// ------------------------------------------------------------------------------------------
typedef void (*fwd_bf16_kernel_t)(const FwdBf16Args);
struct Bf16Variant { int tmax, nparts, p, exact, nrl, pipe, wpb; fwd_bf16_kernel_t fn; const char* name; };
// first hidden layer T1 = 5..8 tiles, every other hidden layer at most four (zero-padded to four): MNISTExperiment's
// 31-100-50-50-50-50-1.  Shape-exact: layer 1's GEMM contracts over T1 tiles, the others over four.
// LIVE=13: every later layer 48..51 wide -- 13 live registers per lane and the merged five-K-step layout from layer 2 on.
struct Bf16WideFirst { int t1, nrl, p; fwd_bf16_kernel_t fn; const char* name; };
#define BF16_WIDE_FIRST(T, NR, PP) { T, NR, PP, cc_fwd_bf16_kernel<T, 2, PP, true, NR, false, false, 4>, FWD_KNAME "<T1=" #T ",TREST=4,PARTS=2,P=" #PP ",EXACT=1,LIVE=" #NR ">" }
static const Bf16WideFirst kBf16WideFirst[] = { BF16_WIDE_FIRST(5, 13, 1), BF16_WIDE_FIRST(6, 13, 1), BF16_WIDE_FIRST(7, 13, 1), BF16_WIDE_FIRST(8, 13, 1),
                                                BF16_WIDE_FIRST(5, 0, 1), BF16_WIDE_FIRST(6, 0, 1), BF16_WIDE_FIRST(7, 0, 1), BF16_WIDE_FIRST(8, 0, 1) };
// (two point tiles per wave, P = 2: 260 registers = one wave per SIMD, 0.63 ms against 0.49 ms at the MNIST shape -- not instantiated)
#define BF16_VARIANT(T, NP, PP, EX, NR) { T, NP, PP, EX, NR, 0, 4, cc_fwd_bf16_kernel<T, NP, PP, (EX) != 0, NR>, FWD_KNAME "<T=" #T ",PARTS=" #NP ",P=" #PP ",EXACT=" #EX ",LIVE=" #NR ">" }
#define BF16_PIPE_VARIANT(NP, PP, NR) { 4, NP, PP, 1, NR, 1, 4, cc_fwd_bf16_kernel<4, NP, PP, true, NR, true>, FWD_KNAME "<T=4,PARTS=" #NP ",P=" #PP ",EXACT=1,LIVE=" #NR ",PIPE>" }
// eight waves per workgroup (one workgroup per CU: see WPB in cc_fwd_bf16_kernel.h)
#define BF16_VARIANT_W8(T, NR) { T, 2, 1, 1, NR, 0, 8, cc_fwd_bf16_kernel<T, 2, 1, true, NR, false, false, 0, 8>, FWD_KNAME "<T=" #T ",PARTS=2,P=1,EXACT=1,LIVE=" #NR ",WAVES=8>" }
static const Bf16Variant kBf16Variants[] = {
    BF16_PIPE_VARIANT(2, 2, 13), BF16_PIPE_VARIANT(2, 2, 0),   // software-pipelined node loop (>= 2 hidden layers, bf16x3)
    BF16_VARIANT(4, 2, 1, 1, 13), BF16_VARIANT(4, 2, 2, 1, 13),   // widths 48..51
    BF16_VARIANT(4, 2, 1, 1, 0), BF16_VARIANT(4, 2, 2, 1, 0),     // widths 52..62
    BF16_VARIANT(2, 2, 1, 0, 0), BF16_VARIANT(2, 2, 2, 0, 0),
    BF16_VARIANT(4, 2, 1, 0, 0), BF16_VARIANT(4, 2, 2, 0, 0),
#ifndef UMNN_FWD_PIECE_F16          // three pieces / six cross terms (bf16x6): pointless on fp16 pieces, whose two already carry 22 bits
    BF16_VARIANT(4, 3, 1, 1, 13), BF16_VARIANT(4, 3, 2, 1, 13), BF16_VARIANT(4, 3, 1, 1, 0), BF16_VARIANT(4, 3, 2, 1, 0),
    BF16_VARIANT(2, 3, 1, 0, 0), BF16_VARIANT(2, 3, 2, 0, 0), BF16_VARIANT(4, 3, 1, 0, 0), BF16_VARIANT(4, 3, 2, 0, 0),
#endif
    BF16_VARIANT(7, 2, 1, 1, 26), BF16_VARIANT(7, 2, 1, 1, 0),   // widths 96..111 (100-wide toy / MonotonicNN nets): 3 K-steps + a half one
    BF16_VARIANT(5, 2, 1, 1, 0), BF16_VARIANT(6, 2, 1, 1, 0), BF16_VARIANT(8, 2, 1, 1, 0),   // widths 64..79, 80..95, 112..127
    BF16_VARIANT_W8(7, 26), BF16_VARIANT_W8(7, 0), BF16_VARIANT_W8(5, 0), BF16_VARIANT_W8(6, 0), BF16_VARIANT_W8(8, 0),
    BF16_VARIANT(8, 2, 1, 0, 0), BF16_VARIANT(8, 2, 2, 0, 0),   // (8 tiles x 3 parts does not fit the register file: fp32 kernels instead)
};

int umnn_launch_forward_p32(FwdArgs& a, const umnn_mlp* net, int nb_steps, hipStream_t stream);      // cc_forward_p32.hip

// ---- overflow protocol of the fp16-piece forward (the library default) -------------------------------------------------------------
// An fp16 piece overflows at 65520; the value it belongs to then reaches the quadrature sum of its integral as inf / NaN (every
// product with an inf piece is inf or NaN and nothing downstream can make it finite again).  The fp16 build therefore checks that sum
// once per tile group, in the epilogue: a group with a non-finite sum writes NOTHING but a NaN marker into its output slots, raises
// the launch's flag word, and leaves its integrals to the bf16 build of the same kernel (bf16 pieces share fp32's exponent range),
// which this launcher queues right behind the fp16 launch with the same group mapping (P, NS forced).  That second launch costs one
// scalar load per workgroup when the flag is down (every benchmarked net); when it is up it recomputes exactly the marked groups --
// on two bf16 pieces, i.e. those integrals come back at bf16x3 accuracy (~6e-6) instead of ~5e-7 -- and publishes them, the
// one-pass log-likelihood rows included (a deferred group has not counted towards its rows yet).  NaN inputs take the same road and come
// back NaN.  Flag words: a per-device ring of 64-bit words, launch generation g uses word g % 256, raised with atomicMax(word, g),
// tested as word >= g -- no memset between launches, safe across streams, and a replayed hipGraph (g baked into its nodes) at worst
// runs a fallback that finds no marked group.
struct FwdOvfPlan { int mode; unsigned long long* flag; unsigned long long gen; };
int umnn_ovf_slot(unsigned long long** flag, unsigned long long* gen);                                  // cc_api.hip
#ifdef UMNN_FWD_PIECE_F16
int umnn_launch_forward_bf16(FwdArgs& a, const umnn_mlp* net, int nparts, int P, int ns, int nb_steps, hipStream_t stream,
                             const FwdOvfPlan* ovf);
#endif

// The planned launch itself.  fp16 build: the launch with ovf_mode = 1, then the bf16 build of the same plan queued as its fallback
// (one profiling bracket around both, one launch note).  bf16 build: a plain launch, or (ovf) that queued fallback.
static int launch_planned(fwd_bf16_kernel_t fn, const char* name, unsigned nblk, size_t lds_bytes, FwdBf16Args& args, FwdArgs& a,
                          const umnn_mlp* net, int P, int ns, int nb_steps, hipStream_t stream, const FwdOvfPlan* ovf, int block) {
    args.f.ovf_mode = ovf ? ovf->mode : 0;
    args.f.ovf_flag = ovf ? ovf->flag : nullptr;
    args.f.ovf_gen = ovf ? ovf->gen : 0;
#ifdef UMNN_FWD_PIECE_F16
    umnn_prof_begin(stream);
    hipLaunchKernelGGL(fn, dim3(nblk), dim3(block), lds_bytes, stream, args);
    const FwdOvfPlan second{2, ovf->flag, ovf->gen};
    int rc = umnn_check(hipGetLastError(), "cc_fwd_f16 launch");
    if (!rc) rc = umnn_launch_forward_bf16(a, net, 2, P, ns, nb_steps, stream, &second);
    if (rc == UMNN_EUNSUPPORTED) rc = umnn_fail(UMNN_EUNSUPPORTED, "cc_fwd_f16: the bf16 build has no kernel for the plan of the fp16 launch");
    umnn_prof_end(stream, umnn_cc_forward_flops_per_integral(net, nb_steps) * (double)a.NI);
    umnn_note_launch(name);
    return rc;
#else
    (void)P; (void)ns;
    if (!ovf) umnn_prof_begin(stream);
    hipLaunchKernelGGL(fn, dim3(nblk), dim3(block), lds_bytes, stream, args);
    if (!ovf) {
        umnn_prof_end(stream, umnn_cc_forward_flops_per_integral(net, nb_steps) * (double)a.NI);
        umnn_note_launch(name);
    }
    return umnn_check(hipGetLastError(), "cc_fwd_bf16 launch");
#endif
}

// Returns 0 and launches, UMNN_EUNSUPPORTED (without setting the error text's prefix) if the shape does not fit
// this kernel family (caller then uses the fp32-MFMA kernels), or another error code.
// ovf (bf16 build only): non-null = this launch is the queued fallback of an fp16-piece launch -- P and ns are that launch's (forced),
// no profiling bracket, no launch note.
int FWD_LAUNCH(FwdArgs& a, const umnn_mlp* net, int nparts, int P, int ns, int nb_steps,
                             hipStream_t stream, const FwdOvfPlan* ovf) {
    const int L = a.m.n_linear - 1;
    const UmnnOptions& opt = umnn_options();
#ifdef UMNN_FWD_PIECE_F16
    FwdOvfPlan own{1, nullptr, 0};
    if (int rc = umnn_ovf_slot(&own.flag, &own.gen)) return rc;
    ovf = &own;
    const bool fb = false;
#else
    const bool fb = ovf != nullptr;
#endif
    const bool p_forced = opt.fwd_p > 0 || fb, ns_forced = opt.fwd_ns > 0 || fb;
    int tmax = 0;
    for (int l = 1; l <= L; ++l) tmax = a.m.t_out[l] > tmax ? a.m.t_out[l] : tmax;
    int T = tmax <= 2 ? 2 : tmax <= 4 ? 4 : 8;
    // every hidden layer the same tile count above four: exact single-tile variants, odd counts with a half K-step
    int wide = (nparts == 2 && tmax >= 5) ? tmax : 0;
    for (int l = 1; l <= L && wide; ++l) if (a.m.t_out[l] != wide) wide = 0;
    if (wide && (P == 1 || !p_forced)) { T = wide; P = 1; } else wide = 0;    // (no two-tile variant at these widths)
    // uniform wide nets: how many workgroups of images fit a CU?  One -> eight waves per workgroup (both waves of every SIMD share the
    // images); two -> four waves each.  Either way a CU runs eight waves.
    int wpb = UMNN_WAVES_PER_BLOCK;
    if (wide) {
        const size_t img = (size_t)(L - 1) * wide * ((wide / 2) * nparts * 512 + (wide & 1) * nparts * 256) * sizeof(unsigned short);
        if (2 * (img + 1024) > 160 * 1024) wpb = 8;
    }
    if (wide && !ns_forced) {
        // split the node range only as far as that fills the SIMDs (two waves each)
        const long long tiles16 = (a.NI + 15) / 16, slots = (long long)umnn_num_cus() * 8;
        ns = tiles16 * 4 <= slots ? 4 : tiles16 * 2 <= slots ? 2 : 1;
        if (ns > nb_steps + 1) ns = 1;
    }
    FwdBf16Args args;
    args.f = a;
    // ---- wide first hidden layer over a narrow rest: its own shape-exact family
    {
        bool wf = nparts == 2 && L >= 2 && a.m.t_out[1] >= 5 && a.m.t_out[1] <= 8 && (P == 1 || !p_forced);
        for (int l = 2; l <= L && wf; ++l) if (a.m.t_out[l] > 4) wf = false;
        if (wf) {
            const int T1 = a.m.t_out[1];
            int o16 = 0;
            for (int l = 1; l <= L; ++l) {
                args.pl.ks32[l] = l == 1 ? T1 / 2 : 2;
                args.pl.half_in[l] = l == 1 ? (T1 & 1) : 0;
                if (l >= 2) args.f.m.t_out[l] = 4;
            }
            for (int l = 1; l < L; ++l) {
                args.pl.off16[l] = o16;
                o16 += 4 * (args.pl.ks32[l] * 2 * 512 + args.pl.half_in[l] * 2 * 256);
            }
            args.f.m.lds_off[L] = (((o16 + 1) / 2) + 3) & ~3;
            if (!ns_forced) {
                const long long tiles16 = (a.NI + 15) / 16, slots = (long long)umnn_num_cus() * 4 * 2;
                ns = tiles16 * 4 <= slots ? 4 : tiles16 * 2 <= slots ? 2 : 1;
                if (ns > nb_steps + 1) ns = 1;
            }
            const int PW = 1;
            const size_t lds_bytes = ((size_t)args.f.m.lds_off[L] + (ns > 1 ? UMNN_WAVES_PER_BLOCK * 3 * PW * 16 : 0)) * sizeof(float);
            int nrest = a.m.ks_in[2];           // live registers of the later layers when they all agree (13 = widths 48..51)
            for (int l = 2; l <= L; ++l) if (a.m.ks_in[l] != nrest) nrest = 0;
            if (nrest != 13) nrest = 0;
            const Bf16WideFirst* pick = nullptr;
            for (const Bf16WideFirst& v : kBf16WideFirst) if (v.t1 == T1 && v.nrl == nrest && v.p == PW) pick = &v;
            if (!pick) for (const Bf16WideFirst& v : kBf16WideFirst) if (v.t1 == T1 && v.nrl == nrest && v.p == 1) pick = &v;
            if (pick && lds_bytes <= 160 * 1024) {
                if (int rc = umnn_allow_lds((const void*)pick->fn, lds_bytes)) return rc;
                args.f.ns = ns;
                args.f.ngroups = (unsigned)((a.NI + 16 * pick->p - 1) / (16 * pick->p));
                const unsigned gpb = UMNN_WAVES_PER_BLOCK / ns;
                const unsigned nblk = (args.f.ngroups + gpb - 1) / gpb;
                return launch_planned(pick->fn, pick->name, nblk, lds_bytes, args, a, net, pick->p, ns, nb_steps, stream, ovf, UMNN_BLOCK);
            }
            args.f = a;         // (not launched: fall through to the generic plan)
        }
    }
    if (a.z2_save) return UMNN_EUNSUPPORTED;      // (only the wide-first kernels above know how to leave z_2 behind)
    int off16 = 0;
    for (int l = 1; l <= L; ++l) {
        args.pl.half_in[l] = wide ? (wide & 1) : 0;
        args.pl.ks32[l] = wide ? wide / 2 : (a.m.t_out[l] + 1) / 2;
    }
    for (int l = 1; l < L; ++l) {
        args.pl.off16[l] = off16;
        off16 += a.m.t_out[l + 1] * (args.pl.ks32[l] * nparts * 512 + args.pl.half_in[l] * nparts * 256);
    }
    const int img_floats = (off16 + 1) / 2;
    args.f.m.lds_off[L] = (img_floats + 3) & ~3;          // NS-reduction scratch starts after the images
    int exact = 1, nrl = a.m.ks_in[1];        // live registers per lane when every hidden layer has the same K-step count
    for (int l = 1; l <= L; ++l) {
        exact = exact && a.m.t_out[l] == T;
        if (a.m.ks_in[l] != nrl) nrl = 0;
    }
    // Mixed or narrow widths up to 63: zero-pad every layer to four tiles (the staged images carry the zeros) and run
    // the shape-exact kernels -- the padded MFMAs cost less than the runtime guards of the generic variants
    // (UMNN_FWD_PAD=0 keeps the generic ones).
    if (!exact && !wide && nparts == 2 && tmax <= 4 && tmax >= opt.fwd_pad_min && opt.fwd_pad) {
        T = 4; exact = 1; nrl = 0;
        for (int l = 1; l <= L; ++l) { args.f.m.t_out[l] = 4; args.pl.ks32[l] = 2; }
        off16 = 0;
        for (int l = 1; l < L; ++l) { args.pl.off16[l] = off16; off16 += 4 * 2 * nparts * 512; }
        args.f.m.lds_off[L] = (((off16 + 1) / 2) + 3) & ~3;
    }
    const bool want_pipe = opt.fwd_pipe != 0 && L >= 2;
    // the 32x32x16 formulation of the flagship shape: opt-in (UMNN_FWD_PIPE=2 / option fwd_pipe = 2).  Same wall time as the
    // default at C3 with 10 % fewer cycles -- the chip clocks it lower (the kernel is power-bound)
#ifndef UMNN_FWD_PIECE_F16
    if (opt.fwd_pipe == 2 && !fb && exact && T == 4 && nparts == 2 && nrl == 13) {
        const int rc = umnn_launch_forward_p32(a, net, nb_steps, stream);
        if (rc != UMNN_EUNSUPPORTED) return rc;
    }
#endif
    // the pipelined loop needs two point tiles per wave and pays off as soon as that still leaves a wave per SIMD
    // (measured at the POWER and VAE shapes: P=2, NS=1 beats every P=1 split by 6-7 %)
    if (want_pipe && exact && T == 4 && nparts == 2 && !p_forced && !ns_forced &&
        (a.NI + 15) / 16 >= 2LL * umnn_num_cus() * 4) { P = 2; ns = 1; }
    if (!(wide && exact)) wpb = UMNN_WAVES_PER_BLOCK;
    size_t lds_bytes = ((size_t)args.f.m.lds_off[L] + (ns > 1 ? wpb * 3 * P * 16 : 0)) * sizeof(float);
    if (lds_bytes > 160 * 1024 && wpb == 8) { wpb = UMNN_WAVES_PER_BLOCK; lds_bytes = ((size_t)args.f.m.lds_off[L] + (ns > 1 ? wpb * 3 * P * 16 : 0)) * sizeof(float); }
    if (lds_bytes > 160 * 1024) return UMNN_EUNSUPPORTED;
    const Bf16Variant* pick = nullptr;
    for (int ex = exact; ex >= 0 && !pick; --ex)
        for (int pass = 0; pass < 2 && !pick; ++pass)      // pass 0: a variant with exactly this live-register count
            for (const Bf16Variant& v : kBf16Variants)
                if (v.tmax == T && v.nparts == nparts && v.p == P && v.exact == ex && (!v.pipe || want_pipe) && v.wpb == (ex ? wpb : 4) &&
                    (pass == 0 ? (ex && nrl && v.nrl == nrl) : v.nrl == 0)) { pick = &v; break; }
    if (!pick) return UMNN_EUNSUPPORTED;
    fwd_bf16_kernel_t kfn = pick->fn;
    const char* kname = pick->name;
    if (int rc = umnn_allow_lds((const void*)kfn, lds_bytes)) return rc;
    args.f.ns = ns;
    args.f.ngroups = (unsigned)((a.NI + 16 * P - 1) / (16 * P));
    const unsigned gpb = pick->wpb / ns;
    const unsigned nblk = (args.f.ngroups + gpb - 1) / gpb;
    return launch_planned(kfn, kname, nblk, lds_bytes, args, a, net, P, ns, nb_steps, stream, ovf, 64 * pick->wpb);
}
