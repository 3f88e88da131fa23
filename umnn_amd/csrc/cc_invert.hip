// Sampling direction of a UMNNMAF block on the matrix cores: the reference's dimension-by-dimension bracket search
// (UMNNMAF.invert, models/UMNN/UMNNMAF.py:182-232, driven by UMNNMAFFlow.invert, UMNNMAFFlow.py:78-90) with the WHOLE search of
// one dimension -- `iter` rounds x 10 candidate integrals per sample -- inside one launch.  The reference issues, per
// dimension and round, a ParallelNeuralIntegral over [10*B, 1] rows (plus a MADE pass per dimension); here one dimension is
// the MADE pass + ONE launch of an INV variant of cc_fwd_bf16_kernel (cc_fwd_bf16_kernel.h): tile = sample, lane p = candidate
// p, the hoisted first-layer term computed once per sample, the argmin / new bracket a 16-lane butterfly between rounds.
// Arithmetic: bf16x3 split products by default (F to ~6e-6 relative), orders of magnitude inside the search's own resolution
// (100 * (2/9)^iter).  Under fwd_precision = fp32 / bf16x6 ("the reference's arithmetic everywhere") nets of up to four tiles per
// layer run the same search with THREE bf16 pieces and six cross terms (PARTS=3: fp32-level products, ~4e-7 on F -- the
// matrix-core arithmetic of the bf16x6 forward mode; there is no fp32-MFMA form of this kernel); wider nets keep the host-driven
// search on the forward kernels of that mode (8 tiles x 3 pieces do not fit the register file).
// This file is compiled twice, like cc_forward_bf16.hip: as is (bf16 pieces; exports umnn_flow_invert_dim) and through cc_invert_f16.hip
// with -DUMNN_FWD_PIECE_F16 (fp16 pieces: the search of the library's default arithmetic, f16x3).  The fp16 build follows the
// forward's overflow protocol (cc_forward_bf16.hip): a sample for which any candidate integral of any round was not finite gets a NaN
// in its slot of x_inv[:, j] and raises the launch's flag word; the two-piece bf16 build of the same search, queued right behind it,
// returns at once when the flag is down and otherwise redoes exactly the samples whose slot holds the NaN.
#ifndef UMNN_ASM_TIED
#define UMNN_ASM_TIED 1      // cc_common.h: inline-assembly outputs tied to inputs in the forward translation units
#endif
#include "cc_fwd_bf16_kernel.h"
using namespace UMNN_FWD_NS;
#ifdef UMNN_FWD_PIECE_F16
#define INV_KNAME "cc_invert_f16"
#define INV_IMPL umnn_invert_impl_f16
#else
#define INV_KNAME "cc_invert_bf16"
#define INV_IMPL umnn_invert_impl_bf16
#endif
struct InvOvfPlan { int mode; unsigned long long* flag; unsigned long long gen; };
int umnn_ovf_slot(unsigned long long** flag, unsigned long long* gen);                                  // cc_api.hip
int umnn_invert_impl_bf16(const umnn_mlp* net, const float* h, const float* z, const float* scaling, const float* cc_w, const float* cc_s,
                          int nb_steps, long long B, int d, int E, int j, int iters, float* x_inv, hipStream_t stream, int nparts,
                          const InvOvfPlan* ovf);
int umnn_invert_impl_f16(const umnn_mlp* net, const float* h, const float* z, const float* scaling, const float* cc_w, const float* cc_s,
                         int nb_steps, long long B, int d, int E, int j, int iters, float* x_inv, hipStream_t stream, int nparts,
                         const InvOvfPlan* ovf);

typedef void (*inv_kernel_t)(const FwdBf16Args);
struct InvVariant { int tmax, exact, nrl, nparts, wpb; inv_kernel_t fn; const char* name; };
#define INV_VARIANT(T, EX, NR) { T, EX, NR, 2, 4, cc_fwd_bf16_kernel<T, 2, 1, (EX) != 0, NR, false, true>, INV_KNAME "<T=" #T ",EXACT=" #EX ",LIVE=" #NR ">" }
// (eight waves per workgroup: images that leave room for one workgroup per CU -- WPB in cc_fwd_bf16_kernel.h)
#define INV_VARIANT_W8(T, NR) { T, 1, NR, 2, 8, cc_fwd_bf16_kernel<T, 2, 1, true, NR, false, true, 0, 8>, INV_KNAME "<T=" #T ",EXACT=1,LIVE=" #NR ",WAVES=8>" }
#define INV_VARIANT3(T, EX, NR) { T, EX, NR, 3, 4, cc_fwd_bf16_kernel<T, 3, 1, (EX) != 0, NR, false, true>, INV_KNAME "<T=" #T ",PARTS=3,EXACT=" #EX ",LIVE=" #NR ">" }
// wide first hidden layer over a narrow rest (MNISTExperiment's integrand: sampling d = 784 images is 3 920 of these launches)
struct InvWideFirst { int t1, nrl; inv_kernel_t fn; const char* name; };
#define INV_WIDE_FIRST(T, NR) { T, NR, cc_fwd_bf16_kernel<T, 2, 1, true, NR, false, true, 4>, INV_KNAME "<T1=" #T ",TREST=4,LIVE=" #NR ">" }
static const InvWideFirst kInvWideFirst[] = { INV_WIDE_FIRST(5, 13), INV_WIDE_FIRST(6, 13), INV_WIDE_FIRST(7, 13), INV_WIDE_FIRST(8, 13),
                                              INV_WIDE_FIRST(5, 0), INV_WIDE_FIRST(6, 0), INV_WIDE_FIRST(7, 0), INV_WIDE_FIRST(8, 0) };
static const InvVariant kInvVariants[] = {
    INV_VARIANT(4, 1, 13), INV_VARIANT(4, 1, 0),       // UCI / VAE nets (31-50^4-1) and every other 3..4-tile net (zero-padded)
    INV_VARIANT(7, 1, 26), INV_VARIANT(7, 1, 0),       // 100-wide toy nets
    INV_VARIANT(5, 1, 0), INV_VARIANT(6, 1, 0), INV_VARIANT(8, 1, 0),
    INV_VARIANT_W8(7, 26), INV_VARIANT_W8(7, 0), INV_VARIANT_W8(5, 0), INV_VARIANT_W8(6, 0), INV_VARIANT_W8(8, 0),
    INV_VARIANT(2, 0, 0), INV_VARIANT(4, 0, 0), INV_VARIANT(8, 0, 0),   // generic (runtime tile counts): mixed widths, e.g. 100-50-50-50-50
#ifndef UMNN_FWD_PIECE_F16
    // three pieces / six cross terms (fwd_precision = fp32 | bf16x6): nets of up to four tiles per layer
    INV_VARIANT3(4, 1, 13), INV_VARIANT3(4, 1, 0), INV_VARIANT3(2, 0, 0), INV_VARIANT3(4, 0, 0),
#endif
};

// One launch of the search (see the file header); ovf: bf16 build only -- non-null = the queued fallback of an fp16-piece launch.
int INV_IMPL(const umnn_mlp* net, const float* h, const float* z, const float* scaling, const float* cc_w, const float* cc_s,
             int nb_steps, long long B, int d, int E, int j, int iters, float* x_inv, hipStream_t stream, int nparts,
             const InvOvfPlan* ovf) {
    FwdBf16Args args;
    FwdArgs& a = args.f;
    int tmax = 0, ksu = 0;
    if (int rc = umnn_prepare_mlp(net, E, &a.m, &tmax, &ksu)) return rc;
    const int L = a.m.n_linear - 1;
#ifdef UMNN_FWD_PIECE_F16
    InvOvfPlan own{1, nullptr, 0};
    if (int rc = umnn_ovf_slot(&own.flag, &own.gen)) return rc;
    ovf = &own;
#endif
    a.ovf_mode = ovf ? ovf->mode : 0; a.ovf_flag = ovf ? ovf->flag : nullptr; a.ovf_gen = ovf ? ovf->gen : 0;
    // the planned launch: fp16 build = the launch, then the two-piece bf16 build of the same search queued as its fallback
    auto launch = [&](inv_kernel_t fn, const char* name, unsigned nblk, size_t lds_bytes, int block = UMNN_BLOCK) -> int {
        const bool queued = ovf && ovf->mode == 2;
        if (!queued) umnn_prof_begin(stream);
        hipLaunchKernelGGL(fn, dim3(nblk), dim3(block), lds_bytes, stream, args);
        int rc = umnn_check(hipGetLastError(), "cc_invert launch");
#ifdef UMNN_FWD_PIECE_F16
        const InvOvfPlan second{2, ovf->flag, ovf->gen};
        if (!rc) rc = umnn_invert_impl_bf16(net, h, z, scaling, cc_w, cc_s, nb_steps, B, d, E, j, iters, x_inv, stream, 2, &second);
#endif
        if (!queued) {
            // algorithmic work: iters rounds x 10 candidate integrals per sample
            umnn_prof_end(stream, umnn_cc_forward_flops_per_integral(net, nb_steps) * 10.0 * iters * (double)B);
            umnn_note_launch(name);
        }
        return rc;
    };
    // Small batches: one sample per WORKGROUP, its node range split over all the workgroup's waves (partials meet in LDS once per
    // round) -- B waves of a 100-image sampling call leave nine SIMDs in ten idle, and every wave walks 10 rounds x (n + 1) nodes
    // alone.  Taken while all B x wpb waves are resident at once (two per SIMD); the queued bf16 build gets the same plan.
    auto split_over = [&](int wpb) -> int {
        return (B * (long long)wpb <= (long long)umnn_num_cus() * 8 && wpb <= nb_steps + 1) ? wpb : 1;
    };
    a.x0 = nullptr; a.x = nullptr; a.h = h; a.ccw = cc_w; a.ccs = cc_s;
    a.F = a.fx = a.fx0 = nullptr; a.scaling = scaling; a.z = nullptr; a.logjac = nullptr; a.logjac_in = nullptr;
    a.reverse_z = 0; a.ll = nullptr; a.row_cnt = nullptr; a.ll_first = a.ll_last = 0;
    a.inv_z = z; a.inv_x = x_inv; a.inv_j = j; a.inv_iters = iters;
    a.NI = B; a.d = d; a.E = E; a.n = nb_steps; a.inv_f = 0; a.ns = 1; a.x_bf16 = 0; a.h_bf16 = 0; a.z2_save = nullptr; a.z2_nl2 = 0;

    // ---- wide first hidden layer, every other layer at most four tiles: shape-exact family (as in cc_forward_bf16.hip)
    {
        bool wf = a.m.t_out[1] >= 5 && a.m.t_out[1] <= 8;
        for (int l = 2; l <= L && wf; ++l) if (a.m.t_out[l] > 4) wf = false;
        if (wf) {
            const int T1 = a.m.t_out[1];
            int o16 = 0;
            for (int l = 1; l <= L; ++l) {
                args.pl.ks32[l] = l == 1 ? T1 / 2 : 2;
                args.pl.half_in[l] = l == 1 ? (T1 & 1) : 0;
                if (l >= 2) a.m.t_out[l] = 4;
            }
            for (int l = 1; l < L; ++l) {
                args.pl.off16[l] = o16;
                o16 += 4 * (args.pl.ks32[l] * 2 * 512 + args.pl.half_in[l] * 2 * 256);
            }
            a.m.lds_off[L] = (((o16 + 1) / 2) + 3) & ~3;
            a.ns = split_over(UMNN_WAVES_PER_BLOCK);
            const size_t lds_bytes = ((size_t)a.m.lds_off[L] + (a.ns > 1 ? UMNN_WAVES_PER_BLOCK * 16 : 0)) * sizeof(float);
            int nrest = a.m.ks_in[2];           // live registers of the later layers when they all agree (13 = widths 48..51)
            for (int l = 2; l <= L; ++l) if (a.m.ks_in[l] != nrest) nrest = 0;
            if (nrest != 13) nrest = 0;
            const InvWideFirst* pick = nullptr;
            for (const InvWideFirst& v : kInvWideFirst) if (v.t1 == T1 && v.nrl == nrest) pick = &v;
            if (pick && lds_bytes <= 160 * 1024) {
                if (int rc = umnn_allow_lds((const void*)pick->fn, lds_bytes)) return rc;
                a.ngroups = (unsigned)B;
                const unsigned gpb = UMNN_WAVES_PER_BLOCK / a.ns;
                const unsigned nblk = (a.ngroups + gpb - 1) / gpb;
                return launch(pick->fn, pick->name, nblk, lds_bytes);
            }
            return umnn_fail(UMNN_EUNSUPPORTED, "invert: weight images exceed 160 KiB of LDS");
        }
    }
    // ---- plan (the P = 1, two-piece subset of umnn_launch_forward_bf16's) ----
    int T = tmax <= 2 ? 2 : tmax <= 4 ? 4 : 8;
    int wide = tmax >= 5 ? tmax : 0;
    for (int l = 1; l <= L && wide; ++l) if (a.m.t_out[l] != wide) wide = 0;
    if (wide) T = wide;
    int off16 = 0;
    for (int l = 1; l <= L; ++l) {
        args.pl.half_in[l] = wide ? (wide & 1) : 0;
        args.pl.ks32[l] = wide ? wide / 2 : (a.m.t_out[l] + 1) / 2;
    }
    for (int l = 1; l < L; ++l) {
        args.pl.off16[l] = off16;
        off16 += a.m.t_out[l + 1] * (args.pl.ks32[l] * nparts * 512 + args.pl.half_in[l] * nparts * 256);
    }
    int exact = 1, nrl = a.m.ks_in[1];
    for (int l = 1; l <= L; ++l) {
        exact = exact && a.m.t_out[l] == T;
        if (a.m.ks_in[l] != nrl) nrl = 0;
    }
    if (!exact && !wide && tmax <= 4 && tmax >= 3) {       // mixed 3..4-tile nets: zero-pad to the shape-exact kernel
        T = 4; exact = 1; nrl = 0;
        for (int l = 1; l <= L; ++l) { a.m.t_out[l] = 4; args.pl.ks32[l] = 2; }
        off16 = 0;
        for (int l = 1; l < L; ++l) { args.pl.off16[l] = off16; off16 += 4 * 2 * nparts * 512; }
    }
    a.m.lds_off[L] = (((off16 + 1) / 2) + 3) & ~3;
    const size_t lds_bytes = (size_t)a.m.lds_off[L] * sizeof(float);
    if (lds_bytes > 160 * 1024) return umnn_fail(UMNN_EUNSUPPORTED, "invert: weight images exceed 160 KiB of LDS");
    // exact variant for (T, live registers) if instantiated, else the generic one of the tile-count bucket (runtime counts:
    // only reached by unpadded plans -- every padded or wide plan has its exact variant above)
    const int wpb = (wide && exact && nparts == 2 && 2 * (lds_bytes + 1024) > 160 * 1024) ? 8 : 4;      // one workgroup per CU: eight waves
    const InvVariant* pick = nullptr;
    for (int ex = exact; ex >= 0 && !pick; --ex)
        for (int pass = 0; pass < 2 && !pick; ++pass)
            for (const InvVariant& v : kInvVariants)
                if (v.tmax == (ex ? T : (tmax <= 2 ? 2 : tmax <= 4 ? 4 : 8)) && v.exact == ex && v.nparts == nparts && v.wpb == (ex ? wpb : 4) &&
                    (pass == 0 ? (ex && nrl && v.nrl == nrl) : v.nrl == 0)) { pick = &v; break; }
    if (!pick) return umnn_fail(UMNN_EUNSUPPORTED, "invert: no kernel variant for this shape");
    a.ngroups = (unsigned)B;                                  // one tile (= one sample) per wave, or per workgroup (small batches)
    a.ns = split_over(pick->wpb);
    size_t lds_run = lds_bytes + (a.ns > 1 ? (size_t)pick->wpb * 16 * sizeof(float) : 0);
    if (lds_run > 160 * 1024) { a.ns = 1; lds_run = lds_bytes; }
    if (int rc = umnn_allow_lds((const void*)pick->fn, lds_run)) return rc;
    const unsigned gpb = pick->wpb / a.ns;
    const unsigned nblk = (a.ngroups + gpb - 1) / gpb;
    return launch(pick->fn, pick->name, nblk, lds_run, 64 * pick->wpb);
}

#ifndef UMNN_FWD_PIECE_F16
extern "C" int umnn_flow_invert_dim(const umnn_mlp* net, const float* h, const float* z, const float* scaling,
                                    const float* cc_w, const float* cc_s, int nb_steps,
                                    long long B, int d, int E, int j, int iters, float* x_inv, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!net) return umnn_fail(UMNN_EINVAL, "net is null");
    if (nb_steps < 1 || iters < 1) return umnn_fail(UMNN_EINVAL, "invert: nb_steps and iters must be >= 1");
    if (B < 0 || d < 1 || j < 0 || j >= d) return umnn_fail(UMNN_EINVAL, "invert: B >= 0, d >= 1, 0 <= j < d");
    {
        MlpDev m; int tmax = 0, ksu = 0;
        if (int rc = umnn_prepare_mlp(net, E, &m, &tmax, &ksu)) return rc;
        if (B == 0) return 0;
        if (!h || !z || !scaling || !cc_w || !cc_s || !x_inv) return umnn_fail(UMNN_EINVAL, "invert: null pointer");
        if (m.n_linear - 1 < 2) return umnn_fail(UMNN_EUNSUPPORTED, "invert: the matrix-core kernels need at least two hidden layers");
        // f16x3 (default): the search on two fp16 pieces (fp32-level products) with its queued bf16x3 fallback.  bf16x3: two bf16 pieces.
        // fwd_precision = fp32 / bf16x6 ("exact products everywhere"): the three-piece variants, which exist for up to four tiles per
        // layer; wider nets keep the caller's host-driven search on the forward kernels of that mode -- never a silent ~6e-6 search
        const int prec = umnn_options().fwd_precision;
        if (prec == UMNN_PRECISION_F16X3)
            return umnn_invert_impl_f16(net, h, z, scaling, cc_w, cc_s, nb_steps, B, d, E, j, iters, x_inv, stream, 2, nullptr);
        const int nparts = prec == UMNN_PRECISION_BF16X3 ? 2 : 3;
        // (8 tiles x 3 bf16 pieces do not fit the register file: nets above four tiles per layer get their fp32-level products from
        // the two-fp16-piece search instead -- the same accuracy class, ~5e-7 on F; a sample whose candidates overflow fp16 is redone
        // on two bf16 pieces, ~6e-6 on F, three orders of magnitude inside the search's own resolution 100 (2/9)^iter.  Until round 4
        // this case returned UMNN_EUNSUPPORTED and the caller drove d x iter forward launches from the host)
        if (nparts == 3 && tmax > 4)
            return umnn_invert_impl_f16(net, h, z, scaling, cc_w, cc_s, nb_steps, B, d, E, j, iters, x_inv, stream, 2, nullptr);
        return umnn_invert_impl_bf16(net, h, z, scaling, cc_w, cc_s, nb_steps, B, d, E, j, iters, x_inv, stream, nparts, nullptr);
    }
}
#endif
