// Shared by the forward kernels (fp32-MFMA and bf16-split): launch arguments and the common epilogue.
#pragma once
#include "cc_common.h"

struct FwdArgs {
    MlpDev m;
    const float* x0;   // nullable
    const float* x;
    const float* h;
    const float* ccw;
    const float* ccs;
    float* F;          // nullable when flow epilogue is used
    float* fx;         // nullable
    float* fx0;        // nullable
    const float* scaling;  // flow epilogue (nullable => plain integral)
    float* z;
    float* logjac;
    const float* logjac_in;   // nullable: running sum of the previous blocks' log_jac (may alias logjac)
    int reverse_z;            // write z with the dimensions reversed (the flip between the blocks of a flow)
    // one-pass log-likelihood (umnn_flow_ll_block_forward): ll[b] (+)= sum_i log_jac[b,i]  (- Gaussian term, last block)
    float* ll;                // [B] running log-likelihood; nullable => off
    unsigned* row_cnt;        // [B] arrival counters: zero on entry, zero again on exit
    int ll_first, ll_last;    // first block overwrites ll; last block adds -1/2 sum_i (log 2pi + z^2)
    // in-kernel inversion (umnn_flow_invert_dim, INV kernel variants): one 16-point tile = ONE sample, its points are
    // the 10 candidates of a bracket-search round; NI counts samples, d is the flow's full dimension
    const float* inv_z;       // [B, d] targets; column inv_j is used
    float* inv_x;             // [B, d] result; column inv_j is written
    int inv_j, inv_iters;
    int x_bf16, h_bf16;       // storage of the x-class tensors (x, x0, F, f_x, f_x0, z, log_jac) / of h: 0 fp32, 1 bf16
    long long NI;      // B*d integrals
    int d, E, n, ns, inv_f;
    unsigned ngroups;  // tile groups (of 16*P integrals)
    // fp16-piece launches and their queued bf16 fallback (cc_forward_bf16.hip, "overflow protocol" there).  ovf_mode 0: plain launch.
    // 1 (the fp16 build): a tile group whose quadrature sum is not finite -- an overflowed fp16 piece always ends there -- writes
    // nothing but a NaN into its slots of the MARKER output (F, or z for the flow entry points), does not count towards the one-pass
    // log-likelihood, and raises *ovf_flag to ovf_gen.  2 (the bf16 build queued right behind it with the same group mapping): returns
    // at once unless *ovf_flag >= ovf_gen, then recomputes exactly the groups whose marker slots hold a NaN and publishes them.
    unsigned long long* ovf_flag;
    unsigned long long ovf_gen;
    int ovf_mode;
    // training forward of the wide-first family (31-100-50-50-50-50-1): the pre-activations of hidden layer 2 of EVERY node leave for
    // HBM in the register layout of the three-stage backward's middle stage -- [tile][node][register < z2_nl2][lane], cc_backward_front.hip
    // -- so that its stage A (which recomputes exactly these) need not run (umnn_flow_stack_block_forward_save).  nullable.
    float* z2_save;
    int z2_nl2;
};

// The marker of a deferred group is a NaN with a payload no arithmetic produces (hardware NaNs are 0x7fc00000, propagated input NaNs keep the
// caller's payload; this one survives the bf16 round trip of the *_io entry points: 0x7fd5 as bf16).  A group that was NOT deferred but whose
// z is legitimately NaN (exp(s) (F + h_0) = inf - inf with a finite quadrature sum) is therefore never mistaken for a marked one by the
// fallback launch -- it would publish twice and count twice towards its rows' arrival counters.
constexpr unsigned FWD_MARKER_BITS = 0x7fd50000u;
// the marker slot of integral q: element q of F when the launch writes F, else this integral's element of z
__device__ __forceinline__ float* fwd_marker(const FwdArgs& a, long long q, long long* idx) {
    if (a.F) { *idx = q; return a.F; }
    const long long bi = q / a.d;
    *idx = a.reverse_z ? bi * a.d + (a.d - 1 - (q - bi * a.d)) : q;
    return a.z;
}
// ovf_mode 2: does any integral of this wave's tile group carry the marker?  (wave-uniform; lanes beyond NI re-read the last integral,
// which belongs to the same -- the last -- group)
template <int P>
__device__ __forceinline__ bool fwd_group_marked(const FwdArgs& a, unsigned grp, int p) {
    bool m = false;
#pragma unroll
    for (int pt = 0; pt < P; ++pt) {
        long long q = ((long long)grp * P + pt) * 16 + p;
        if (q >= a.NI) q = a.NI - 1;
        long long idx;
        float* mk = fwd_marker(a, q, &idx);
        const float v = io_ld(mk, idx, a.x_bf16);
        m = m || __float_as_uint(v) == FWD_MARKER_BITS;
    }
    return __any(m);
}

// Combine the node-range partials of the NS waves sharing a tile group (through LDS), then write F, f(x), f(x0)
// and, for the flow entry point, z and log_jac.  Called by every wave of the workgroup (it contains a barrier).
template <int P>
__device__ __forceinline__ void fwd_epilogue(const FwdArgs& a, float* lds, float (&Facc)[P], float (&fxv)[P],
                                             float (&fx0v)[P], const bool (&ok)[P], const long long (&qv)[P],
                                             const float (&dxv)[P], bool live, int part, int ns, int wid, int g, int p) {
    const MlpDev& m = a.m;
    const int L = m.n_linear - 1, d = a.d, E = a.E;
    if (ns > 1) {
        float* red = lds + m.lds_off[L];      // [waves][3][P*16]
        if (live && g == 0) {
#pragma unroll
            for (int pt = 0; pt < P; ++pt) {
                float* rw = red + wid * (3 * P * 16) + pt * 16 + p;
                rw[0] = Facc[pt];
                rw[P * 16] = fxv[pt];
                rw[2 * P * 16] = fx0v[pt];
            }
        }
        __syncthreads();
        if (live && part == 0 && g == 0) {
#pragma unroll
            for (int pt = 0; pt < P; ++pt) {
                float s = 0.f;
                for (int j = 0; j < ns; ++j) s += red[(wid + j) * (3 * P * 16) + pt * 16 + p];
                Facc[pt] = s;
                fx0v[pt] = red[(wid + ns - 1) * (3 * P * 16) + 2 * P * 16 + pt * 16 + p];
            }
        }
    }
    // overflow protocol, fp16 build: a non-finite quadrature sum anywhere in the group defers the WHOLE group to the queued bf16 launch
    bool deferred = false;
    if (a.ovf_mode == 1) {
        bool bad = false;
        if (live && part == 0 && g == 0) {
#pragma unroll
            for (int pt = 0; pt < P; ++pt) bad = bad || (ok[pt] && !(__builtin_fabsf(Facc[pt]) < __builtin_inff()));
        }
        deferred = __any(bad);
        if (deferred) {
            if (g == 0 && part == 0) {
#pragma unroll
                for (int pt = 0; pt < P; ++pt) {
                    if (!ok[pt]) continue;
                    long long idx;
                    float* mk = fwd_marker(a, qv[pt], &idx);
                    io_st(mk, idx, __uint_as_float(FWD_MARKER_BITS), a.x_bf16);
                }
            }
            if (part == 0 && g == 0 && p == 0) atomicMax(a.ovf_flag, a.ovf_gen);
        }
    }
    if (live && part == 0 && g == 0 && !deferred) {
#pragma unroll
        for (int pt = 0; pt < P; ++pt) {
            if (!ok[pt]) continue;
            const long long q = qv[pt];
            const float Fv = Facc[pt] * dxv[pt] * 0.5f;
            if (a.F) io_st(a.F, q, Fv, a.x_bf16);
            if (a.fx) io_st(a.fx, q, fxv[pt], a.x_bf16);
            if (a.fx0) io_st(a.fx0, q, fx0v[pt], a.x_bf16);
            if (a.scaling) {
                const long long bi = q / d;
                const int i = (int)(q - bi * d);
                const float sc = a.scaling[i];
                const float z0 = io_ld(a.h, bi * ((long long)E * d) + i, a.h_bf16);
                io_st(a.z, a.reverse_z ? bi * d + (d - 1 - i) : q, __expf(sc) * (Fv + z0), a.x_bf16);
                const float lj = __logf(fxv[pt] + 1e-10f) + sc;
                io_st(a.logjac, q, a.logjac_in ? io_ld(a.logjac_in, q, a.x_bf16) + lj : lj, a.x_bf16);
            }
        }
    }
    // ---- one-pass log-likelihood: the LAST tile to deliver a piece of row b sums that row, in a fixed order ---------
    // (UMNNMAFFlow.compute_ll, UMNNMAFFlow.py:109-119: ll = sum_blocks sum_i log_jac - 1/2 sum_i (log 2pi + z_i^2).)
    // Every tile publishes its z / log_jac stores (release fence), then the head lane of each row segment adds the
    // segment length to the row's arrival counter; whoever brings it to d owns the row: after an acquire fence the
    // whole wave reads the d values back (L2 / HBM), reduces them with a fixed butterfly and updates ll[b].  One
    // writer per row and launch, fixed summation order: bit-reproducible, no float atomics, no extra launch.
    if (a.ll) {
        // release (cdna_hip_programming.md Guideline 16): drain this wave's stores, agent-scope release, drain again
        // (the compiler may drop the fence's own wait), THEN move the counters
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int lane = 16 * g + p;
#pragma unroll
        for (int pt = 0; pt < P; ++pt) {
            const bool writer = live && part == 0 && g == 0 && ok[pt] && !deferred;
            const long long q = qv[pt];
            const long long bi = q / d;
            const int i = (int)(q - bi * d);
            bool fin = false;
            if (writer && (p == 0 || i == 0)) {
                long long cnt = d - i;
                if (cnt > 16 - p) cnt = 16 - p;
                if (cnt > a.NI - q) cnt = a.NI - q;
                const unsigned old = __hip_atomic_fetch_add(a.row_cnt + bi, (unsigned)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                fin = old + (unsigned)cnt == (unsigned)d;
                // self-cleaning: the next launch on this stream starts from zero again
                if (fin) __hip_atomic_store(a.row_cnt + bi, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            unsigned long long todo = __ballot(fin);
            if (todo) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ONE acquire, then plain vector loads
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const long long b = ((long long)__builtin_amdgcn_readlane((int)(bi >> 32), src) << 32) |
                                    (unsigned)__builtin_amdgcn_readlane((int)bi, src);
                float* ljr = a.logjac + b * d;
                float* zr = a.z + b * d;
                float s = 0.f;
                for (int e = lane; e < d; e += 64) {
                    float v = ljr[e];
                    if (a.ll_last) {
                        const float zz = zr[e];
                        v -= 0.5f * (1.8378770664093453f + zz * zz);         // log(2 pi)
                    }
                    s += v;
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
                if (lane == 0) a.ll[b] = a.ll_first ? s : a.ll[b] + s;
            }
        }
    }
}
