// Backward of the Clenshaw-Curtis quadrature of the integrand MLP (the reference's gradient convention).
//
// Replaces (reference):
//   models/UMNN/ParallelNeuralIntegral.py:66-94   integrate(compute_grad=True) + computeIntegrand:
//        VJP of f at every node with cotangent g*(x-x0)/2*w_k  ->  d_theta (flat), d_h (summed over nodes)
//   models/UMNN/ParallelNeuralIntegral.py:110-123 Leibniz terms d_x = f(x;h) g, d_x0 = -f(x0;h) g
//   (NeuralIntegral.py:47-58,69-75,90-99 is the same arithmetic, node by node)
// plus, for the fused flow block, the VJP of the extra output f_x = f(x;h) (cotangent g_fx) that the
// reference obtains from plain autograd through IntegrandNetwork.forward (UMNNMAF.py:138,143,148).
//
// Structure (one launch of cc_bwd_kernel per "pass", then three small finishing kernels):
//   * persistent waves (one per SIMD, the whole 512-register file): each wave walks tiles of 16 integrals and,
//     per quadrature node, recomputes the forward chain on MFMA (weights: row-major padded LDS images, used for
//     both W and W^T fragments), keeps only the SIGN BITS of the activations, back-propagates delta through
//     W^T on MFMA, and accumulates dW_l += delta_{l+1} (x) a_l on MFMA.  That last product contracts over the
//     16 points of the tile, which sit on the wrong lane axis, so delta and a take one trip through a
//     wave-private LDS tile (written [feature][point], read back as b128 rows = "feature on lane&15, points on
//     (lane>>4, r)").  dW accumulators for up to NACC hidden layers live in registers for the whole kernel;
//   * what depends on an integral only once (not per node) leaves the kernel as dc[q][f] = sum_k delta_1: the
//     finishing kernels turn it into d_h = W1h^T dc and dW1[:,1:], db1;
//   * every wave writes its partial d_theta to its own slice of the workspace; a last kernel sums the slices
//     (deterministic, no atomics).
#include <cstring>

#include "cc_bwd_shared.h"

__device__ __forceinline__ void stage_rowmajor_images(const BwdArgs& a, float* lds, int tid, int nthreads) {
    const MlpDev& m = a.m;
    const int L = m.n_linear - 1;
    for (int l = 1; l < L; ++l) {
        const int Hin = m.width[l], Hout = m.width[l + 1];
        // rows: only the 4*ks_in[l+1] features the next layer consumes are stored.  Fragment reads of the rows
        // (and, for W^T fragments, columns) beyond that run into the following image / the zero-initialised
        // scratch: finite values that only ever reach padding features, whose results are never consumed.
        const int rows = 4 * m.ks_in[l + 1], LD = a.ld[l];
        const float* __restrict__ W = m.W[l];
        const float* __restrict__ b = m.b[l];
        float* img = lds + a.roff[l];
        for (int idx = tid; idx < rows * LD; idx += nthreads) {
            const int fo = idx / LD, fi = idx - fo * LD;
            float v = 0.f;
            if (fo < Hout) {
                if (fi < Hin) v = W[fo * Hin + fi];
                else if (fi == Hin) v = b[fo];
            } else if (fo == Hout && fi == Hin) {
                v = 1.f;
            }
            img[idx] = v;
        }
    }
}

// out[t] = sum_s W-fragment(t, s) * in[s]   (forward orientation; image row = output feature)
// KSC > 0: exact shapes -- every hidden layer has KSC K-steps and fills all TMAX tiles, so the wave-uniform guards
// (and the register copies they force at every basic-block boundary) vanish.  KSC = 0: per-layer runtime counts.
template <int TMAX, int KSC>
__device__ __forceinline__ void layer_fwd(const float* wf, int LD, int ks, int to, const f32x4 (&in)[TMAX],
                                          f32x4 (&out)[TMAX]) {
#pragma unroll
    for (int t = 0; t < TMAX; ++t) out[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4 * TMAX; ++s) {
        if (KSC ? (s < KSC) : (s < ks)) {
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (KSC || t < to) out[t] = mfma16(wf[16 * t * LD + 4 * s], in[s >> 2][s & 3], out[t]);
        }
    }
}

// out[t] = sum_s W^T-fragment(t, s) * in[s]   (backward orientation; image row = K index)
template <int TMAX, int KSC>
__device__ __forceinline__ void layer_bwd(const float* wt, int LD, int ks, int to, const f32x4 (&in)[TMAX],
                                          f32x4 (&out)[TMAX]) {
#pragma unroll
    for (int t = 0; t < TMAX; ++t) out[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4 * TMAX; ++s) {
        if (KSC ? (s < KSC) : (s < ks)) {
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (KSC || t < to) out[t] = mfma16(wt[4 * s * LD + 16 * t], in[s >> 2][s & 3], out[t]);
        }
    }
}

template <int TMAX>
__device__ __forceinline__ unsigned sign_bits(const f32x4 (&v)[TMAX]) {
    unsigned bits = 0;
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) bits |= (v[t][r] > 0.f ? 1u : 0u) << (4 * t + r);
    return bits;
}

template <int TMAX, int NACC, bool EDGE, int KSC>
__global__ __launch_bounds__(UMNN_BLOCK, 1) void cc_bwd_kernel(const BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const MlpDev& m = a.m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, p = lane & 15;
    const int L = m.n_linear - 1;
    const int H1 = m.width[1], HL = m.width[L];
    const int E = a.E, d = a.d, n = a.n;
    const float slope = m.hidden_act == UMNN_ACT_RELU ? 0.f : 0.01f;
    constexpr int NA = NACC > 0 ? NACC : 1;

    const int nwb = blockDim.x >> 6;               // waves in this workgroup (1, 2 or 4: whatever fits in LDS)
    stage_rowmajor_images(a, lds, tid, blockDim.x);
    for (int i = tid; i < nwb * a.scratch_per_wave; i += blockDim.x) lds[a.scratch_off + i] = 0.f;
    __syncthreads();

    float* scratch = lds + a.scratch_off + wid * a.scratch_per_wave;   // [NACC a-tiles | 1 delta tile] x (TMAX*256)
    float* s_delta = scratch + NACC * (TMAX * 256);
    const int wr_off = g * 16 + p;                 // + (16t+4r)*16 : standard-layout write
    const int rd_off = (lane & 15) * 16 + 4 * g;   // + 16t*16      : transposed b128 read

    // per-lane constants
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

    // accumulators that live for the whole kernel
    f32x4 dW[NA][TMAX][TMAX];
#pragma unroll
    for (int j = 0; j < NA; ++j)
#pragma unroll
        for (int to = 0; to < TMAX; ++to)
#pragma unroll
            for (int ti = 0; ti < TMAX; ++ti) dW[j][to][ti] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 dW1x[TMAX], dwo[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) { dW1x[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dwo[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const unsigned wave_global = blockIdx.x * nwb + wid;
    const unsigned nwaves = gridDim.x * nwb;

    const unsigned nsp = a.ns > 1 ? (unsigned)a.ns : 1u;
    for (unsigned item = wave_global; item < a.ngroups * nsp; item += nwaves) {
        const unsigned grp = item / nsp, part = item - grp * nsp;
        const int k_lo = (int)(((long long)part * (n + 1)) / nsp), k_hi = (int)(((long long)(part + 1) * (n + 1)) / nsp);
        const long long q = (long long)grp * 16 + p;
        const bool ok = q < a.NI;
        const long long qq = ok ? q : a.NI - 1;
        const float xv = io_ld(a.x, qq, a.x_bf16);
        const float x0v = a.x0 ? io_ld(a.x0, qq, a.x_bf16) : 0.f;
        const float dxv = xv - x0v;
        const float gv = ok ? io_ld(a.g, qq, a.x_bf16) : 0.f;                       // dead lanes contribute nothing
        const float gfxv = (ok && a.gfx) ? io_ld(a.gfx, qq, a.x_bf16) : 0.f;
        const float cotbase = gv * dxv * 0.5f;
        const long long bi = qq / d;
        const IoView hb = IoView{a.h, a.h_bf16} + (bi * ((long long)E * d) + (qq - bi * d));

        // hoisted first-layer term (same as the forward kernel)
        f32x4 c[TMAX];
        {
            const float* __restrict__ W0 = m.W[0];
            const float* __restrict__ b0 = m.b[0];
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = feat_of(t, r, g);
                    c[t][r] = f < H1 ? b0[f] : (f == H1 ? 1.f : 0.f);
                }
            for (int se = 0; se < (E + 3) / 4; ++se) {
                const int e = 4 * se + g;
                const float hv = e < E ? hb[(long long)e * d] : 0.f;
#pragma unroll
                for (int t = 0; t < TMAX; ++t) {
                    if (KSC || t < m.t_out[1]) {
                        const int fo = fout_of(t, p);
                        const float A = (fo < H1 && e < E) ? W0[fo * (1 + E) + 1 + e] : 0.f;
                        c[t] = mfma16(A, hv, c[t]);
                    }
                }
            }
        }

        f32x4 dcs[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) dcs[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        float fxv = 0.f, fx0v = 0.f, dfdt = 0.f;

        for (int k = k_lo; k < k_hi; ++k) {
            const float u = a.ccs[k] + 1.f;
            const float wk = a.ccw[k];
            const float tk = k == 0 ? xv : __fadd_rn(x0v, __fmul_rn(dxv, u) * 0.5f);
            unsigned bits[UMNN_MAX_LINEAR];          // bits[l]: sign bits of hidden layer l's activation
            f32x4 act[TMAX];
            // ---------------- forward recompute ----------------
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) act[t][r] = hidden_act_f(fmaf(w1x[t][r], tk, c[t][r]), slope);
            bits[1] = sign_bits<TMAX>(act);
            for (int l = 1; l < L; ++l) {
                // a_l is the input of image l: park it for the dW product if this pass owns layer l
#pragma unroll
                for (int j = 0; j < NACC; ++j)
                    if (l == a.l_lo + j) {
#pragma unroll
                        for (int t = 0; t < TMAX; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) scratch[j * (TMAX * 256) + (16 * t + 4 * r) * 16 + wr_off] = act[t][r];
                    }
                f32x4 acc[TMAX];
                layer_fwd<TMAX, KSC>(lds + a.roff[l] + perm16(p) * a.ld[l] + g, a.ld[l], m.ks_in[l], m.t_out[l + 1], act, acc);
#pragma unroll
                for (int t = 0; t < TMAX; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        act[t][r] = (KSC || t < m.t_out[l + 1]) ? hidden_act_f(acc[t][r], slope) : 0.f;
                bits[l + 1] = sign_bits<TMAX>(act);
            }
            float sdot = 0.f;
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) sdot = fmaf(wout[t][r], act[t][r], sdot);
            sdot = group_allreduce(sdot);
            const float f = out_act_f(sdot, m.out_act);
            const float fp = out_grad_f(sdot, m.out_act);
            if (k == 0) fxv = f;
            if (k == n) fx0v = f;

            // ---------------- tangent pass at node 0: d f / d x for the g_fx term ----------------
            if (EDGE && k == 0 && a.gfx) {
                f32x4 ta[TMAX];
#pragma unroll
                for (int t = 0; t < TMAX; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ta[t][r] = w1x[t][r] * ((bits[1] >> (4 * t + r)) & 1u ? 1.f : slope);
                for (int l = 1; l < L; ++l) {
                    f32x4 tz[TMAX];
                    layer_fwd<TMAX, KSC>(lds + a.roff[l] + perm16(p) * a.ld[l] + g, a.ld[l], m.ks_in[l], m.t_out[l + 1], ta, tz);
#pragma unroll
                    for (int t = 0; t < TMAX; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            ta[t][r] = (KSC || t < m.t_out[l + 1]) ? tz[t][r] * ((bits[l + 1] >> (4 * t + r)) & 1u ? 1.f : slope) : 0.f;
                }
                float ds = 0.f;
#pragma unroll
                for (int t = 0; t < TMAX; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ds = fmaf(wout[t][r], ta[t][r], ds);
                dfdt = fp * group_allreduce(ds);
            }

            // ---------------- backward sweep ----------------
            const float invs = a.inv_f ? -__frcp_rn(f * f) : 1.f;
            const float cot = fmaf(cotbase * invs, wk, k == 0 ? gfxv : 0.f);
            const float dout = cot * fp;
            f32x4 delta[TMAX];
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (EDGE) dwo[t][r] = fmaf(dout, act[t][r], dwo[t][r]);
                    delta[t][r] = dout * wout[t][r] * ((bits[L] >> (4 * t + r)) & 1u ? 1.f : slope);
                }
            for (int l = L - 1; l >= 1; --l) {
                // dW_l += delta_{l+1} (x) a_l  (contract over the 16 points through the LDS transpose)
#pragma unroll
                for (int j = 0; j < NACC; ++j)
                    if (l == a.l_lo + j) {
#pragma unroll
                        for (int t = 0; t < TMAX; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) s_delta[(16 * t + 4 * r) * 16 + wr_off] = delta[t][r];
                        f32x4 dT[TMAX], aT[TMAX];
#pragma unroll
                        for (int t = 0; t < TMAX; ++t) {
                            dT[t] = *reinterpret_cast<const f32x4*>(s_delta + 16 * t * 16 + rd_off);
                            aT[t] = *reinterpret_cast<const f32x4*>(scratch + j * (TMAX * 256) + 16 * t * 16 + rd_off);
                        }
#pragma unroll
                        for (int to = 0; to < TMAX; ++to)
                            if (KSC || to < m.t_out[l + 1]) {
#pragma unroll
                                for (int ti = 0; ti < TMAX; ++ti)
                                    if (KSC || ti < m.t_out[l]) {
#pragma unroll
                                        for (int r = 0; r < 4; ++r)
                                            dW[j][to][ti] = mfma16(dT[to][r], aT[ti][r], dW[j][to][ti]);
                                    }
                            }
                    }
                // delta_l = (W_l^T delta_{l+1}) * act'(z_l)
                f32x4 nd[TMAX];
                layer_bwd<TMAX, KSC>(lds + a.roff[l] + g * a.ld[l] + perm16(p), a.ld[l], m.ks_in[l + 1], m.t_out[l], delta, nd);
#pragma unroll
                for (int t = 0; t < TMAX; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        delta[t][r] = (KSC || t < m.t_out[l]) ? nd[t][r] * ((bits[l] >> (4 * t + r)) & 1u ? 1.f : slope) : 0.f;
            }
            if (EDGE) {
#pragma unroll
                for (int t = 0; t < TMAX; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dcs[t][r] += delta[t][r];
                        dW1x[t][r] = fmaf(delta[t][r], tk, dW1x[t][r]);
                    }
            }
        }

        if (EDGE && ok) {
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = feat_of(t, r, g);
                    if (f < H1) a.dc[(size_t)part * a.NI * H1 + q * H1 + f] = dcs[t][r];
                }
            if (g == 0) {       // Leibniz terms: from the work items that own node 0 / node n
                if (a.dx && k_lo == 0) io_st(a.dx, q, fmaf(gfxv, dfdt, fxv * gv), a.x_bf16);
                if (a.dx0 && k_hi == n + 1) io_st(a.dx0, q, -fx0v * gv, a.x_bf16);
            }
        }
    }

    // ---------------- write this wave's partial d_theta ----------------
    float* part = a.partials + (size_t)wave_global * a.n_params;
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
        const int l = a.l_lo + j;
        if (l < L) {
            const int Hin = m.width[l], Hout = m.width[l + 1];
#pragma unroll
            for (int to = 0; to < TMAX; ++to)
#pragma unroll
                for (int ti = 0; ti < TMAX; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int fo = 16 * to + 4 * g + r, fi = 16 * ti + (lane & 15);
                        if (fo < Hout) {
                            if (fi < Hin) part[a.poffW[l] + fo * Hin + fi] = dW[j][to][ti][r];
                            else if (fi == Hin) part[a.poffb[l] + fo] = dW[j][to][ti][r];
                        }
                    }
        }
    }
    if (EDGE) {
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v1 = dW1x[t][r], v2 = dwo[t][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { v1 += __shfl_xor(v1, o); v2 += __shfl_xor(v2, o); }
                const int f = feat_of(t, r, g);
                if (p == 0) {
                    if (f < H1) part[a.poffW[0] + f * (1 + E)] = v1;
                    if (f < HL) part[a.poffW[L] + f] = v2;
                    else if (f == HL) part[a.poffb[L]] = v2;
                }
            }
    }
}

// d_h[b, e*d+i] = sum_f W1[f][1+e] dc[q][f]
__global__ __launch_bounds__(256) void cc_bwd_dh_kernel(const float* __restrict__ dc, const float* __restrict__ W0,
                                                        float* __restrict__ dh, long long NI, int d, int E, int H1, int qpb,
                                                        int h_bf16) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sW = sm;                       // [H1][E]
    float* sdc = sm + H1 * E;             // [qpb][H1+1], qpb = integrals per block (64, or 16 for small batches)
    const int tid = threadIdx.x;
    for (int i = tid; i < H1 * E; i += 256) { const int f = i / E, e = i - f * E; sW[i] = W0[f * (1 + E) + 1 + e]; }
    const long long q0 = (long long)blockIdx.x * qpb;
    const int nq = (int)min((long long)qpb, NI - q0);
    for (int i = tid; i < nq * H1; i += 256) {
        const int ql = i / H1, f = i - ql * H1;
        sdc[ql * (H1 + 1) + f] = dc[q0 * H1 + i];
    }
    __syncthreads();
    for (int o = tid; o < nq * E; o += 256) {
        const int e = o / nq, ql = o - e * nq;      // ql fastest: consecutive threads -> consecutive i
        float s = 0.f;
        for (int f = 0; f < H1; ++f) s = fmaf(sW[f * E + e], sdc[ql * (H1 + 1) + f], s);
        const long long q = q0 + ql, bi = q / d;
        io_st(dh, bi * ((long long)E * d) + (long long)e * d + (q - bi * d), s, h_bf16);
    }
}

// The same product on the fp32 matrix cores (large batches): d_h^T[e][q] = sum_f W1[f][1+e] dc[q][f] as D[e][q] += A[e][f] B[f][q]
// with v_mfma_f32_16x16x4_f32.  A wave owns 16 integrals at a time: B operands are its dc rows straight from HBM (lane (g, p):
// integral q0 + p, feature 4 ks + g -- the four lane groups of a row read 16 contiguous bytes), A operands come from a [ks][te][lane]
// image of W1[:, 1:]^T in LDS (one conflict-free ds_read_b32 per MFMA), the result tile (lane (g, p): integral p, embedding columns
// 16 te + 4 g + r) is stored with 64-byte runs along the integrals.  The first version above reads two LDS words per FMA and sat
// at 1.6 TB/s of dc + d_h traffic (101 us per C3 call against 165 MB); this one is bound by that stream.
__global__ __launch_bounds__(256) void cc_bwd_dh_mfma_kernel(const float* __restrict__ dc, const float* __restrict__ W0,
                                                             float* __restrict__ dh, long long NI, int d, int E, int H1, int h_bf16) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int KS = (H1 + 3) / 4, TE = (E + 15) / 16;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, p = lane & 15;
    for (int i = tid; i < KS * TE * 64; i += 256) {
        const int ln = i & 63, te = (i >> 6) % TE, ks = (i >> 6) / TE;
        const int e = 16 * te + (ln & 15), f = 4 * ks + (ln >> 4);
        sm[i] = (e < E && f < H1) ? W0[f * (1 + E) + 1 + e] : 0.f;
    }
    __syncthreads();
    const long long ngroups = (NI + 15) / 16;
    const long long wave = (long long)blockIdx.x * 4 + (tid >> 6), nwaves = (long long)gridDim.x * 4;
    for (long long grp = wave; grp < ngroups; grp += nwaves) {
        const long long q = grp * 16 + p;
        const bool ok = q < NI;
        const float* __restrict__ row = dc + (ok ? q : NI - 1) * H1;
        f32x4 acc[5];
#pragma unroll
        for (int te = 0; te < 5; ++te) acc[te] = f32x4{0.f, 0.f, 0.f, 0.f};
        float bnext = g < H1 ? row[g] : 0.f;
        for (int ks = 0; ks < KS; ++ks) {
            const float bv = bnext;
            const int fn = 4 * (ks + 1) + g;
            bnext = (ks + 1 < KS && fn < H1) ? row[fn] : 0.f;
#pragma unroll
            for (int te = 0; te < 5; ++te)
                if (te < TE) acc[te] = mfma16(sm[(ks * TE + te) * 64 + lane], bv, acc[te]);
        }
        if (ok) {
            const long long bi = q / d;
            const long long base = bi * ((long long)E * d) + (q - bi * d);
#pragma unroll
            for (int te = 0; te < 5; ++te)
                if (te < TE) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = 16 * te + 4 * g + r;
                        if (e < E) io_st(dh, base + (long long)e * d, acc[te][r], h_bf16);
                    }
                }
        }
    }
}

// node-split runs: dc[0][i] += dc[1][i] + ... + dc[ns-1][i], in that order (deterministic)
__global__ __launch_bounds__(256) void cc_bwd_dcsum_kernel(float* __restrict__ dc, long long count, int ns) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    float v = dc[i];
    for (int j = 1; j < ns; ++j) v += dc[(size_t)j * count + i];
    dc[i] = v;
}

// partial[blk][f][e] = sum_{q in chunk} dc[q][f] * hext[q][e],  hext[q][E] = 1  (-> dW1[:,1:], db1)
// A skinny GEMM (H1 x (E+1) outputs, contraction over the chunk's integrals) on the fp32 matrix cores: the chunk of dc and of
// the gathered embedding rows is staged in LDS -- dc integral-major with a row stride of 16 mod 32 floats (the four 16-float row
// segments of a fragment read fall into four different bank groups), h column-major with a stride of chunk + 4 (lane (g, p) reads
// bank 4 * (k p mod 16) + g): both the staging writes and the fragment reads are conflict-free -- every wave owns output tiles
// (16 features x 16 embedding columns) and walks the chunk four integrals per v_mfma_f32_16x16x4_f32.  (Round 1-2 had one thread per output and two LDS reads per FMA:
// 0.40 ms per C3 call against 165 MB of traffic; this one is bound by the dc / h stream.)
__host__ __device__ inline int dw0_stride(int cols) { return ((cols + 15) / 16 * 16 + 16 + 31) / 32 * 32 - 16; }   // >= cols, = 16 mod 32
__host__ __device__ inline size_t dw0_lds_floats(int chunk, int H1, int E) {
    return (size_t)chunk * dw0_stride(H1) + (size_t)((E + 1 + 15) / 16 * 16) * (chunk + 4);
}
// DW0_MAXT = output tiles per wave (compile-time: the accumulators stay in registers): 2 covers the 48..64-wide, E <= 31 nets,
// 4 the 100-wide ones, 10 = ceil(8 x 5 / 4) everything up to H1 = 127, E = 79 in one launch (wider embeddings: one launch per
// 40 tiles, `tile_base`; every launch writes its own entries of the same partial slices).
template <int DW0_MAXT>
__global__ __launch_bounds__(256) void cc_bwd_dw0_kernel(const float* __restrict__ dc, const float* __restrict__ h,
                                                         float* __restrict__ partial, long long NI, int d, int E,
                                                         int H1, int chunk, int h_bf16, int tile_base) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int HS = dw0_stride(H1), CS = chunk + 4;
    float* sdc = sm;                          // [chunk][HS]
    float* sh = sm + chunk * HS;              // [16 ET][CS]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int FT = (H1 + 15) / 16, ET = (E + 1 + 15) / 16, ntiles = FT * ET;
    const int g = lane >> 4, p = lane & 15;
    const int lc = 31 - __clz(chunk);         // (chunk is a power of two)
    const long long nchunks = (NI + chunk - 1) >> lc;
    f32x4 acc[DW0_MAXT][2];
#pragma unroll
    for (int k = 0; k < DW0_MAXT; ++k) acc[k][0] = acc[k][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // persistent blocks: block b takes chunks b, b + gridDim.x, ... and keeps its output tiles in registers across them
    for (long long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const long long q0 = c << lc;
        const int nq = (int)min((long long)chunk, NI - q0);
        for (int ql = wid; ql < chunk; ql += 4)
            for (int f = lane; f < FT * 16; f += 64) sdc[ql * HS + f] = (ql < nq && f < H1) ? dc[(q0 + ql) * H1 + f] : 0.f;
        {
            // integral q0 + ql = sample b0 + (r0 + ql) / d, dimension (r0 + ql) % d: one 64-bit division per chunk, 32-bit ones per element
            const long long b0 = q0 / d;
            const int r0 = (int)(q0 - b0 * d);
            const IoView hv = IoView{h, h_bf16} + b0 * ((long long)E * d);
            for (int i = tid; i < chunk * ET * 16; i += 256) {
                const int e = i >> lc, ql = i & (chunk - 1);
                float v = 0.f;
                if (ql < nq) {
                    const int rb = (r0 + ql) / d, ri = (r0 + ql) - rb * d;
                    v = e < E ? hv[(long long)(rb * E + e) * d + ri] : (e == E ? 1.f : 0.f);
                }
                sh[e * CS + ql] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < DW0_MAXT; ++k) {
            const int tile = tile_base + wid + 4 * k;
            if (tile < ntiles) {
                const int ft = tile / ET, et = tile - ft * ET;
                const float* pa = sdc + g * HS + 16 * ft + p;
                const float* pb = sh + (16 * et + p) * CS + g;
                for (int s = 0; s < chunk; s += 16) {          // (chunk is a multiple of 16; two accumulators: independent MFMAs)
                    acc[k][0] = mfma16(pa[s * HS], pb[s], acc[k][0]);
                    acc[k][1] = mfma16(pa[(s + 4) * HS], pb[s + 4], acc[k][1]);
                    acc[k][0] = mfma16(pa[(s + 8) * HS], pb[s + 8], acc[k][0]);
                    acc[k][1] = mfma16(pa[(s + 12) * HS], pb[s + 12], acc[k][1]);
                }
            }
        }
        __syncthreads();
    }
    const int nout = H1 * (E + 1);
    float* out = partial + (size_t)blockIdx.x * nout;
#pragma unroll
    for (int k = 0; k < DW0_MAXT; ++k) {
        const int tile = tile_base + wid + 4 * k;
        if (tile < ntiles) {
            const int ft = tile / ET, et = tile - ft * ET;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * ft + 4 * g + r, e = 16 * et + p;
                if (f < H1 && e < E + 1) out[f * (E + 1) + e] = acc[k][0][r] + acc[k][1][r];
            }
        }
    }
}

// dtheta[i] = sum_w partials[w][i]  (+ first-layer slices from the dw0 partials).  A block owns 32 consecutive parameters
// (one 128-byte row segment per slice: coalesced) and splits the slices over 8 phases: thread (phase, i) sums the slices phase,
// phase + 8, ..., then the 8 phase sums are combined through LDS in a fixed order -- deterministic, and enough loads in flight to
// hide the latency of a thousand-slice sum (small batches run many node-split waves).
__global__ __launch_bounds__(256) void cc_bwd_reduce_kernel(const float* __restrict__ partials, int nparts, int n_params,
                                                            const float* __restrict__ p0, int nparts0, int E, int H1,
                                                            int poffW0, int poffb0, float* __restrict__ dtheta) {
    __shared__ float red[8][33];
    const int il = threadIdx.x & 31, ph = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + il;
    const bool ok = i < n_params;
    float s = 0.f;
    if (ok) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int w = ph;
        for (; w + 24 < nparts; w += 32) {
            s0 += partials[(size_t)w * n_params + i];
            s1 += partials[(size_t)(w + 8) * n_params + i];
            s2 += partials[(size_t)(w + 16) * n_params + i];
            s3 += partials[(size_t)(w + 24) * n_params + i];
        }
        for (; w < nparts; w += 8) s0 += partials[(size_t)w * n_params + i];
        s = (s0 + s1) + (s2 + s3);
        // is i an entry of W0[:,1:] or b0 ?  those come from the dw0 partials
        int o = -1;
        if (i >= poffW0 && i < poffW0 + H1 * (1 + E)) {
            const int f = (i - poffW0) / (1 + E), col = (i - poffW0) - f * (1 + E);
            if (col >= 1) o = f * (E + 1) + (col - 1);
        } else if (i >= poffb0 && i < poffb0 + H1) {
            o = (i - poffb0) * (E + 1) + E;
        }
        if (o >= 0) {
            const int nout = H1 * (E + 1);
            float t0 = 0.f, t1 = 0.f;
            int w2 = ph;
            for (; w2 + 8 < nparts0; w2 += 16) { t0 += p0[(size_t)w2 * nout + o]; t1 += p0[(size_t)(w2 + 8) * nout + o]; }
            for (; w2 < nparts0; w2 += 8) t0 += p0[(size_t)w2 * nout + o];
            s += t0 + t1;
        }
    }
    red[ph][il] = s;
    __syncthreads();
    if (ph == 0 && ok) {
        float v = red[0][il];
#pragma unroll
        for (int k = 1; k < 8; ++k) v += red[k][il];
        dtheta[i] = v;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
typedef void (*bwd_kernel_t)(const BwdArgs);
struct BwdVariant { int tmax, nacc, edge, ksc; bwd_kernel_t fn; const char* name; };
#define BWD_VARIANT(T, N, E, K) { T, N, E, K, cc_bwd_kernel<T, N, (E) != 0, K>, "cc_bwd<T=" #T ",NACC=" #N ",EDGE=" #E ",KS=" #K ">" }
static const BwdVariant kBwdVariants[] = {
    BWD_VARIANT(4, 3, 1, 13), BWD_VARIANT(4, 3, 0, 13),     // exact: every hidden layer 48..51 wide (UCI / VAE nets)
    BWD_VARIANT(2, 3, 1, 0), BWD_VARIANT(2, 3, 0, 0),
    BWD_VARIANT(4, 3, 1, 0), BWD_VARIANT(4, 3, 0, 0),
    BWD_VARIANT(5, 2, 1, 17), BWD_VARIANT(5, 3, 0, 17),     // exact: every hidden layer 64 wide
    BWD_VARIANT(7, 1, 1, 26), BWD_VARIANT(7, 0, 1, 26), BWD_VARIANT(7, 1, 0, 26),   // exact: every hidden layer 100 wide (toy flows, MonotonicNN): no spills
    BWD_VARIANT(7, 0, 1, 0), BWD_VARIANT(7, 1, 0, 0),
    BWD_VARIANT(8, 0, 1, 0), BWD_VARIANT(8, 1, 0, 0),
    BWD_VARIANT(8, 0, 1, 32), BWD_VARIANT(8, 1, 0, 32),     // exact: every hidden layer 124..127 wide, or zero-padded to that (edge pass without dW: no spills; 51 in the others)
};

static const BwdVariant* find_bwd(int tmax, int nacc, int edge, int ksu) {
    for (int exact = 1; exact >= 0; --exact)
        for (const BwdVariant& v : kBwdVariants)
            if (v.tmax == tmax && v.nacc == nacc && v.edge == edge && (exact ? (v.ksc && v.ksc == ksu) : !v.ksc)) return &v;
    return nullptr;
}
// template tile count: the net's own when a shape-exact family exists for it, else the next generic bucket
static int pick_tmax_bwd(int tmax, int ksu) {
    if (ksu)
        for (const BwdVariant& v : kBwdVariants)
            if (v.ksc == ksu && v.tmax == tmax) return tmax;
    return tmax <= 2 ? 2 : tmax <= 4 ? 4 : tmax <= 7 ? 7 : 8;
}
// dW layers a pass can hold: the largest NACC instantiated for (T, edge), preferring the shape-exact family
static int best_nacc(int T, int edge, int ksu) {
    for (int exact = 1; exact >= 0; --exact)
        for (int n = 3; n >= 0; --n)
            for (const BwdVariant& v : kBwdVariants)
                if (v.tmax == T && v.nacc == n && v.edge == edge && (exact ? (v.ksc && v.ksc == ksu) : !v.ksc)) return n;
    return -1;
}

// Nets with UNEQUAL hidden widths above 63 units (5..7 tiles) have no shape-exact variant of their own, and the generic
// variants with runtime tile counts spill hundreds of registers there (687 ms per call at 256 x 784 integrals).  Pad them --
// virtually: only the per-layer tile and K-step counts change, the staged images carry the zeros and the constant-one feature
// of every layer stays at its own index width[l] -- to the smallest shape-exact family that holds the widest layer: (5 tiles,
// 17 K-steps: widths up to 67) or (7, 26: up to 103).  A padded unit has zero weights and bias, so z = 0, act(0) = 0, and it
// contributes exact zeros to every sum; d_theta is written for the real entries only.  Returns 1 if the net was padded.
static int pad_to_exact_family(MlpDev& m, int* tmax, int* ksu) {
    if (*tmax <= 4) return 0;
    if (*ksu)
        for (const BwdVariant& v : kBwdVariants)
            if (v.ksc == *ksu && v.tmax == *tmax) return 0;
    const int L = m.n_linear - 1;
    int ksmax = 0;
    for (int l = 1; l <= L; ++l) ksmax = m.ks_in[l] > ksmax ? m.ks_in[l] : ksmax;
    static const int kFamilies[][2] = {{5, 17}, {7, 26}, {8, 32}};
    for (const auto& f : kFamilies) {
        if (*tmax > f[0] || ksmax > f[1]) continue;
        for (int l = 1; l <= L; ++l) { m.t_out[l] = m.t_mfma[l] = f[0]; m.ks_in[l] = f[1]; }
        int off = 0;
        for (int l = 1; l < L; ++l) { m.lds_off[l] = off; off += m.t_out[l + 1] * m.ks_in[l] * 64; }
        m.lds_off[L] = off;
        *tmax = f[0]; *ksu = f[1];
        return 1;
    }
    return 0;
}

// row stride for the row-major image: >= cols, minimising bank conflicts of both fragment shapes
// (forward: 16 rows x 2 adjacent cols per half-wave; backward: 2 adjacent rows x 16 cols)
static int pick_ld(int cols) {
    int best = cols, best_cost = 1 << 30;
    for (int ld = cols; ld < cols + 33; ++ld) {
        int cost = 0;
        for (int pattern = 0; pattern < 2; ++pattern) {
            int cnt[32] = {0};
            for (int a16 = 0; a16 < 16; ++a16)
                for (int b2 = 0; b2 < 2; ++b2) {
                    const int addr = pattern == 0 ? a16 * ld + b2 : b2 * ld + a16;
                    cnt[((addr % 32) + 32) % 32]++;
                }
            int worst = 0;
            for (int b = 0; b < 32; ++b) worst = cnt[b] > worst ? cnt[b] : worst;
            cost += worst;
        }
        if (cost < best_cost) { best_cost = cost; best = ld; }
    }
    return best;
}

struct BwdPlan {
    BwdArgs a;
    int tmax;          // template tile count
    int ksu;           // common K-step count when every hidden layer fills exactly `tmax` tiles, else 0
    int nwaves, nblocks, wpb;
    int ns;            // node-range split of the fp32 kernels (small batches: fewer tiles than waves)
    size_t lds_bytes_for(int nacc, int waves) const { return (size_t)(a.scratch_off + waves * (nacc + 1) * tmax * 256) * sizeof(float); }
    long long ws_partials, ws_dc, ws_p0;   // byte offsets in the workspace
    long long ws_scal;                     // 256 B of launch scalars (cc_bwd_ws16_kernel.h)
    long long ws_front, ws_front_bytes;    // HBM scratch of the staged backward (wide first hidden layer), 0 if not that family
    long long ws_total;
    int nparts0, chunk0;
};

static int plan_backward_impl(const umnn_mlp* net, long long B, int d, int E, BwdPlan* pl, bool allow_pad, int* padded);
static int plan_backward(const umnn_mlp* net, long long B, int d, int E, BwdPlan* pl, int* padded_out = nullptr) {
    int padded = 0;
    int rc = plan_backward_impl(net, B, d, E, pl, true, &padded);
    if (rc == UMNN_EUNSUPPORTED && padded) {              // (the padded images of a deep net can exceed the LDS: the net's own counts then)
        padded = 0;
        rc = plan_backward_impl(net, B, d, E, pl, false, &padded);
    }
    if (padded_out) *padded_out = padded;
    return rc;
}

static int plan_backward_impl(const umnn_mlp* net, long long B, int d, int E, BwdPlan* pl, bool allow_pad, int* padded) {
    int tmax = 0, ksu = 0;
    if (int rc = umnn_prepare_mlp(net, E, &pl->a.m, &tmax, &ksu)) return rc;
    BwdArgs& a = pl->a;
    const int L = a.m.n_linear - 1;
    // (the three-stage kernels of cc_backward_front.hip take their shape first: they read the unpadded tile counts)
    const bool front = umnn_backward_front_shape(a.m);
    *padded = (!front && allow_pad) ? pad_to_exact_family(a.m, &tmax, &ksu) : 0;
    pl->tmax = pick_tmax_bwd(tmax, ksu);
    pl->ksu = (ksu && tmax == pl->tmax) ? ksu : 0;
    int off = 0;
    for (int l = 1; l < L; ++l) {
        a.ld[l] = pick_ld(4 * a.m.ks_in[l]);
        a.roff[l] = off;
        off += 4 * a.m.ks_in[l + 1] * a.ld[l];
        off = (off + 3) & ~3;
    }
    a.scratch_off = off;
    int po = 0;
    for (int l = 0; l <= L; ++l) {
        a.poffW[l] = po; po += net->widths[l + 1] * net->widths[l];
        a.poffb[l] = po; po += net->widths[l + 1];
    }
    a.n_params = po;
    a.NI = B * (long long)d; a.d = d; a.E = E;
    a.ngroups = (unsigned)((a.NI + 15) / 16);
    // waves per workgroup: as many (4, 2, 1) as leave room for the wave-private transpose tiles
    const int nm = best_nacc(pl->tmax, 1, pl->ksu), nr = best_nacc(pl->tmax, 0, pl->ksu);
    const int nacc_max = nm > nr ? nm : nr;
    pl->wpb = 4;
    while (pl->wpb > 1 && pl->lds_bytes_for(nacc_max, pl->wpb) > 160 * 1024) pl->wpb >>= 1;
    if (pl->lds_bytes_for(nacc_max, pl->wpb) > 160 * 1024)
        return umnn_fail(UMNN_EUNSUPPORTED, "backward: weight images exceed 160 KiB of LDS");
    // fewer tiles than waves (the reference's own training batches are 100 samples): share a tile's nodes among
    // up to 32 waves; every work item keeps its own d_theta / dc partial, summed in a fixed order afterwards
    pl->ns = 1;
    {
        const long long waves = (long long)umnn_num_cus() * pl->wpb;
        if ((long long)a.ngroups * 2 <= waves) pl->ns = (int)(waves / a.ngroups > 32 ? 32 : waves / a.ngroups);
        if (const int v = umnn_options().bwd_ns; v >= 1 && v <= 32) pl->ns = v;
    }
    const long long items = (long long)a.ngroups * pl->ns;
    pl->nblocks = umnn_num_cus();
    if ((long long)pl->nblocks * pl->wpb > items) pl->nblocks = (int)((items + pl->wpb - 1) / pl->wpb);
    if (pl->nblocks < 1) pl->nblocks = 1;
    pl->nwaves = pl->nblocks * pl->wpb;
    a.ns = 1;
    const int H1 = net->widths[1];
    pl->chunk0 = 64;
    while (pl->chunk0 > 16 && dw0_lds_floats(pl->chunk0, H1, E) * sizeof(float) > 32 * 1024) pl->chunk0 /= 2;
    while (pl->chunk0 > 16 && a.NI / pl->chunk0 < 2LL * umnn_num_cus()) pl->chunk0 /= 2;     // small batches: more, smaller chunks
    {   // persistent blocks (four per CU), each accumulating its chunks in registers: at most that many partial slices
        const long long nchunks = (a.NI + pl->chunk0 - 1) / pl->chunk0, cap = 4LL * umnn_num_cus();
        pl->nparts0 = (int)(nchunks < cap ? nchunks : cap);
    }
    long long o = 0;
    pl->ws_partials = o; o += (long long)pl->nwaves * a.n_params * 4; o = (o + 255) & ~255LL;
    pl->ws_dc = o; o += a.NI * H1 * 4 * pl->ns; o = (o + 255) & ~255LL;
    pl->ws_p0 = o; o += (long long)pl->nparts0 * H1 * (E + 1) * 4; o = (o + 255) & ~255LL;
    pl->ws_scal = o; o += 256;
    pl->ws_front = o;
    pl->ws_front_bytes = (umnn_backward_front_shape(a.m) && pl->wpb == 4)
                             ? umnn_backward_front_scratch_bytes(a.m, a.NI) : 0;
    o += pl->ws_front_bytes; o = (o + 255) & ~255LL;
    pl->ws_total = o;
    return 0;
}

extern "C" long long umnn_cc_backward_workspace_bytes(const umnn_mlp* net, long long B, int d, int E) {
    BwdPlan pl;
    if (B <= 0) return 0;
    if (plan_backward(net, B, d, E, &pl)) return -1;
    return pl.ws_total;
}


int umnn_launch_backward_bf16(const BwdArgs& base, const umnn_mlp* net, int nblocks_max, int* nwaves_out, hipStream_t stream);

// Which kernel family umnn_cc_backward would use for this net: 1 = shape-exact kernels (compile-time tile counts),
// 0 = generic kernels with at most four tiles per layer (runtime guards, ~2.5x slower), -1 = generic kernels with more
// than four tiles -- those spill hundreds of registers per lane and are kept for completeness only: the Python host
// routes such nets to the materialised ATen chain on the GPU instead.  Since round 3 that is only what is left when the zero-padded
// weight images of a deep wide net do not fit the LDS: unequal widths of 64..127 run zero-padded on the 5- / 7- / 8-tile
// shape-exact families (pad_to_exact_family) and report 1.  < -1: error code.
extern "C" int umnn_cc_backward_kind(const umnn_mlp* net, int E) {
    MlpDev m;
    int tmax = 0, ksu = 0;
    if (int rc = umnn_prepare_mlp(net, E, &m, &tmax, &ksu)) return rc < -1 ? rc : -2;
    // wide first hidden layer + narrow rest (MNISTExperiment's 100-50-50-50-50): the three-stage kernels of
    // cc_backward_front.hip (bwd_precision = fp32: their six-term build, cc_backward_front_p3.hip)
    if (umnn_backward_front_shape(m)) return 1;
    {   // unequal widths of 5..8 tiles run zero-padded on a shape-exact family, if the padded weight images fit the LDS
        BwdPlan pl;
        int padded = 0;
        if (plan_backward(net, 16, 1, E, &pl, &padded) == 0 && padded) pad_to_exact_family(m, &tmax, &ksu);
    }
    const int T = pick_tmax_bwd(tmax, ksu);
    const int ks = (ksu && tmax == T) ? ksu : 0;
    const int nm = best_nacc(T, 1, ks);
    if (ks && nm >= 0 && find_bwd(T, nm, 1, ks)->ksc) return 1;
    return T <= 4 ? 0 : -1;
}

extern "C" int umnn_cc_backward(const umnn_mlp* net, const float* x0, const float* x, const float* h,
                                const float* g, const float* g_fx,
                                const float* cc_w, const float* cc_s, int nb_steps, long long B, int d, int E,
                                float* dx0, float* dx, float* dh, float* dtheta,
                                void* workspace, long long workspace_bytes, void* stream_) {
    return umnn_cc_backward_io(net, nullptr, x0, x, h, g, g_fx, cc_w, cc_s, nb_steps, B, d, E, 0, dx0, dx, dh, dtheta,
                               workspace, workspace_bytes, stream_);
}

static int backward_impl(const umnn_mlp* net, const umnn_io* io, const void* x0_, const void* x_, const void* h_,
                         const void* g_, const void* g_fx_,
                         const float* cc_w, const float* cc_s, int nb_steps, long long B, int d, int E, int inv_f,
                         void* dx0_, void* dx_, void* dh_, float* dtheta,
                         void* workspace, long long workspace_bytes, void* stream_, const float* z2_saved);
extern "C" int umnn_cc_backward_io(const umnn_mlp* net, const umnn_io* io, const void* x0_, const void* x_, const void* h_,
                                   const void* g_, const void* g_fx_,
                                   const float* cc_w, const float* cc_s, int nb_steps, long long B, int d, int E, int inv_f,
                                   void* dx0_, void* dx_, void* dh_, float* dtheta,
                                   void* workspace, long long workspace_bytes, void* stream_) {
    return backward_impl(net, io, x0_, x_, h_, g_, g_fx_, cc_w, cc_s, nb_steps, B, d, E, inv_f, dx0_, dx_, dh_, dtheta, workspace,
                         workspace_bytes, stream_, nullptr);
}
// umnn_cc_backward (lower limit 0, fp32 storage) with the z_2 buffer umnn_flow_stack_block_forward_save left: the three-stage family
// skips its stage A.  z2_floats must be what umnn_cc_forward_z2_floats returned for the same (net, B, d, E, nb_steps).
extern "C" int umnn_cc_backward_saved(const umnn_mlp* net, const float* x, const float* h, const float* g, const float* g_fx,
                                      const float* cc_w, const float* cc_s, int nb_steps, long long B, int d, int E,
                                      float* dx, float* dh, float* dtheta, const float* z2_saved, long long z2_floats,
                                      void* workspace, long long workspace_bytes, void* stream_) {
    // `need` follows the CURRENT process-wide precision options: if they changed between the training forward and this call (need == 0:
    // the pair no longer applies), the buffer is simply ignored and the backward recomputes z_2 -- as the comment in backward_impl says
    const long long need = umnn_cc_forward_z2_floats(net, B, d, E, nb_steps);
    if (need == 0) z2_saved = nullptr;
    else if (!z2_saved || z2_floats < need)
        return umnn_fail(UMNN_EINVAL, "backward (z_2 saved): buffer missing or smaller than umnn_cc_forward_z2_floats()");
    return backward_impl(net, nullptr, nullptr, x, h, g, g_fx, cc_w, cc_s, nb_steps, B, d, E, 0, nullptr, dx, dh, dtheta, workspace,
                         workspace_bytes, stream_, z2_saved);
}
extern "C" int umnn_cc_backward_saved_io(const umnn_mlp* net, const umnn_io* io, const void* x, const void* h, const void* g, const void* g_fx,
                                         const float* cc_w, const float* cc_s, int nb_steps, long long B, int d, int E,
                                         void* dx, void* dh, float* dtheta, const float* z2_saved, long long z2_floats,
                                         void* workspace, long long workspace_bytes, void* stream_) {
    const long long need = umnn_cc_forward_z2_floats(net, B, d, E, nb_steps);
    if (need == 0) z2_saved = nullptr;
    else if (!z2_saved || z2_floats < need)
        return umnn_fail(UMNN_EINVAL, "backward (z_2 saved): buffer missing or smaller than umnn_cc_forward_z2_floats()");
    return backward_impl(net, io, nullptr, x, h, g, g_fx, cc_w, cc_s, nb_steps, B, d, E, 0, nullptr, dx, dh, dtheta, workspace,
                         workspace_bytes, stream_, z2_saved);
}
static int backward_impl(const umnn_mlp* net, const umnn_io* io, const void* x0_, const void* x_, const void* h_,
                         const void* g_, const void* g_fx_,
                         const float* cc_w, const float* cc_s, int nb_steps, long long B, int d, int E, int inv_f,
                         void* dx0_, void* dx_, void* dh_, float* dtheta,
                         void* workspace, long long workspace_bytes, void* stream_, const float* z2_saved) {
    const float *x0 = (const float*)x0_, *x = (const float*)x_, *h = (const float*)h_, *g = (const float*)g_, *g_fx = (const float*)g_fx_;
    float *dx0 = (float*)dx0_, *dx = (float*)dx_, *dh = (float*)dh_;
    if (int rc = umnn_check_io(io)) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (B < 0 || d < 1) return umnn_fail(UMNN_EINVAL, "backward: B must be >= 0 and d >= 1");
    if (nb_steps < 1) return umnn_fail(UMNN_EINVAL, "backward: nb_steps must be >= 1");
    BwdPlan pl;
    if (B == 0) {
        if (!net) return umnn_fail(UMNN_EINVAL, "net is null");
        if (dtheta) return umnn_check(hipMemsetAsync(dtheta, 0, umnn_param_count(net) * sizeof(float), stream), "memset");
        return 0;
    }
    if (int rc = plan_backward(net, B, d, E, &pl)) return rc;
    if (!x || !h || !g || !cc_w || !cc_s) return umnn_fail(UMNN_EINVAL, "backward: x, h, g, cc_w, cc_s must be non-null");
    if (!workspace || workspace_bytes < pl.ws_total)
        return umnn_fail(UMNN_EINVAL, "backward: workspace smaller than umnn_cc_backward_workspace_bytes()");
    BwdArgs& a = pl.a;
    const int L = a.m.n_linear - 1, H1 = net->widths[1];
    char* ws = (char*)workspace;
    a.x0 = x0; a.x = x; a.h = h; a.g = g; a.gfx = g_fx; a.ccw = cc_w; a.ccs = cc_s;
    a.dx0 = dx0; a.dx = dx; a.n = nb_steps;
    a.x_bf16 = io && io->x_dtype == UMNN_DTYPE_BF16; a.h_bf16 = io && io->h_dtype == UMNN_DTYPE_BF16;
    a.inv_f = inv_f != 0;
    a.partials = (float*)(ws + pl.ws_partials);
    a.dc = (float*)(ws + pl.ws_dc);
    a.scal = (unsigned*)(ws + pl.ws_scal);
    a.z2_saved = (pl.ws_front_bytes > 0 && umnn_options().bwd_precision == UMNN_PRECISION_BF16X3) ? z2_saved : nullptr;
    // (a call the three-stage kernels do not serve -- tiny batches, bwd_precision fp32 -- simply recomputes z_2: the buffer is ignored)
    float* p0 = (float*)(ws + pl.ws_p0);
    if (int rc = umnn_check(hipMemsetAsync(a.partials, 0, (size_t)pl.nwaves * a.n_params * 4, stream), "memset partials")) return rc;

    // ---- bf16-split kernels (default) where the shape allows; otherwise / on request the fp32-MFMA kernels below
    bool done = false;
    int ns_used = pl.ns > nb_steps + 1 ? nb_steps + 1 : pl.ns;
    a.ns = ns_used;
    if (pl.ws_front_bytes > 0) {
        a.ns = 1;
        const int rc = umnn_options().bwd_precision == UMNN_PRECISION_BF16X3
                           ? umnn_launch_backward_front(a, net, pl.nblocks, ws + pl.ws_front, pl.ws_front_bytes, stream)
                           : umnn_launch_backward_front_p3(a, net, pl.nblocks, ws + pl.ws_front, pl.ws_front_bytes, stream);
        if (rc == 0) { done = true; ns_used = 1; }
        else if (rc != UMNN_EUNSUPPORTED) return rc;
        else a.ns = ns_used;
    }
    int nslices = pl.nwaves;              // d_theta slices the main pass wrote (the reduction reads no more than that)
    if (!done && umnn_options().bwd_precision == UMNN_PRECISION_BF16X3) {
        int nw = 0;
        const int rc = umnn_launch_backward_bf16(a, net, pl.nblocks, &nw, stream);
        if (rc == 0) { done = true; if (nw > 0 && nw < nslices) nslices = nw; }
        else if (rc != UMNN_EUNSUPPORTED) return rc;
    }
    // ---- passes: the EDGE pass (with as many dW layers as its variant holds), then the remaining layers
    const int T = pl.tmax;
    const int nacc_main = best_nacc(T, 1, pl.ksu), nacc_rest = best_nacc(T, 0, pl.ksu);
    if (nacc_main < 0 || nacc_rest < 1) return umnn_fail(UMNN_EUNSUPPORTED, "backward: no kernel variant for this width");
    int l_next = 1;
    for (int pass = 0; !done && (pass == 0 || l_next < L); ++pass) {
        const int nacc = pass == 0 ? nacc_main : nacc_rest;
        const BwdVariant* v = find_bwd(T, nacc, pass == 0 ? 1 : 0, pl.ksu);
        if (!v) return umnn_fail(UMNN_EUNSUPPORTED, "backward: no kernel variant for this width");
        a.l_lo = l_next;
        a.scratch_per_wave = (nacc + 1) * T * 256;
        const size_t lds_bytes = pl.lds_bytes_for(nacc, pl.wpb);
        if (lds_bytes > 160 * 1024) return umnn_fail(UMNN_EUNSUPPORTED, "backward: weight images exceed 160 KiB of LDS");
        if (int rc = umnn_allow_lds((const void*)v->fn, lds_bytes)) return rc;
        umnn_prof_begin(stream);
        hipLaunchKernelGGL(v->fn, dim3(pl.nblocks), dim3(64 * pl.wpb), lds_bytes, stream, a);
        umnn_prof_end(stream, pass == 0 ? 3.0 * umnn_cc_forward_flops_per_integral(net, nb_steps) * (double)a.NI : 0.0, UMNN_PROF_BACKWARD);
        umnn_note_launch(v->name);
        if (int rc = umnn_check(hipGetLastError(), "cc_bwd launch")) return rc;
        l_next += nacc;
    }

    // ---- finishing kernels
    umnn_prof_begin(stream);
    if (ns_used > 1) {      // node-split partials of dc -> one sum, before anything reads dc
        const long long count = a.NI * H1;
        hipLaunchKernelGGL(cc_bwd_dcsum_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, a.dc, count, ns_used);
    }
    if (dh) {
        const long long groups = (a.NI + 15) / 16;
        if (E <= 80 && groups >= 8LL * umnn_num_cus()) {              // large batches: the matrix-core version (persistent waves)
            const size_t sm = (size_t)((H1 + 3) / 4) * ((E + 15) / 16) * 64 * sizeof(float);
            const long long want = (groups + 3) / 4, cap = 8LL * umnn_num_cus();
            const unsigned nb = (unsigned)(want < cap ? want : cap);
            if (int rc = umnn_allow_lds((const void*)cc_bwd_dh_mfma_kernel, sm)) return rc;
            hipLaunchKernelGGL(cc_bwd_dh_mfma_kernel, dim3(nb), dim3(256), sm, stream, a.dc, net->W[0], dh, a.NI, d, E, H1, a.h_bf16);
            umnn_note_launch("cc_bwd_dh<mfma>");
        } else {
        const int qpb = a.NI / 64 < 2LL * umnn_num_cus() ? 16 : 64;      // integrals per workgroup
        const size_t sm = ((size_t)H1 * E + qpb * (H1 + 1)) * sizeof(float);
        const unsigned nb = (unsigned)((a.NI + qpb - 1) / qpb);
        if (int rc = umnn_allow_lds((const void*)cc_bwd_dh_kernel, sm)) return rc;
        hipLaunchKernelGGL(cc_bwd_dh_kernel, dim3(nb), dim3(256), sm, stream, a.dc, net->W[0], dh, a.NI, d, E, H1, qpb, a.h_bf16);
        umnn_note_launch("cc_bwd_dh");
        }
    }
    if (dtheta) {
        const size_t sm = dw0_lds_floats(pl.chunk0, H1, E) * sizeof(float);
        const int tiles_per_wave = (((H1 + 15) / 16) * ((E + 1 + 15) / 16) + 3) / 4;
        auto dw0 = tiles_per_wave <= 2 ? cc_bwd_dw0_kernel<2> : tiles_per_wave <= 4 ? cc_bwd_dw0_kernel<4> : cc_bwd_dw0_kernel<10>;
        const int per_launch = 4 * (tiles_per_wave <= 2 ? 2 : tiles_per_wave <= 4 ? 4 : 10);
        if (int rc = umnn_allow_lds((const void*)dw0, sm)) return rc;
        for (int tb = 0; tb < 4 * tiles_per_wave; tb += per_launch)
            hipLaunchKernelGGL(dw0, dim3(pl.nparts0), dim3(256), sm, stream, a.dc, h, p0, a.NI, d, E, H1, pl.chunk0, a.h_bf16, tb);
        hipLaunchKernelGGL(cc_bwd_reduce_kernel, dim3((a.n_params + 31) / 32), dim3(256), 0, stream,
                           a.partials, nslices, a.n_params, p0, pl.nparts0, E, H1, a.poffW[0], a.poffb[0], dtheta);
        umnn_note_launch("cc_bwd_reduce");
    }
    umnn_prof_end(stream, 0.0, UMNN_PROF_FINISH);
    return umnn_check(hipGetLastError(), "cc_bwd finishing launch");
}
