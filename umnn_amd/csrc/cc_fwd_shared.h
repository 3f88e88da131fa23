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
    long long NI;      // B*d integrals
    int d, E, n, ns, inv_f;
    unsigned ngroups;  // tile groups (of 16*P integrals)
};

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
    if (live && part == 0 && g == 0) {
#pragma unroll
        for (int pt = 0; pt < P; ++pt) {
            if (!ok[pt]) continue;
            const long long q = qv[pt];
            const float Fv = Facc[pt] * dxv[pt] * 0.5f;
            if (a.F) a.F[q] = Fv;
            if (a.fx) a.fx[q] = fxv[pt];
            if (a.fx0) a.fx0[q] = fx0v[pt];
            if (a.scaling) {
                const long long bi = q / d;
                const int i = (int)(q - bi * d);
                const float sc = a.scaling[i];
                const float z0 = a.h[bi * ((long long)E * d) + i];
                a.z[a.reverse_z ? bi * d + (d - 1 - i) : q] = __expf(sc) * (Fv + z0);
                const float lj = __logf(fxv[pt] + 1e-10f) + sc;
                a.logjac[q] = a.logjac_in ? a.logjac_in[q] + lj : lj;
            }
        }
    }
}

