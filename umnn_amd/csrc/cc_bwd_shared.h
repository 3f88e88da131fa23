// Shared by the backward kernels (fp32-MFMA and bf16-split): launch arguments.
#pragma once
#include "cc_host.h"

struct BwdArgs {
    MlpDev m;
    int ld[UMNN_MAX_LINEAR];        // row stride (floats) of the row-major image of hidden layer l -> l+1
    int roff[UMNN_MAX_LINEAR];      // float offset of that image in LDS
    int poffW[UMNN_MAX_LINEAR];     // offset of W_l / b_l in the flat theta vector
    int poffb[UMNN_MAX_LINEAR];
    const float* x0;
    const float* x;
    const float* h;
    const float* g;
    const float* gfx;               // nullable
    const float* ccw;
    const float* ccs;
    float* dx0;                     // nullable
    float* dx;                      // nullable
    float* dc;                      // [NI][H1] (EDGE pass)
    float* partials;                // [nwaves][n_params]
    int x_bf16, h_bf16;             // storage of x, x0, g, g_fx, dx, dx0 / of h (and dh): 0 fp32, 1 bf16
    int inv_f;                      // the quadrature integrated 1/f (ParallelNeuralIntegral.py:58-59,70-72): d_theta and d_h
                                    // differentiate 1/f -- node cotangent x (-1/f^2); the Leibniz terms keep f (:120-123)
    long long NI;
    int d, E, n;
    unsigned ngroups;               // tiles of 16 integrals
    int ns;                         // node-range split: a tile's nodes are shared by ns work items (small batches);
                                    // part j writes its dc partial at dc + j*NI*H1 (summed by the finishing kernels)
    int l_lo;                       // this pass accumulates dW for hidden layers l_lo .. l_lo+NACC-1
    int n_params;
    int scratch_off;                // float offset of the per-wave scratch region in LDS
    int scratch_per_wave;           // floats
    const float* z2_saved;          // nullable: hidden layer 2's pre-activations of every node, left by the training forward
                                    // (umnn_flow_stack_block_forward_save) -- the three-stage backward then skips its stage A
    unsigned* scal;                 // 256 B of launch scalars in the workspace (cotangent scale / overflow flag of cc_bwd_ws16_kernel.h)
};

// permutation that turns an accumulator row (lane&15 in an A operand) into a feature offset inside a tile
__device__ __forceinline__ int perm16(int rho) { return 4 * (rho & 3) + (rho >> 2); }


// staged backward for nets with a wide first hidden layer (cc_backward_front.hip)
int umnn_backward_front_shape(const MlpDev& m);
long long umnn_backward_front_scratch_bytes(const MlpDev& m, long long NI);
int umnn_launch_backward_front(const BwdArgs& base, const umnn_mlp* net, int nblocks_max, void* scratch, long long scratch_bytes,
                               hipStream_t stream);
// (the same with three pieces in every product: cc_backward_front_p3.hip, bwd_precision = fp32)
int umnn_launch_backward_front_p3(const BwdArgs& base, const umnn_mlp* net, int nblocks_max, void* scratch, long long scratch_bytes,
                                  hipStream_t stream);
