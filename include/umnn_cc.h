/*
 * umnn_cc.h -- C ABI of the MI355X (gfx950) Clenshaw-Curtis neural-integration library.
 *
 * The reference (AWehenkel/UMNN, pure Python) has no FFI; its operator boundary for this
 * path is the Python call
 *
 *     ParallelNeuralIntegral.apply(x0, x, integrand, flat_params, h, nb_steps, inv_f)
 *         models/UMNN/ParallelNeuralIntegral.py:97-123   (forward :100-108, backward :110-123)
 *     NeuralIntegral.apply(x0, x, integrand, flat_params, h, nb_steps)
 *         models/UMNN/NeuralIntegral.py:78-99
 *     integrate(x0, nb_steps, step_sizes, integrand, h, compute_grad, x_tot, inv_f, cc_weights, steps)
 *         models/UMNN/ParallelNeuralIntegral.py:37-80,  models/UMNN/NeuralIntegral.py:37-66
 *     compute_cc_weights(nb_steps)
 *         models/UMNN/ParallelNeuralIntegral.py:14-34
 *
 * and, one level up, the per-block flow arithmetic of models/UMNN/UMNNMAF.py:76-139
 * (z = exp(scaling) * (integral + h[:,0,:]),  log_jac = log(f(x;h) + 1e-10) + scaling).
 *
 * Each entry point below replaces one of those calls when the integrand is an MLP
 * (models/UMNN/UMNNMAF.py:235-284 IntegrandNetwork, models/UMNN/MonotonicNN.py:12-27
 * IntegrandNN).  Plain pointers and sizes only: every pointer is a DEVICE pointer to fp32
 * data owned by the caller (PyTorch's caching allocator in the shipped host code) unless the
 * name ends in _host.  Work is enqueued on `stream` (a hipStream_t passed as void*; NULL =
 * the default stream) and the call returns without synchronising.
 *
 * Layouts (identical to the reference tensors, contiguous, row-major):
 *     x0, x, F, f_x, f_x0, g, dx, dx0, z, log_jac   [B, d]
 *     h, dh                                         [B, E*d], element (b, e*d + i)  (UMNNMAF.py:279-281)
 *     W[l]  [widths[l+1], widths[l]]   b[l]  [widths[l+1]]     (torch.nn.Linear layout)
 *     cc_w, cc_s                                    [nb_steps+1]
 *
 * Return value: 0 on success, otherwise a negative UMNN_E* code or a positive hipError_t;
 * umnn_last_error() gives the message (thread-local).
 */
#ifndef UMNN_CC_H
#define UMNN_CC_H

#ifdef __cplusplus
extern "C" {
#endif

#define UMNN_MAX_LINEAR 8            /* Linear layers in the integrand MLP (hidden layers + 1) */
#define UMNN_MAX_HIDDEN_WIDTH 127    /* widest hidden layer the LDS-resident kernels accept */

#define UMNN_ACT_LEAKY_RELU 0        /* nn.LeakyReLU(0.01): IntegrandNetwork, UMNNMAF.py:247 */
#define UMNN_ACT_RELU 1              /* nn.ReLU: IntegrandNN, MonotonicNN.py:20 */
#define UMNN_OUT_ELU_PLUS_ONE 0      /* ELU(.)+1: ELUPlus UMNNMAF.py:11-16 / MonotonicNN.py:23,27 */
#define UMNN_OUT_SIGMOID 1           /* nn.Sigmoid: dict_act_func UMNNMAF.py:19 */

#define UMNN_EINVAL (-1)             /* bad argument (message says which) */
#define UMNN_EUNSUPPORTED (-2)       /* shape outside what the kernels cover */
#define UMNN_ENODEVICE (-3)          /* no gfx950 device / kernel image not loadable */

typedef struct umnn_mlp {
    int n_linear;                         /* number of Linear layers */
    int widths[UMNN_MAX_LINEAR + 1];      /* [1+E, H1, ..., HL, 1] */
    const float* W[UMNN_MAX_LINEAR];      /* device pointers */
    const float* b[UMNN_MAX_LINEAR];
    int hidden_act;                       /* UMNN_ACT_* */
    int out_act;                          /* UMNN_OUT_* */
} umnn_mlp;

/* Storage types of the activation tensors (configuration C4, the bf16 VAE prior flow: models/vae_lib/models/flows.py:305-323).
 * The arithmetic is fp32 in every case; the *_io entry points take a descriptor saying how the caller STORES
 *   x-class tensors: x0, x, F, f_x, f_x0, z, log_jac, log_jac_in, g, g_fx, dx, dx0      (x_dtype)
 *   h-class tensors: h (the [B, E*d] embedding the conditioner writes -- the large one), dh   (h_dtype)
 * bf16 loads are exact widenings, bf16 stores round to nearest even.  io == NULL means fp32 everywhere (= the plain
 * entry points).  d_theta, the quadrature tables, scaling and the weights are always fp32. */
#define UMNN_DTYPE_F32 0
#define UMNN_DTYPE_BF16 1
typedef struct umnn_io {
    int x_dtype;
    int h_dtype;
} umnn_io;

/* Replaces integrate(..., compute_grad=False) -- ParallelNeuralIntegral.py:49-65 and
 * NeuralIntegral.py:53-66 (both solvers are the same arithmetic; the kernel never
 * materialises the node axis).  Quadrature node 0 is x and node n is x0, so the same pass
 * also yields f(x;h) and f(x0;h) (what backward's Leibniz terms :117-118 and
 * UMNNMAF.compute_log_jac :138 re-evaluate in the reference).
 *   x0     may be NULL (= zeros, the only value UMNNMAF/MonotonicNN ever pass)
 *   inv_f  != 0 integrates 1/f (ParallelNeuralIntegral.py:58-59)
 *   f_x, f_x0 may be NULL. */
int umnn_cc_forward(const umnn_mlp* net, const float* x0, const float* x, const float* h,
                    const float* cc_w, const float* cc_s, int nb_steps,
                    long long B, int d, int E, int inv_f,
                    float* F, float* f_x, float* f_x0, void* stream);

/* Replaces one UMNNMAF block's arithmetic after the conditioner (UMNNMAF.py:80-83,134,138-139):
 *   z = exp(scaling_i) * (int_0^x f + h[b, 0*d+i]),  log_jac = log(f(x;h) + 1e-10) + scaling_i.
 * f_x / f_x0 (nullable) are also written so the autograd wrapper need not recompute them. */
int umnn_flow_block_forward(const umnn_mlp* net, const float* x, const float* h, const float* scaling,
                            const float* cc_w, const float* cc_s, int nb_steps,
                            long long B, int d, int E,
                            float* z, float* log_jac, float* f_x, float* f_x0, void* stream);

/* The same block as one link of a UMNNMAFFlow stack (UMNNMAFFlow.py:109-123: `for net: z, lj = ...; log_jac += lj;
 * x = z[:, inv_idx]`), with the glue between blocks folded into the kernel's stores:
 *   reverse_z != 0   z is written with its dimensions reversed, z[b, d-1-i] -- the input of the next block;
 *   log_jac_in       nullable [B,d]: running sum over the previous blocks; log_jac = log_jac_in + this block's
 *                    (same index order as the reference's elementwise `+=`; may alias log_jac). */
int umnn_flow_stack_block_forward(const umnn_mlp* net, const float* x, const float* h, const float* scaling,
                                  const float* cc_w, const float* cc_s, int nb_steps,
                                  long long B, int d, int E, int reverse_z, const float* log_jac_in,
                                  float* z, float* log_jac, float* f_x, float* f_x0, void* stream);

/* umnn_cc_forward / umnn_flow_stack_block_forward / umnn_cc_backward with bf16 or fp32 activation storage (see umnn_io). */
int umnn_cc_forward_io(const umnn_mlp* net, const umnn_io* io, const void* x0, const void* x, const void* h,
                       const float* cc_w, const float* cc_s, int nb_steps,
                       long long B, int d, int E, int inv_f, void* F, void* f_x, void* f_x0, void* stream);
int umnn_flow_stack_block_forward_io(const umnn_mlp* net, const umnn_io* io, const void* x, const void* h,
                                     const float* scaling, const float* cc_w, const float* cc_s, int nb_steps,
                                     long long B, int d, int E, int reverse_z, const void* log_jac_in,
                                     void* z, void* log_jac, void* f_x, void* f_x0, void* stream);
/* umnn_cc_backward_io also takes inv_f (io may be null = fp32 storage): the backward of an integral of 1/f,
 * ParallelNeuralIntegral.apply(..., inv_f=True) -- d_theta and d_h differentiate 1/f (computeIntegrand, ParallelNeuralIntegral.py:70-72),
 * d_x / d_x0 keep f(x) g and -f(x0) g exactly as the reference's backward returns them (:117-123). */
int umnn_cc_backward_io(const umnn_mlp* net, const umnn_io* io, const void* x0, const void* x, const void* h,
                        const void* g, const void* g_fx,
                        const float* cc_w, const float* cc_s, int nb_steps, long long B, int d, int E, int inv_f,
                        void* dx0, void* dx, void* dh, float* dtheta,
                        void* workspace, long long workspace_bytes, void* stream);

/* One link of UMNNMAFFlow.compute_ll (UMNNMAFFlow.py:109-119) with the WHOLE log-likelihood arithmetic inside the
 * launch: besides z (reversed for the next block when reverse_z != 0) the kernel keeps the per-sample running sum
 *   ll[b]  = (first ? 0 : ll[b]) + sum_i log_jac[b,i]            (UMNNMAF.compute_log_jac, UMNNMAF.py:136-139)
 *   ll[b] += -1/2 * sum_i (log(2 pi) + z[b,i]^2)   when last != 0  (UMNNMAFFlow.py:116-118)
 * so a flow's compute_ll is exactly nb_flow x (conditioner + this launch): no [B,d] log_jac accumulation between
 * blocks, no elementwise epilogue kernels.  The row sum is taken by the last tile to deliver a piece of the row
 * (arrival counters + device-scope fences), in a fixed order: bit-reproducible, no floating-point atomics.
 *   log_jac_scratch  [B,d] write-then-read scratch of this launch (this block's log_jac; may be reused by the next)
 *   row_counters     [B] uint32, ZERO on entry; the kernel leaves them zero again
 *   z must not alias x. */
int umnn_flow_ll_block_forward(const umnn_mlp* net, const float* x, const float* h, const float* scaling,
                               const float* cc_w, const float* cc_s, int nb_steps,
                               long long B, int d, int E, int reverse_z, int first, int last,
                               float* z, float* log_jac_scratch, float* ll, unsigned* row_counters, void* stream);

/* Sampling direction, one flow dimension per call: replaces the inner loops of UMNNMAF.invert (UMNNMAF.py:195-231) for
 * dimension j -- `iters` rounds of the 10-candidate bracket search on [-50, 50], each candidate's image being
 * exp(scaling_j) * (h[b, 0*d+j] + int_0^cand f(t; h[b, :, j]) dt) -- for all B samples in ONE launch.
 *   h      [B, E*d]  conditioner output for the current x_inv (dimensions < j already final; MADE is autoregressive,
 *                    so the caller runs it once per dimension, exactly like the reference :198)
 *   z      [B, d]    targets (column j is read);   x_inv [B, d]: column j is WRITTEN (the round's best candidate)
 * UMNN_EUNSUPPORTED for nets with a single hidden layer or images beyond 160 KiB of LDS (callers keep their own loop). */
int umnn_flow_invert_dim(const umnn_mlp* net, const float* h, const float* z, const float* scaling,
                         const float* cc_w, const float* cc_s, int nb_steps,
                         long long B, int d, int E, int j, int iters, float* x_inv, void* stream);

/* Replaces integrate(..., compute_grad=True) + the Leibniz terms -- ParallelNeuralIntegral.py:66-94,
 * 110-123 (NeuralIntegral.py:47-58,69-75,90-99).  g is grad_output [B,d] (cotangent of F).
 *   g_fx    nullable [B,d]: cotangent of the f_x output of umnn_cc_forward.  The reference gets this
 *           term from ordinary autograd through `self.net.forward(x)` (UMNNMAF.py:138,143,148); here it
 *           is one more VJP at quadrature node 0, including d f/d x.
 *   dtheta  [n_params] in integrand.parameters() order (W0,b0,W1,b1,...); OVERWRITTEN
 *   dh      [B,E*d]; dx, dx0 [B,d]  (dx = f(x;h)*g [+ g_fx * df/dx], dx0 = -f(x0;h)*g); each nullable.
 *   workspace: device scratch of at least umnn_cc_backward_workspace_bytes() bytes. */
int umnn_cc_backward(const umnn_mlp* net, const float* x0, const float* x, const float* h,
                     const float* g, const float* g_fx,
                     const float* cc_w, const float* cc_s, int nb_steps,
                     long long B, int d, int E,
                     float* dx0, float* dx, float* dh, float* dtheta,
                     void* workspace, long long workspace_bytes, void* stream);
long long umnn_cc_backward_workspace_bytes(const umnn_mlp* net, long long B, int d, int E);

/* Training forward / backward pair for nets of the three-stage backward family (first hidden layer of 5..8 sixteen-feature tiles
 * over 2..4 narrower ones: MNISTExperiment's 31-100-50-50-50-50-1, /root/reference MNISTExperiment.py:238).  That backward recomputes
 * the pre-activations z_2 of hidden layer 2 at every node in its stage A and hands them to stage B through HBM; the forward has just
 * computed the same numbers.  umnn_flow_stack_block_forward_save = umnn_flow_stack_block_forward that also leaves them in z2_save
 * ([tile][node][register][lane] fp32, umnn_cc_forward_z2_floats() floats: 0.83 GB for 100 x 784 integrals at n = 50);
 * umnn_cc_backward_saved = umnn_cc_backward (lower limit 0) that reads them and runs stage A for the tangent element of the g_fx term
 * only.  umnn_cc_forward_z2_floats returns 0 when the pair does not apply (other nets; fwd_precision fp32 / bf16x6; bwd_precision fp32):
 * callers then use the plain entry points.  Trade: memory held from forward to backward for ~0.26 ms per MNIST-shaped block. */
long long umnn_cc_forward_z2_floats(const umnn_mlp* net, long long B, int d, int E, int nb_steps);
int umnn_flow_stack_block_forward_save(const umnn_mlp* net, const float* x, const float* h, const float* scaling,
                                       const float* cc_w, const float* cc_s, int nb_steps,
                                       long long B, int d, int E, int reverse_z, const float* log_jac_in,
                                       float* z, float* log_jac, float* f_x, float* f_x0,
                                       float* z2_save, long long z2_floats, void* stream);
int umnn_cc_backward_saved(const umnn_mlp* net, const float* x, const float* h, const float* g, const float* g_fx,
                           const float* cc_w, const float* cc_s, int nb_steps, long long B, int d, int E,
                           float* dx, float* dh, float* dtheta, const float* z2_saved, long long z2_floats,
                           void* workspace, long long workspace_bytes, void* stream);
/* The pair with bf16 or fp32 activation storage (umnn_io; io may be null = fp32): configuration C4's bf16 embedding on the training path. */
int umnn_flow_stack_block_forward_save_io(const umnn_mlp* net, const umnn_io* io, const void* x, const void* h, const float* scaling,
                                          const float* cc_w, const float* cc_s, int nb_steps,
                                          long long B, int d, int E, int reverse_z, const void* log_jac_in,
                                          void* z, void* log_jac, void* f_x, void* f_x0,
                                          float* z2_save, long long z2_floats, void* stream);
int umnn_cc_backward_saved_io(const umnn_mlp* net, const umnn_io* io, const void* x, const void* h, const void* g, const void* g_fx,
                              const float* cc_w, const float* cc_s, int nb_steps, long long B, int d, int E,
                              void* dx, void* dh, float* dtheta, const float* z2_saved, long long z2_floats,
                              void* workspace, long long workspace_bytes, void* stream);

/* Elementwise glue of a block's TRAINING path (umnn_amd/csrc/cc_flow_glue.hip), one launch each.
 * umnn_flow_block_cotangents: backward of the block epilogue (UMNNMAF.py:80-83,134,138-139)
 *     z[b, rev(i)] = exp(s_i) (F[b,i] + h_0[b,i]),  log_jac[b,i] = log(f_x[b,i] + 1e-10) + s_i      (rev(i) = d-1-i if reverse_z)
 *   from the cotangents g_z, g_log_jac [B,d] (either nullable = zero) to gF = exp(s_i) g_z[b, rev(i)] (also the cotangent of h_0) and
 *   g_fx = g_log_jac / (f_x + 1e-10) (g_fx nullable: not wanted) -- the g / g_fx inputs of umnn_cc_backward.
 * umnn_flow_ll_forward / _backward: UMNNMAFFlow.compute_ll's reduction (UMNNMAFFlow.py:109-119)
 *     ll[b] = sum_i log_jac[b,i] - 1/2 sum_i (log 2 pi + z[b,i]^2);   g_log_jac[b,i] = g_ll[b],  g_z[b,i] = -z[b,i] g_ll[b]. */
int umnn_flow_block_cotangents(const float* g_z, const float* g_log_jac, const float* f_x, const float* scaling,
                               long long B, int d, int reverse_z, float* gF, float* g_fx, void* stream);
int umnn_flow_ll_forward(const float* z, const float* log_jac, long long B, int d, float* ll, void* stream);
int umnn_flow_ll_backward(const float* z, const float* g_ll, long long B, int d, float* g_z, float* g_log_jac, void* stream);

/* Replaces compute_cc_weights -- ParallelNeuralIntegral.py:14-34: writes nb_steps+1 fp32
 * weights and nodes into HOST buffers (float64 arithmetic, cast at the end). */
int umnn_cc_tables_host(int nb_steps, float* w_host, float* s_host);

/* Algorithmic FLOPs of one forward integral, SURVEY 8(d):
 *   2*[(n+1)*(H1 + sum H_l*H_{l+1} + H_L) + E*H1]. */
double umnn_cc_forward_flops_per_integral(const umnn_mlp* net, int nb_steps);

/* Introspection used by tests, the smoke test and the bench. */
const char* umnn_last_error(void);
int umnn_version(void);
long long umnn_launch_count(void);          /* quadrature kernels (forward / backward / finishing) launched by this process so far */
long long umnn_made_launch_count(void);     /* launches of the fused conditioner kernel (umnn_made_mlp_forward) so far */
const char* umnn_last_made_kernel_name(void);
const char* umnn_last_kernel_name(void);    /* variant picked by the last forward/backward */
/* Times `reps` back-to-back umnn_cc_forward launches with hipEvents recorded on `stream`
 * (the stream the kernels run on) and returns the average milliseconds per launch in *ms. */
int umnn_cc_forward_timed(const umnn_mlp* net, const float* x0, const float* x, const float* h,
                          const float* cc_w, const float* cc_s, int nb_steps,
                          long long B, int d, int E, float* F, float* f_x, float* f_x0,
                          int reps, float* ms, void* stream);

/* Arithmetic of the hidden-layer GEMMs in the FORWARD kernels -- umnn_cc_forward, the umnn_flow_*_forward entry points, their _io
 * forms and umnn_flow_invert_dim (process-wide; default UMNN_PRECISION_F16X3, or the environment variable UMNN_FWD_PRECISION =
 * fp32 | bf16x3 | bf16x6 | f16x3 at first use).  Every mode meets the 1e-4 tolerance of the path with margin (measured max
 * relative error of F on the golden set: fp32 ~4e-7, bf16x6 ~4e-7, f16x3 ~5e-7, bf16x3 ~6e-6; layer 1, the hoisted first-layer
 * term, the output dot product, ELU and the quadrature sum are fp32 in every mode).
 *   F16X3   (default since round 5)  operands split in two fp16 pieces (11 bits each), 3 cross terms on v_mfma_f32_16x16x32_f16,
 *           fp32 accumulation: fp32-level accuracy at the cost of BF16X3.  fp16's exponent range is handled by an overflow
 *           protocol (umnn_amd/csrc/cc_forward_bf16.hip): a tile group (16 or 32 integrals) whose quadrature sum is not finite --
 *           where an overflowed piece always ends -- writes only a NaN marker into its slots of F (z for the flow entry points,
 *           x_inv[:, j] for umnn_flow_invert_dim), and the BF16X3 build of the same kernel, queued behind every launch on the
 *           same stream, recomputes exactly the marked groups (one scalar load per workgroup when nothing overflowed).  Those
 *           integrals therefore come back at BF16X3 accuracy -- never NaN unless an input was, never a wrong finite value.  A
 *           launch whose marker output aliases x, x0 or h runs BF16X3 outright.  umnn_launch_count() counts the pair as one.
 *   FP32    v_mfma_f32_16x16x4_f32: exact fp32 products (an fmaf chain), fp32-vector-rate matrix path -- the reference's arithmetic
 *   BF16X3  operands split in two bf16 pieces, 3 cross terms on v_mfma_f32_16x16x32_bf16, fp32 accumulation (the default until round 4)
 *   BF16X6  three bf16 pieces, 6 cross terms: fp32-level accuracy at twice the matrix work
 * umnn_flow_invert_dim: F16X3 and BF16X3 run the in-kernel bracket search for every net the forward covers; FP32 / BF16X6 run it on
 * three bf16 pieces for nets of at most four 16-feature tiles per layer and on two fp16 pieces (the F16X3 search: fp32-level
 * products, overflowing samples redone on bf16 pieces) for wider ones, whose three-piece form does not fit the register file. */
#define UMNN_PRECISION_FP32 0
#define UMNN_PRECISION_BF16X3 1
#define UMNN_PRECISION_BF16X6 2
#define UMNN_PRECISION_F16X3 3
int umnn_set_forward_precision(int mode);
int umnn_get_forward_precision(void);
/* Arithmetic of the BACKWARD kernels (umnn_cc_backward, _io): UMNN_PRECISION_BF16X3 (default) or UMNN_PRECISION_FP32 (env
 * UMNN_BWD_PRECISION = bf16x3 | fp32).  Under both, d_x0 / d_x Leibniz terms, the output layer, dc and every reduction are fp32.
 *   BF16X3  the matrix-core kernels.  The forward recompute inside them is always fp32-LEVEL (the sign of every hidden
 *           pre-activation decides a LeakyReLU slope): six cross terms on three bf16 pieces, or three cross terms on two fp16
 *           pieces; the delta chain and the dW products carry three cross terms.  Which kernel runs:
 *             - four hidden layers of 32..63 units, ELU+1 output, un-split node range, >= 4 tiles per workgroup:
 *               the eight-wave workgroup pipeline -- on fp16 pieces (cc_bwd_ws16_kernel.h, kernel names cc_bwd_f16<...,WS>) for
 *               launches of >= 2^21 node evaluations (option bwd_ws16 = 1; 2 = whenever eligible; 0 = never), on bf16 pieces
 *               (cc_bwd_ws_kernel.h, cc_bwd_bf16<...,WS>) otherwise and for 1/f launches and sigmoid outputs.  The fp16 kernel
 *               carries cotangents scaled by a per-launch power of two (cc_bwd_cotmax_kernel), detects an overflowed piece in
 *               its output-layer and dc sums, raises a flag word in the workspace, and the bf16 pipeline queued behind it
 *               rewrites every output when -- and only when -- the flag is set.  Measured against a float64 run of the reference
 *               algorithm at the benchmarked size (8192 x 63 x 101 nodes): d_theta 5.8e-5 of its largest entry (exact-fp32
 *               kernels 3.1e-5, a float32 ATen run of the reference's own algorithm 2.7e-5, the bf16 pipeline 2.4e-4);
 *             - two or three hidden layers of 32..63 units, or small batches: the software-pipelined one-pass loop
 *               (cc_bwd_swp_kernel.h; option bwd_swp = 0: the round-2 loop);
 *             - a first hidden layer of 5..8 tiles over 2..4 narrower ones (31-100-50-50-50-50-1): three stages through HBM
 *               (cc_backward_front.hip) -- stages A and B on fp16 pieces for single-chunk calls of >= 2^21 node evaluations with
 *               their bf16 builds queued as the overflow fallback, on bf16 pieces otherwise;
 *             - every other shape: the fp32 kernels below.
 *   FP32    exact fp32 MFMA (cc_backward.hip) for every one-pass shape; the three-stage family runs its build with three bf16
 *           pieces / six cross terms in EVERY product (cc_backward_front_p3.hip, kernel names cc_bwd_bf16x6<...>): fp32-level,
 *           not fp32 instructions.
 * Narrower layers of a 32..63-wide net are zero-padded to four 16-feature tiles by the staged weight images. */
/* Which kernel family umnn_cc_backward would run for this net: 1 shape-exact kernels (one pass; or, for a first hidden layer
 * of 5..8 sixteen-feature tiles over 2..4 narrower ones such as MNISTExperiment's 100-50-50-50-50, the three-stage kernels
 * of cc_backward_front.hip -- under UMNN_PRECISION_FP32 their build with six bf16 cross terms in every product, fp32-level; or, for unequal hidden widths of 64..127, the shape-exact fp32 kernels of
 * the 5- / 7- / 8-tile family that holds the widest layer, the narrower layers zero-padded virtually), 0 generic kernels with
 * at most four tiles per layer, -1 generic kernels with more tiles (deep wide nets whose zero-padded weight images exceed the
 * LDS: they spill registers and are ~100x slower -- the shipped host code sends such nets to the materialised ATen chain on the
 * GPU), < -1 error.  The staged family takes its HBM scratch from the
 * workspace (umnn_cc_backward_workspace_bytes includes up to 2 GiB for it). */
int umnn_cc_backward_kind(const umnn_mlp* net, int E);
int umnn_set_backward_precision(int mode);
int umnn_get_backward_precision(void);

/* Launch options.  The UMNN_* environment variables are read once per process -- at first use or on
 * umnn_reload_env() -- and live in atomics afterwards, so the launch path never calls getenv.  Names:
 * "fwd_precision", "bwd_precision" (UMNN_PRECISION_*), "fwd_p" (1|2), "fwd_ns" (1|2|4), "fwd_tail" (0|1), "fwd_pipe"
 * (0 plain loop | 1 pipelined 16x16x32 loop, default | 2 the 32x32x16 formulation of the 48..51-wide shapes),
 * "fwd_pad", "fwd_pad_min", "bwd_ns" (1..32), "bwd_swp" (1: software-pipelined one-pass bf16 backward, default; 0: the
 * round-2 node loop), "bwd_ws" (1, default: four-hidden-layer nets at large batch take the weight-stationary workgroup
 * pipeline, cc_bwd_ws_kernel.h; 0: they stay on the one-pass loop), "bwd_ws16" (that pipeline on fp16 pieces,
 * cc_bwd_ws16_kernel.h -- three-term recompute, cotangents scaled by a per-launch power of two, overflow flag + queued bf16
 * fallback: 1, default: launches of >= 2^21 node evaluations; 2: whenever the pipeline is eligible; 0: never), "front_bwd2" (1,
 * default: the last stage of the three-stage backward runs two waves per tile of integrals, each on half of the wide first hidden
 * layer's feature tiles -- on fp16 pieces behind the fp16 middle stage, its bf16 build queued as the overflow fallback; 2: two waves,
 * bf16 pieces always; 0: one wave per tile);
 * -1 = automatic for the tuning knobs.  "fwd_precision" defaults to UMNN_PRECISION_F16X3 (see above). */
int umnn_set_option(const char* name, int value);
int umnn_get_option(const char* name, int* value);
int umnn_reload_env(void);

/* Per-launch timing: while enabled, every forward/backward launch is bracketed by hipEvents recorded on its own
 * launch stream.  umnn_profile_read synchronises on them and returns the summed kernel milliseconds, the number of
 * launches and the summed algorithmic FLOPs (forward launches only carry FLOPs).  enable(0/1) clears the records. */
int umnn_profile_enable(int on);
int umnn_profile_read(double* total_ms, long long* launches, double* total_flops);
/* The same per class of launch: forward quadrature kernels (FLOPs = the algorithmic forward count), the main backward
 * kernels (FLOPs = 3 x the forward count of the same integrals: 2 x for the two gradient GEMMs per forward GEMM + 1 x for the
 * forward recompute, SURVEY 8d "Backward ~ 2x forward MACs + recompute"), and the finishing kernels of the backward
 * (dc sum, d_h, dW1, d_theta reduction; bracketed as one record, no FLOPs). */
#define UMNN_PROF_FORWARD 0
#define UMNN_PROF_BACKWARD 1
#define UMNN_PROF_FINISH 2
int umnn_profile_read_tag(int tag, double* total_ms, long long* launches, double* total_flops);
/* Variant name of the last launch of that class by ANY thread (umnn_last_kernel_name is per calling thread, and the
 * backward runs on autograd worker threads). */
const char* umnn_last_kernel_name_of(int tag);

/* MADE conditioner operand builder (reference models/UMNN/made.py:16-27: MaskedLinear chains around ReLU).  Reads the
 * previous layer's raw fp32 output x [rows, cols] and writes, per row, the bf16 operand
 *   [ hi(a) | lo(a) | hi(a) | 1 | 1 | 0 ... ]  (ld_out >= 3*cols+2 bf16 per row),  a = relu ? max(x,0) : x,
 * hi/lo = round-to-nearest bf16 pieces of a.  Multiplied (one bf16 GEMM, fp32 accumulate) by the host-packed weights
 * [Wh | Wh | Wl | bh | bl | 0...] it gives W a + b to ~3e-6 of the output range. */
int umnn_made_split3(const float* x, long long rows, int cols, int relu, void* out_bf16, int ld_out, void* stream);

/* The whole MADE conditioner of one flow block as ONE launch (reference models/UMNN/made.py:16-27 MaskedLinear, :113-119
 * MADE.forward, :165-168 ConditionnalMADE.forward; called once per block, UMNNMAF.py:79):
 *     h = W_L relu( ... relu(W_1 x + b_1) ... ) + b_L,   x [B, widths[0]] fp32  ->  h [B, widths[n_layers]] fp32 or bf16.
 * W[l]: the MASKED weight of layer l as bf16 MFMA fragments [tile t][K-step s][piece (hi, lo)][lane 64][8 bf16], lane
 * (g = lane>>4, rho = lane&15) slot j holding piece(W[16 t + rho][32 s + 16 (j>>2) + 4 g + (j&3)]) (zero outside the matrix)
 * -- the K order in which the kernel's accumulators become the next layer's operand; b[l]: fp32 bias.  Input and hidden
 * widths up to 512 (UMNN_EUNSUPPORTED beyond: the caller keeps the library GEMMs); products as xh Wh + xl Wh + xh Wl with
 * fp32 accumulation (the arithmetic of umnn_made_split3's GEMMs), bias added in fp32. */
#define UMNN_MADE_MAX_LAYERS 8
typedef struct umnn_made_net {
    int n_layers;
    int widths[UMNN_MADE_MAX_LAYERS + 1];
    const void* W[UMNN_MADE_MAX_LAYERS];
    const float* b[UMNN_MADE_MAX_LAYERS];
} umnn_made_net;
int umnn_made_mlp_forward(const umnn_made_net* net, const float* x, long long B, void* h_out, int out_bf16, void* stream);
/* The same kernel with the output format spelt out: out_mode 0 fp32 h, 1 bf16 h, 2 = the masked MLP's HIDDEN stack only -- the last
 * layer given here is followed by ReLU and written as umnn_made_split3's operand [hi | lo | hi | 1 | 1 | 0...] (out_ld bf16 per row)
 * for a wide output layer that stays a library GEMM (BSDS300's 1890, the VAE flow's 1920 columns). */
int umnn_made_mlp_forward_ex(const umnn_made_net* net, const float* x, long long B, void* out, int out_mode, int out_ld, void* stream);
/* ONE masked linear layer as its own launch (made.py:16-27 MaskedLinear.forward = F.linear(input, mask * weight, bias)), for
 * conditioners whose OUTPUT layer is wide (the VAE flow's 1920, BSDS300's 1890 columns), where one workgroup per row group cannot
 * fill the chip: the grid is (row groups of 16*row_tiles rows) x (feature_groups groups of output tiles); the bf16 split of the
 * input happens in the operand load (relu_in != 0: the input is the previous layer's pre-activation and ReLU is applied on load, so
 * a hidden layer needs no separate activation / split launch); x2 != NULL: the input is [x | x2] with x [B, K1] and x2 [B, K - K1]
 * (ConditionnalMADE's cat((context, x), 1), made.py:167, without the copy); out = in' W^T + b as fp32 (out_bf16 == 0) or bf16.
 * W_frag / bias / arithmetic as in umnn_made_net; K <= 512.  row_tiles 0 (auto) | 1 | 2 | 4; feature_groups 0 (auto) or a count. */
int umnn_made_linear_forward(const void* W_frag, const float* bias, int K, int N, const float* x, const float* x2, int K1,
                             long long B, int relu_in, void* out, int out_bf16, int row_tiles, int feature_groups, void* stream);

/* Training chain of the conditioner (csrc/made_train.hip; replaces, per masked linear, the ReluBackward / bias-reduction / mask
 * product nodes PyTorch's autograd puts around F.linear in models/UMNN/made.py:16-27,113-119): g [B, N] fp32 is the gradient w.r.t.
 * the layer's ReLU output `relu_out` (NULL for the last layer: no activation behind it).  In place g <- g . [relu_out > 0], and
 * gb[c] <- sum_r g[r][c] by a two-stage reduction with a fixed order (bit-reproducible; no atomics).  `partial`: row_blocks x N floats
 * of scratch, row_blocks from umnn_made_relu_bwd_bias_row_blocks(B, N). */
int umnn_made_relu_bwd_bias_row_blocks(long long B, int N);
int umnn_made_relu_bwd_bias(float* g, const float* relu_out, long long B, int N, float* partial, int row_blocks, float* gb, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UMNN_CC_H */
