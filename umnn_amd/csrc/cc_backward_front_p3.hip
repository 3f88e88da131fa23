// The three-stage backward of cc_backward_front.hip with THREE bf16 pieces (six cross terms) in the delta chain and the dW products
// as well as in the recompute: fp32-level arithmetic on the bf16 matrix pipe.  This is what bwd_precision = fp32 runs for nets with a
// wide first hidden layer (MNISTExperiment's 31-100-50^4-1, /root/reference MNISTExperiment.py:18,238) -- the exact-fp32 one-pass
// kernels do not hold that shape (cc_backward_front.hip, header), and until round 4 such calls left the library for an ATen chain.
// Same kernels, same launcher, compiled into namespace bwd_p3; entry point umnn_launch_backward_front_p3.
#define UMNN_BWD_NPB 3
#include "cc_backward_front.hip"
