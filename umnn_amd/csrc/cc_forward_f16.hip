// Forward quadrature on fp16 PIECES: cc_forward_bf16.hip (variant table + launcher) and cc_fwd_bf16_kernel.h (the kernels) compiled
// with the 16-bit piece type switched to fp16 -- v_mfma_f32_16x16x32_f16, v_cvt_pk_f16_f32.  Two 11-bit pieces and three cross terms
// give fp32-level accuracy (~4e-7 on F: the accuracy of the three-piece bf16x6 mode at the cost of the two-piece bf16x3 mode).
// Selected by fwd_precision = UMNN_PRECISION_F16X3 (umnn_amd.set_precision("f16x3")); exports umnn_launch_forward_f16.
// Range: an fp16 piece overflows at 65520; a tile group in which that happens is detected in its quadrature sum, writes only a NaN
// marker, and is recomputed by the bf16 build queued behind every launch of this build (overflow protocol: cc_forward_bf16.hip) --
// which is what made this the library DEFAULT in round 5.
#define UMNN_FWD_PIECE_F16 1
#include "cc_forward_bf16.hip"
