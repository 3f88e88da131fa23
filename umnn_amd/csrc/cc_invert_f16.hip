// In-kernel sampling on fp16 PIECES: cc_invert.hip (variant tables + launcher) and cc_fwd_bf16_kernel.h (the kernels) compiled with
// the 16-bit piece type switched to fp16.  Exports umnn_invert_impl_f16, which umnn_flow_invert_dim (the bf16 build of that file)
// calls under fwd_precision = f16x3, the library default; every launch is followed by its queued bf16x3 fallback (file header there).
#define UMNN_FWD_PIECE_F16 1
#include "cc_invert.hip"
