// Host-side helpers shared by the translation units of libumnn_cc.so.
#pragma once
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "cc_common.h"

int umnn_fail(int code, const char* msg);                 // records msg (thread-local), returns code
int umnn_check(hipError_t e, const char* what);           // 0 or records + returns the hipError_t
// Validates `net`, fills the device-side descriptor (tile / K-step counts, LDS offsets) and reports
// tmax = max tiles over hidden layers and ksu = the common K-step count if all hidden layers agree (else 0).
int umnn_prepare_mlp(const umnn_mlp* net, int E, MlpDev* out, int* tmax, int* ksu);
int umnn_check_io(const umnn_io* io);                     // 0, or UMNN_EINVAL for an unknown dtype code
int umnn_num_cus();                                      // CU count of the CURRENT device (cached per device)

// Process-wide launch options.  The UMNN_* environment variables are read ONCE (first use, or umnn_reload_env());
// after that every launch reads plain atomics -- no getenv on the launch path.  -1 = "auto" for the tuning knobs.
struct UmnnOptions {
    std::atomic<int> fwd_precision{UMNN_PRECISION_F16X3};    // UMNN_FWD_PRECISION = fp32 | bf16x3 | bf16x6 | f16x3 (default)
    std::atomic<int> bwd_precision{UMNN_PRECISION_BF16X3};   // UMNN_BWD_PRECISION = fp32 | bf16x3
    std::atomic<int> fwd_p{-1};          // UMNN_FWD_P: point tiles per wave (1|2)
    std::atomic<int> fwd_ns{-1};         // UMNN_FWD_NS: node-range split (1|2|4)
    std::atomic<int> fwd_tail{-1};       // UMNN_FWD_TAIL: VALU-tail fp32 variant (0|1)
    std::atomic<int> fwd_pipe{1};        // UMNN_FWD_PIPE: software-pipelined two-tile loop
    std::atomic<int> fwd_pad{1};         // UMNN_FWD_PAD: zero-pad mixed 3..4-tile nets to the shape-exact kernels
    std::atomic<int> fwd_pad_min{1};     // UMNN_FWD_PAD_MIN
    std::atomic<int> bwd_ns{-1};         // UMNN_BWD_NS: node-range split of the backward (1..32)
    std::atomic<int> bwd_ws{1};          // UMNN_BWD_WS: weight-stationary workgroup pipeline for four-hidden-layer nets at large batch (cc_bwd_ws_kernel.h)
    std::atomic<int> bwd_ws16{1};        // UMNN_BWD_WS16: that pipeline on fp16 pieces (cc_bwd_ws16_kernel.h: three-term recompute): 1 = for launches of >= 2^21 node evaluations, 2 = whenever eligible, 0 = never (the bf16 pipeline)
    std::atomic<int> front_bwd2{1};      // UMNN_FRONT_BWD2: stage C of the three-stage backward with two waves per tile of integrals (cc_backward_front.hip; 1: on fp16 pieces behind the fp16 middle stage, 2: bf16 pieces always); 0 = one
    std::atomic<int> bwd_swp{1};         // UMNN_BWD_SWP: software-pipelined one-pass bf16 backward (cc_bwd_swp_kernel.h); 0 = round-2 loop
};
UmnnOptions& umnn_options();
int umnn_allow_lds(const void* fn, size_t bytes);         // hipFuncSetAttribute(MaxDynamicSharedMemorySize)
void umnn_note_launch(const char* kernel_name);
void umnn_note_made_launch(const char* kernel_name);      // conditioner kernels: separate counter / name
long long umnn_param_count(const umnn_mlp* net);

// Optional per-launch timing (umnn_profile_enable): hipEvents recorded on the launch stream around each kernel.
void umnn_prof_begin(hipStream_t stream);
void umnn_prof_end(hipStream_t stream, double flops, int tag = UMNN_PROF_FORWARD);
