// Host-side helpers shared by the translation units of libumnn_cc.so.
#pragma once
#include <cstdio>
#include <cstdlib>

#include "cc_common.h"

int umnn_fail(int code, const char* msg);                 // records msg (thread-local), returns code
int umnn_check(hipError_t e, const char* what);           // 0 or records + returns the hipError_t
// Validates `net`, fills the device-side descriptor (tile / K-step counts, LDS offsets) and reports
// tmax = max tiles over hidden layers and ksu = the common K-step count if all hidden layers agree (else 0).
int umnn_prepare_mlp(const umnn_mlp* net, int E, MlpDev* out, int* tmax, int* ksu);
int umnn_num_cus();
int umnn_allow_lds(const void* fn, size_t bytes);         // hipFuncSetAttribute(MaxDynamicSharedMemorySize)
void umnn_note_launch(const char* kernel_name);
long long umnn_param_count(const umnn_mlp* net);

// Optional per-launch timing (umnn_profile_enable): hipEvents recorded on the launch stream around each kernel.
void umnn_prof_begin(hipStream_t stream);
void umnn_prof_end(hipStream_t stream, double flops);
