// Elementwise glue of a UMNN-MAF block's TRAINING path, one launch each instead of the dozen ATen launches (and autograd nodes)
// the same arithmetic costs when composed from torch ops -- HBM-bound [B, d] passes, a few microseconds each; at the launch-bound
// training shapes (POWER, VAE prior flow, MNISTExperiment) they were ~70 of a step's ~300 launches.
//   umnn_flow_block_cotangents   backward of the block epilogue z = e^s (F + h_0) [reversed], log_jac = log(f_x + 1e-10) + s
//                                (UMNNMAF.py:80-83,134,138-139): cotangents of F and f_x from those of z and log_jac
//   umnn_flow_ll_forward         ll = sum_i log_jac - 1/2 sum_i (log 2 pi + z_i^2)                  (UMNNMAFFlow.py:109-119)
//   umnn_flow_ll_backward        its backward: g_log_jac = g_ll, g_z = -z g_ll
#include "cc_host.h"

__global__ __launch_bounds__(256) void flow_block_cotangents_kernel(const float* __restrict__ g_z, const float* __restrict__ g_lj,
                                                                    const float* __restrict__ f_x, const float* __restrict__ scaling,
                                                                    long long NI, int d, int reverse_z, float* __restrict__ gF,
                                                                    float* __restrict__ g_fx) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= NI) return;
    const long long b = q / d;
    const int i = (int)(q - b * d);
    // z[b, reverse ? d-1-i : i] = e^{s_i} (F[b,i] + h_0[b,i])  =>  dL/dF[b,i] = e^{s_i} g_z[b, reverse ? d-1-i : i]
    const float gz = g_z ? g_z[reverse_z ? b * d + (d - 1 - i) : q] : 0.f;
    gF[q] = __expf(scaling[i]) * gz;
    if (g_fx) g_fx[q] = g_lj ? g_lj[q] / (f_x[q] + 1e-10f) : 0.f;
}

extern "C" int umnn_flow_block_cotangents(const float* g_z, const float* g_log_jac, const float* f_x, const float* scaling,
                                          long long B, int d, int reverse_z, float* gF, float* g_fx, void* stream) {
    if (B < 0 || d < 1) return umnn_fail(UMNN_EINVAL, "flow cotangents: B >= 0, d >= 1");
    if (B == 0) return 0;
    if (!scaling || !gF || (g_fx && g_log_jac && !f_x)) return umnn_fail(UMNN_EINVAL, "flow cotangents: null pointer");
    const long long NI = B * (long long)d;
    hipLaunchKernelGGL(flow_block_cotangents_kernel, dim3((unsigned)((NI + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       g_z, g_log_jac, f_x, scaling, NI, d, reverse_z, gF, g_fx);
    return umnn_check(hipGetLastError(), "flow_block_cotangents launch");
}

// one wave per row: lane-strided loads, fixed butterfly (deterministic)
__global__ __launch_bounds__(256) void flow_ll_forward_kernel(const float* __restrict__ z, const float* __restrict__ lj, long long B, int d,
                                                              float* __restrict__ ll) {
    const long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int e = lane; e < d; e += 64) {
        const float zz = z[b * d + e];
        s += lj[b * d + e] - 0.5f * (1.8378770664093453f + zz * zz);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) ll[b] = s;
}
__global__ __launch_bounds__(256) void flow_ll_backward_kernel(const float* __restrict__ z, const float* __restrict__ g_ll, long long NI, int d,
                                                               float* __restrict__ g_z, float* __restrict__ g_lj) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= NI) return;
    const float g = g_ll[q / d];
    if (g_z) g_z[q] = -z[q] * g;
    if (g_lj) g_lj[q] = g;
}

extern "C" int umnn_flow_ll_forward(const float* z, const float* log_jac, long long B, int d, float* ll, void* stream) {
    if (B < 0 || d < 1) return umnn_fail(UMNN_EINVAL, "flow ll: B >= 0, d >= 1");
    if (B == 0) return 0;
    if (!z || !log_jac || !ll) return umnn_fail(UMNN_EINVAL, "flow ll: null pointer");
    hipLaunchKernelGGL(flow_ll_forward_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, z, log_jac, B, d, ll);
    return umnn_check(hipGetLastError(), "flow_ll_forward launch");
}
extern "C" int umnn_flow_ll_backward(const float* z, const float* g_ll, long long B, int d, float* g_z, float* g_log_jac, void* stream) {
    if (B < 0 || d < 1) return umnn_fail(UMNN_EINVAL, "flow ll backward: B >= 0, d >= 1");
    if (B == 0) return 0;
    if (!z || !g_ll) return umnn_fail(UMNN_EINVAL, "flow ll backward: null pointer");
    const long long NI = B * (long long)d;
    hipLaunchKernelGGL(flow_ll_backward_kernel, dim3((unsigned)((NI + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, g_ll, NI, d, g_z,
                       g_log_jac);
    return umnn_check(hipGetLastError(), "flow_ll_backward launch");
}
