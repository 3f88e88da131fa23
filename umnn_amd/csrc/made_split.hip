// Operand builder for the MADE conditioner's GEMMs on the bf16 matrix cores (adjacent row a13 / "next" row f1 of the
// scope table; reference models/UMNN/made.py:16-27,113-119: h = MaskedLinear(ReLU(...))).
//
// The conditioner's three masked linears are plain library GEMMs (hipBLASLt through torch.mm), but at fp32 they run at
// the fp32 MFMA rate (105 TFLOP/s measured for the 8192x512x1890 output layer).  With x = xh + xl and W = Wh + Wl in
// bf16 pieces,   x W^T ~= [xh | xl | xh] [Wh | Wh | Wl]^T   is ONE bf16 GEMM with 3x the K and fp32 accumulation
// (max error 3e-6 of the output range, measured) that runs 2.8x faster.  This kernel builds the left operand in one
// pass over the previous layer's raw fp32 output:   out[r] = [hi(a) | lo(a) | hi(a) | 1 | 1 | 0...],  a = act(x[r]),
// act = ReLU or identity; the two constant columns meet the bias rows [bh | bl] the host appends to the weights.
// HBM-bound: 4 B read, 6 B written per element.
#include <hip/hip_runtime.h>
#include "cc_bf16.h"
#include "cc_host.h"
#include "../../include/umnn_cc.h"

__global__ __launch_bounds__(256) void made_split3_kernel(const float* __restrict__ x, long long rows, int cols, int relu,
                                                          unsigned short* __restrict__ out, int ld) {
    const long long npair = (long long)((cols + 1) / 2);
    const long long total = rows * npair;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long r = idx / npair;
        const int c = (int)(idx - r * npair) * 2;
        const float* xr = x + r * cols;
        float a0 = xr[c], a1 = c + 1 < cols ? xr[c + 1] : 0.f;
        if (relu) { a0 = fmaxf(a0, 0.f); a1 = fmaxf(a1, 0.f); }
        unsigned q[2];
        split_pair<2>(a0, a1, q);
        unsigned short* o = out + r * ld;
        const unsigned short h0 = (unsigned short)(q[0] & 0xffffu), h1 = (unsigned short)(q[0] >> 16);
        const unsigned short l0 = (unsigned short)(q[1] & 0xffffu), l1 = (unsigned short)(q[1] >> 16);
        o[c] = h0; o[cols + c] = l0; o[2 * cols + c] = h0;
        if (c + 1 < cols) { o[c + 1] = h1; o[cols + c + 1] = l1; o[2 * cols + c + 1] = h1; }
        if (c == 0) {
            o[3 * cols] = 0x3f80; o[3 * cols + 1] = 0x3f80;                  // bf16(1.0) twice: bias pieces
            for (int k = 3 * cols + 2; k < ld; ++k) o[k] = 0;
        }
    }
}

extern "C" int umnn_made_split3(const float* x, long long rows, int cols, int relu, void* out_bf16, int ld_out,
                                void* stream) {
    if (rows < 0 || cols < 1 || ld_out < 3 * cols + 2) return umnn_fail(UMNN_EINVAL, "made_split3: bad shape");
    if (rows == 0) return 0;
    if (!x || !out_bf16) return umnn_fail(UMNN_EINVAL, "made_split3: null pointer");
    const long long total = rows * (long long)((cols + 1) / 2);
    long long blocks = (total + 255) / 256;
    const long long cap = (long long)umnn_num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(made_split3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, rows, cols, relu,
                       (unsigned short*)out_bf16, ld_out);
    return umnn_check(hipGetLastError(), "made_split3 launch");
}
