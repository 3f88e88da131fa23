// Training path of the MADE conditioner (adjacent row a13 / "next" row f1 of the scope table; reference models/UMNN/made.py:16-27,
// 113-119: h = MaskedLinear(ReLU(MaskedLinear(...))) under autograd).  The GEMMs of the chain stay library GEMMs (hipBLASLt through
// torch); what PyTorch's autograd adds around every layer -- a ReLU-backward pass, a column reduction for the bias gradient (16 us each
// at the UCI shapes, a single-pass reduction whose order torch picks), and the mask product with its own backward node -- is one
// pass here:    g_y = g_a . [a > 0]  (in place; a = the layer's ReLU output)   and   d_b[c] = sum_r g_y[r][c]
// as a two-stage reduction with a FIXED order: stage 1 sums a block of rows per workgroup (each wave its rows in ascending order, the
// four waves combined in wave order), stage 2 adds the row-block partials in a fixed interleaved order.  No atomics: data-parallel replicas that see the same rows produce the same bits.
// HBM-bound: 8 B read (4 on the last layer, which has no ReLU behind it), 4 B written per element.
#include <hip/hip_runtime.h>
#include "cc_host.h"
#include "../../include/umnn_cc.h"

// stage 1.  Workgroup = 256 columns x a block of rows; its four waves take the rows of the block round-robin, a lane owns the four
// columns c0 + lane + 64 j (every access a coalesced 256-B row segment, eight independent loads in flight per row pair); the waves'
// sums meet in LDS and are added in wave order.
__global__ __launch_bounds__(256) void made_relu_bwd_bias_kernel(float* __restrict__ g, const float* __restrict__ a, long long B, int N,
                                                                 long long rows_per_block, float* __restrict__ partial) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 256 + lane;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > B) r1 = B;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if ((N & 3) == 0) {
        // rows are 16-byte aligned: a lane owns FOUR CONSECUTIVE columns (one 16-byte access per tensor and row, a wave a contiguous
        // 1 KB segment), four rows of the wave in flight at once
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int cv = blockIdx.x * 256 + 4 * lane;
        if (cv < N) {
            for (long long r = r0 + w; r < r1; r += 16) {
                f4 gv[4], av[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long rr = r + 4 * u;
                    const bool h = rr < r1;
                    gv[u] = h ? *reinterpret_cast<const f4*>(g + rr * N + cv) : f4{0.f, 0.f, 0.f, 0.f};
                    av[u] = (a && h) ? *reinterpret_cast<const f4*>(a + rr * N + cv) : f4{1.f, 1.f, 1.f, 1.f};
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long rr = r + 4 * u;
                    f4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = av[u][j] > 0.f ? gv[u][j] : 0.f; acc[j] += v[j]; }
                    if (a && rr < r1) *reinterpret_cast<f4*>(g + rr * N + cv) = v;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) red[w][4 * lane + j] = acc[j];
        __syncthreads();
        const int c = blockIdx.x * 256 + threadIdx.x;
        if (c < N) partial[(long long)blockIdx.y * N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        return;
    }
    bool in[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) in[j] = c0 + 64 * j < N;
    for (long long r = r0 + w; r < r1; r += 8) {
        const long long ra = r, rb = r + 4;
        const bool hb = rb < r1;
        float ga[4], gb2[4], aa[4], ab[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long ia = ra * N + c0 + 64 * j, ib = rb * N + c0 + 64 * j;
            ga[j] = in[j] ? g[ia] : 0.f;
            gb2[j] = (in[j] && hb) ? g[ib] : 0.f;
            aa[j] = (a && in[j]) ? a[ia] : 1.f;
            ab[j] = (a && in[j] && hb) ? a[ib] : 1.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long ia = ra * N + c0 + 64 * j, ib = rb * N + c0 + 64 * j;
            const float va = aa[j] > 0.f ? ga[j] : 0.f, vb = ab[j] > 0.f ? gb2[j] : 0.f;
            if (a && in[j]) { g[ia] = va; if (hb) g[ib] = vb; }
            acc[j] += va;            // (row order within a wave: ascending -- fixed)
            acc[j] += vb;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[w][lane + 64 * j] = acc[j];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < N) partial[(long long)blockIdx.y * N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// stage 2: workgroup = 64 columns; wave w adds the row-block partials j = w, w + 4, ... of its columns (two alternating chains: loads
// of both in flight), the four waves' sums meet in LDS and are combined in wave order -- every step in a fixed order
__global__ __launch_bounds__(256) void made_bias_finish_kernel(const float* __restrict__ partial, int N, int row_blocks, float* __restrict__ gb) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (c < N) {
        int j = w;
        for (; j + 4 < row_blocks; j += 8) {
            s0 += partial[(long long)j * N + c];
            s1 += partial[(long long)(j + 4) * N + c];
        }
        if (j < row_blocks) s0 += partial[(long long)j * N + c];
    }
    red[w][lane] = s0 + s1;
    __syncthreads();
    if (w == 0 && c < N) gb[c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// how many row blocks stage 1 uses for a [B, N] gradient (the caller sizes `partial` as row_blocks x N floats with it): enough
// workgroups to cover the chip a few times over, never more than one block per 8 rows
extern "C" int umnn_made_relu_bwd_bias_row_blocks(long long B, int N) {
    if (B < 1 || N < 1) return 1;
    // 64 rows per workgroup (16 per wave), fewer (>= 16) when that leaves fewer than four workgroups per CU -- and never more than 64
    // row blocks: stage 2 walks them one after the other (a dependent chain of L2 round trips per column)
    const long long colb = (N + 255) / 256;
    long long rows = 64;
    while (rows > 16 && ((B + rows - 1) / rows) * colb < (long long)umnn_num_cus() * 4) rows /= 2;
    long long rb = (B + rows - 1) / rows;
    if (rb > 64) rb = 64;
    if (rb < 1) rb = 1;
    return (int)rb;
}

extern "C" int umnn_made_relu_bwd_bias(float* g, const float* relu_out, long long B, int N, float* partial, int row_blocks, float* gb,
                                       void* stream) {
    if (B < 0 || N < 1 || row_blocks < 1) return umnn_fail(UMNN_EINVAL, "made_relu_bwd_bias: bad shape");
    if (!gb) return umnn_fail(UMNN_EINVAL, "made_relu_bwd_bias: null pointer");
    if (B == 0) return umnn_check(hipMemsetAsync(gb, 0, (size_t)N * sizeof(float), (hipStream_t)stream), "memset");
    if (!g || !partial) return umnn_fail(UMNN_EINVAL, "made_relu_bwd_bias: null pointer");
    if (row_blocks > B) row_blocks = (int)B;
    const long long rpb = (B + row_blocks - 1) / row_blocks;
    const int used = (int)((B + rpb - 1) / rpb);
    const unsigned colb = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(made_relu_bwd_bias_kernel, dim3(colb, (unsigned)used), dim3(256), 0, (hipStream_t)stream, g, relu_out, B, N, rpb, partial);
    hipLaunchKernelGGL(made_bias_finish_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, (hipStream_t)stream, partial, N, used, gb);
    return umnn_check(hipGetLastError(), "made_relu_bwd_bias launch");
}
