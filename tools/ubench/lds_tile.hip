// Microbenchmark: LDS cost of the [point][slot] activation tile of the backward kernels in two layouts.
//   hipcc --offload-arch=gfx950 -O3 -w lds_tile.hip -o lds_tile && ./lds_tile
// A tile is one bf16 piece of 16 points x 64 feature slots.  Three access patterns hit it (cc_bwd_ws_kernel.h):
//   own b128   lane (g, p) reads / writes its own 16 bytes (K-step s): row p, 16-byte unit 2 g + s
//   tr 16x16   ds_read_b64_tr_b16 of a 16x16x16 operand: lane (g, p) -> row 4 g + (p >> 2), ushort column 16 tau + 4 (p & 3)
//   tr 32x32   ds_read_b64_tr_b16 of a 32x32x16 operand: lane (g, p) -> row 8 (g >> 1) + 4 half + (p >> 2), column 32 tau + 16 (g & 1) + 4 (p & 3)
// Layout P (shipped): rows of 72 ushorts (64 + 8 padding, 144 bytes).  Layout X (candidate): rows of 64 ushorts, the eight
// 16-byte units of a row XOR-swizzled with (row & 7) -- 11 % less LDS per tile (17 KB of the workgroup pipeline's 153 KB).
// Reported: cycles per access instruction for one wave and for eight waves of a workgroup hammering their own tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define NIT 2000

__device__ __forceinline__ int addrP(int row, int col) { return row * 72 + col; }                                   // ushort index
__device__ __forceinline__ int addrX(int row, int col) { return row * 64 + ((((col >> 3) ^ (row & 7)) << 3) | (col & 7)); }

template <int LAYOUT, int PATTERN>
__global__ __launch_bounds__(512) void k(unsigned long long* out, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned short sm[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, p = lane & 15;
    unsigned short* tile = sm + wid * 4 * 16 * 72;                 // four tiles per wave (tau / piece variety), padded size for both
    for (int i = lane; i < 4 * 16 * 72; i += 64) tile[i] = (unsigned short)i;
    __syncthreads();
    int a[4];
    for (int j = 0; j < 4; ++j) {
        int row, col;
        if (PATTERN == 0) { row = p; col = (2 * g + (j & 1)) * 8; }
        else if (PATTERN == 1) { row = 4 * g + (p >> 2); col = 16 * j + 4 * (p & 3); }
        else { row = 8 * (g >> 1) + 4 * (j & 1) + (p >> 2); col = 32 * (j >> 1) + 16 * (g & 1) + 4 * (p & 3); }
        a[j] = (j >> (PATTERN == 0 ? 1 : 2)) * 16 * 72 + (LAYOUT ? addrX(row, col) : addrP(row, col));
    }
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < NIT; ++it) {
        // sixteen reads in flight, one wait: throughput (issue + bank conflicts), not latency
        u32x4 v4[4]; unsigned long long v2[4];
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned ad = (unsigned)(uintptr_t)(tile + a[j]);
                if (PATTERN == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(v4[j]) : "v"(ad));
                else asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v2[j]) : "v"(ad));
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= PATTERN == 0 ? v4[j][0] : (unsigned)v2[j];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 8 + wid] = t1 - t0;
    sink[blockIdx.x * blockDim.x + tid] = acc;
}

template <int LAYOUT, int PATTERN>
void run(unsigned long long* out, unsigned* sink) {
    static const char* pat[] = {"own b128 ", "tr 16x16 ", "tr 32x32 "};
    for (int waves = 1; waves <= 8; waves *= 8) {
        hipLaunchKernelGGL((k<LAYOUT, PATTERN>), dim3(64), dim3(64 * waves), 8 * 4 * 16 * 72 * 2, 0, out, sink);
        hipDeviceSynchronize();
        unsigned long long h[64 * 8];
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0;
        for (int b = 0; b < 64; ++b) for (int w = 0; w < waves; ++w) s += (double)h[b * 8 + w];
        printf("%s layout %s  %d wave(s) per workgroup: %6.1f s_memtime ticks per access instruction\n", pat[PATTERN],
               LAYOUT ? "X (64-ushort rows, XOR swizzle)" : "P (72-ushort rows)            ", waves, s / (64.0 * waves) / (16.0 * NIT));
    }
}

int main() {
    unsigned long long* out; unsigned* sink;
    hipMalloc(&out, 64 * 8 * 8); hipMalloc(&sink, 64 * 512 * 4);
    run<0, 0>(out, sink); run<1, 0>(out, sink);
    run<0, 1>(out, sink); run<1, 1>(out, sink);
    run<0, 2>(out, sink); run<1, 2>(out, sink);
    return 0;
}
