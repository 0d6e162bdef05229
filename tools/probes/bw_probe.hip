// bw_probe.hip — device memory ceilings on the box the bench runs on: pure read, pure write (plain / nt stores), copy, and the
// 1 R : 5 W mix of the threshold+segment stage.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/bw_probe tools/probes/bw_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_fill(u4 *dst, size_t n16, int nt) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; const size_t step = (size_t)gridDim.x * 256;
    const u4 v = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (; i < n16; i += step) { if (nt) __builtin_nontemporal_store(v, &dst[i]); else dst[i] = v; }
}
__global__ __launch_bounds__(256) void k_read(const u4 *src, size_t n16, uint32_t *sink) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; const size_t step = (size_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (; i < n16; i += step) { u4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) *sink = acc;
}
__global__ __launch_bounds__(256) void k_copy(const u4 *src, u4 *dst, size_t n16) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; const size_t step = (size_t)gridDim.x * 256;
    for (; i < n16; i += step) dst[i] = src[i];
}
// per 16 source bytes: 16 bytes out to A and 64 bytes out to B (the stage's 1 R : 1 W : 4 W)
__global__ __launch_bounds__(256) void k_mix(const u4 *src, u4 *a, u4 *b, size_t n16, int nt) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; const size_t step = (size_t)gridDim.x * 256;
    for (; i < n16; i += step) {
        const u4 v = src[i];
        if (nt) { __builtin_nontemporal_store(v, &a[i]);
#pragma unroll
            for (int k = 0; k < 4; k++) __builtin_nontemporal_store(v, &b[4 * i + k]); }
        else { a[i] = v;
#pragma unroll
            for (int k = 0; k < 4; k++) b[4 * i + k] = v; }
    }
}
int main() {
    const size_t src_bytes = (size_t)1280 * 800 * 256, n16 = src_bytes / 16;
    u4 *src, *a, *b; uint32_t *sink;
    CK(hipMalloc(&src, src_bytes)); CK(hipMalloc(&a, src_bytes)); CK(hipMalloc(&b, 4 * src_bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(src, 1, src_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grids[3] = {2048, 8192, 65536};
    for (int g = 0; g < 3; g++) for (int test = 0; test < 7; test++) {
        float best = 1e9f;
        for (int rep = 0; rep < 12; rep++) {
            CK(hipEventRecord(e0));
            switch (test) {
            case 0: hipLaunchKernelGGL(k_fill, dim3(grids[g]), dim3(256), 0, 0, b, 4 * n16, 0); break;
            case 1: hipLaunchKernelGGL(k_fill, dim3(grids[g]), dim3(256), 0, 0, b, 4 * n16, 1); break;
            case 2: hipLaunchKernelGGL(k_read, dim3(grids[g]), dim3(256), 0, 0, (const u4 *)b, 4 * n16, sink); break;
            case 3: hipLaunchKernelGGL(k_copy, dim3(grids[g]), dim3(256), 0, 0, (const u4 *)src, a, n16); break;
            case 4: hipLaunchKernelGGL(k_mix, dim3(grids[g]), dim3(256), 0, 0, (const u4 *)src, a, b, n16, 0); break;
            case 5: hipLaunchKernelGGL(k_mix, dim3(grids[g]), dim3(256), 0, 0, (const u4 *)src, a, b, n16, 1); break;
            case 6: CK(hipMemsetAsync(b, 0xFF, 4 * src_bytes, 0)); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 2 && ms < best) best = ms;
        }
        const double bytes[7] = {4.0 * src_bytes, 4.0 * src_bytes, 4.0 * src_bytes, 2.0 * src_bytes, 6.0 * src_bytes, 6.0 * src_bytes, 4.0 * src_bytes};
        const char *nm[7] = {"fill", "fill_nt", "read", "copy", "mix1R5W", "mix1R5W_nt", "hipMemset"};
        printf("grid %6d %-11s %.3f ms  %.0f GB/s\n", grids[g], nm[test], best, bytes[test] / best / 1e6);
    }
    return 0;
}
