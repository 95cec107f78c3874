// HBM write bandwidth: what a pure coalesced 256-byte-row store stream reaches (the Forward
// fill's 12 B/cell are such a stream), plus hipMemset for comparison.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void wr4(float* p, size_t n_per_block, float v) {
    float* q = p + blockIdx.x * n_per_block;
    for(size_t i = threadIdx.x; i < n_per_block; i += blockDim.x) q[i] = v;
}
__global__ void wr16(float4* p, size_t n_per_block, float v) {
    float4* q = p + blockIdx.x * n_per_block;
    const float4 x{v, v, v, v};
    for(size_t i = threadIdx.x; i < n_per_block; i += blockDim.x) q[i] = x;
}
int main() {
    const size_t bytes = 24ull << 30;
    float* d; CHECK(hipMalloc(&d, bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for(int blocks : {512, 1024, 2048, 4096}) {
        for(int mode = 0; mode < 2; ++mode) {
            float best = 1e9;
            for(int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                if(mode == 0) hipLaunchKernelGGL(wr4, dim3(blocks), dim3(256), 0, 0, d, bytes / 4 / blocks, 1.0f);
                else hipLaunchKernelGGL(wr16, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<float4*>(d), bytes / 16 / blocks, 1.0f);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if(ms < best) best = ms;
            }
            printf("%s blocks %4d: %.2f ms  %.0f GB/s\n", mode ? "dwordx4" : "dword  ", blocks, best, bytes / best / 1e6);
        }
    }
    float best = 1e9;
    for(int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0)); CHECK(hipMemsetAsync(d, 0, bytes, 0)); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if(ms < best) best = ms;
    }
    printf("hipMemsetAsync: %.2f ms  %.0f GB/s\n", best, bytes / best / 1e6);
    return 0;
}
