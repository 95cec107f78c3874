// Can the scalar unit carry the decision masks?  Each wave issues, per "step", NST
// s_store_dwordx4 (64 B of SGPR data each) interleaved with VALU work, like a DP step
// whose v_cmp results (lane masks in SGPRs) go straight to memory.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NST, int NVALU, bool kMov>
__global__ __launch_bounds__(256) void k(unsigned* out, float* sink, int steps, float seed) {
    const unsigned wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) / 64);
    unsigned long long p = reinterpret_cast<unsigned long long>(out) + static_cast<unsigned long long>(wave) * steps * NST * 16ull;
    p = static_cast<unsigned long long>(static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<unsigned>(p)))) |
        (static_cast<unsigned long long>(static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<unsigned>(p >> 32)))) << 32);
    float a = seed + threadIdx.x, b = seed * 2, c = seed * 3, d = seed * 4;
    for(int s = 0; s < steps; ++s) {
        unsigned tag = wave * 65536u + s;
#pragma unroll
        for(int q = 0; q < (NST > 0 ? NST : 1); ++q) {
#pragma unroll
            for(int v = 0; v < NVALU / (NST > 0 ? NST : 1) / 4; ++v)
                asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %2, %2, %3\n\tv_add_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if(NST > 0) {
                if(kMov)
                    asm volatile("s_mov_b32 s20, %1\n\ts_mov_b32 s21, %1\n\ts_mov_b32 s22, %1\n\ts_mov_b32 s23, %1\n\t"
                                 "s_store_dwordx4 s[20:23], %0, 0x0" ::"s"(p), "s"(tag + (q << 8)) : "s20", "s21", "s22", "s23", "memory");
                else
                    asm volatile("s_store_dwordx4 s[20:23], %0, 0x0" ::"s"(p) : "memory");
                p += 16;
            }
        }
    }
    asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}

template <int NST, int NVALU, bool kMov> int run(unsigned* d_out, float* d_sink, int wps, size_t out_bytes) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int steps = 200;
    float best = 1e9;
    for(int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(d_out, 0, out_bytes));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<NST, NVALU, kMov>), dim3(256 * wps), dim3(256), 0, 0, d_out, d_sink, steps, 1.0f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if(ms < best) best = ms;
    }
    // verify
    int bad = 0;
    if(NST > 0 && kMov) {
        const size_t nwaves = 256ull * wps * 4, n = nwaves * steps * NST * 4;
        std::vector<unsigned> h(n);
        CHECK(hipMemcpy(h.data(), d_out, n * 4, hipMemcpyDeviceToHost));
        for(size_t w = 0; w < nwaves && bad < 5; ++w)
            for(int s = 0; s < steps; ++s)
                for(int q = 0; q < NST; ++q)
                    for(int e = 0; e < 4; ++e)
                        if(h[((w * steps + s) * NST + q) * 4 + e] != static_cast<unsigned>(w * 65536u + s + (q << 8))) ++bad;
    }
    printf("mov=%d NST %2d NVALU %3d waves/SIMD %d: %.3f ms  %.1f ns/step/SIMD  bad=%d\n", int(kMov), NST, NVALU, wps, best, best * 1e6 / (steps * wps), bad);
    return 0;
}
int main() {
    const size_t out_bytes = 256ull * 4 * 4 * 200 * 40 * 16;
    unsigned* d_out; float* d_sink;
    CHECK(hipMalloc(&d_out, out_bytes)); CHECK(hipMalloc(&d_sink, 256 * 256 * 4 * 4));
    for(int wps : {1, 2, 3}) {
        if(run<0, 320, true>(d_out, d_sink, wps, out_bytes)) return 1;
        if(run<10, 320, false>(d_out, d_sink, wps, out_bytes)) return 1;
        if(run<20, 320, false>(d_out, d_sink, wps, out_bytes)) return 1;
        if(run<40, 320, false>(d_out, d_sink, wps, out_bytes)) return 1;
        if(run<40, 320, true>(d_out, d_sink, wps, out_bytes)) return 1;
    }
    return 0;
}
