// Which bits of HW_REG_HW_ID / HW_REG_XCC_ID tell a wavefront's SIMD on gfx950?  Launch as many 2-wave
// workgroups as viterbi_ck does, each wave records its registers; the host prints how many distinct values every
// candidate field takes and how many waves share a (xcc, se, sh, cu, simd) key.
// build: hipcc --offload-arch=gfx950 -O2 -o hwid_probe tools/ubench/hwid_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ __launch_bounds__(128, 4) void probe(uint32_t* out, int spin) {
    __shared__ float pad[3000];  // ~12 KB like viterbi_ck
    pad[threadIdx.x] = 1.0f;
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float v = pad[threadIdx.x];
    for(int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;  // stay resident until the whole grid is
    const uint32_t wave = blockIdx.x * 2 + threadIdx.x / 64;
    if((threadIdx.x & 63) == 0) {
        out[wave * 2] = hw;
        out[wave * 2 + 1] = xcc + (v > 1e30f ? 1u : 0u);
    }
}
int main() {
    const int waves = 4096;
    uint32_t* d;
    hipMalloc(&d, waves * 8);
    hipLaunchKernelGGL(probe, dim3(waves / 2), dim3(128), 0, 0, d, 2000000);
    std::vector<uint32_t> h(waves * 2);
    hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
    for(int lo = 0; lo < 32; lo += 4) {
        std::set<uint32_t> s;
        for(int w = 0; w < waves; ++w) s.insert((h[2 * w] >> lo) & 0xf);
        std::printf("HW_ID bits %2d..%2d: %zu distinct values:", lo, lo + 3, s.size());
        for(uint32_t v : s) std::printf(" %x", v);
        std::printf("\n");
    }
    std::set<uint32_t> xs;
    for(int w = 0; w < waves; ++w) xs.insert(h[2 * w + 1]);
    std::printf("XCC_ID: %zu distinct raw values:", xs.size());
    for(uint32_t v : xs) std::printf(" %x", v);
    std::printf("\n");
    std::map<uint64_t, int> per;
    for(int w = 0; w < waves; ++w) {
        const uint32_t hw = h[2 * w];
        const uint64_t key = (static_cast<uint64_t>(h[2 * w + 1] & 0xf) << 32) | (hw & 0xff30u);  // se, sh, cu, simd (pipe + wave masked out)
        per[key]++;
    }
    std::map<int, int> hist;
    for(auto& kv : per) hist[kv.second]++;
    std::printf("%zu distinct (xcc, HW_ID & 0xff30) keys; waves per key:", per.size());
    for(auto& kv : hist) std::printf(" %d x%d", kv.first, kv.second);
    std::printf("\n");
    return 0;
}
