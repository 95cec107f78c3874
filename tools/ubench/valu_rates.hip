// Micro-benchmark: issue cost of the VALU instructions the DP cell uses, at 1..8
// waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int REP = 64;      // instructions per asm block
constexpr int ITER = 2000;   // loop iterations

#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)

template <int KIND>
__global__ void bench(float* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float c0 = 1e-7f, c1 = 2e-7f;
    unsigned u0 = threadIdx.x, u1 = 1;
    for (int it = 0; it < ITER; ++it) {
        if (KIND == 0) {  // v_add_f32, 8 independent chains
            asm volatile(R4(R4("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"))
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c0));
        } else if (KIND == 1) {  // v_pk_add_f32 on VGPR pairs, 4 independent chains
            asm volatile(R16("v_pk_add_f32 %0, %4, %0\n v_pk_add_f32 %1, %4, %1\n v_pk_add_f32 %2, %4, %2\n v_pk_add_f32 %3, %4, %3\n")
                         : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&c0));
        } else if (KIND == 2) {  // v_max3_f32
            asm volatile(R16("v_max3_f32 %0, %4, %0, %1\n v_max3_f32 %1, %4, %1, %2\n v_max3_f32 %2, %4, %2, %3\n v_max3_f32 %3, %4, %3, %0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c0));
        } else if (KIND == 3) {  // v_cmp + v_addc (bit accumulate)
            asm volatile(R16("v_cmp_gt_f32 vcc, %2, %3\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_cmp_gt_f32 vcc, %3, %2\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n")
                         : "+v"(u0), "+v"(u1) : "v"(a0), "v"(a1) : "vcc");
        } else if (KIND == 4) {  // v_cmp + s_nop + 2 v_writelane
            asm volatile(R16("v_cmp_gt_f32 vcc, %2, %3\n s_nop 0\n v_writelane_b32 %0, vcc_lo, 3\n v_writelane_b32 %1, vcc_hi, 3\n")
                         : "+v"(u0), "+v"(u1) : "v"(a0), "v"(a1) : "vcc");
        } else if (KIND == 5) {  // v_writelane only
            asm volatile(R64("v_writelane_b32 %0, s0, 3\n") : "+v"(u0));
        } else if (KIND == 6) {  // v_cmp only (to vcc)
            asm volatile(R64("v_cmp_gt_f32 vcc, %0, %1\n") : : "v"(a0), "v"(a1) : "vcc");
        } else if (KIND == 7) {  // v_max_f32
            asm volatile(R16("v_max_f32 %0, %4, %0\n v_max_f32 %1, %4, %1\n v_max_f32 %2, %4, %2\n v_max_f32 %3, %4, %3\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c0));
        } else if (KIND == 8) {  // dpp wave_shr
            asm volatile(R16("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (KIND == 9) {  // v_cmp to sgpr pair (e64) + v_cndmask reading it
            asm volatile(R16("v_cmp_gt_f32 s[10:11], %2, %3\n v_cndmask_b32 %0, %0, %1, s[10:11]\n v_cmp_gt_f32 s[12:13], %3, %2\n v_cndmask_b32 %1, %1, %0, s[12:13]\n")
                         : "+v"(u0), "+v"(u1) : "v"(a0), "v"(a1) : "s10", "s11", "s12", "s13");
        } else if (KIND == 10) {  // s_store_dwordx2 of masks
            asm volatile(R16("v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n") : "+v"(a0) : "v"(c0));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + u0 + u1 + c1;
}

template <int KIND>
int run(const char* name, float* d_out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD
        const int block = 256;                 // 4 waves -> one per SIMD
        const int grid = 256 * wps;            // one block per CU per wps
        hipLaunchKernelGGL(bench<KIND>, dim3(grid), dim3(block), 0, 0, d_out, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench<KIND>, dim3(grid), dim3(block), 0, 0, d_out, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_wave = double(REP) * ITER;
        // cycles per instruction per SIMD at an assumed 2.4 GHz: time * clk / (instr per wave * waves per simd)
        const double cyc = ms * 1e-3 * 2.4e9 / (instr_per_wave * wps);
        printf("%-28s waves/SIMD %d: %8.3f ms  -> %.2f cyc/instr/SIMD (@2.4GHz)\n", name, wps, ms, cyc);
    }
    return 0;
}

int main() {
    float* d_out; CHECK(hipMalloc(&d_out, sizeof(float) * 256 * 256 * 8));
    run<0>("v_add_f32", d_out);
    run<1>("v_pk_add_f32 (vgpr)", d_out);
    run<2>("v_max3_f32", d_out);
    run<7>("v_max_f32", d_out);
    run<3>("v_cmp+v_addc", d_out);
    run<4>("v_cmp+s_nop+2xwritelane (4)", d_out);
    run<5>("v_writelane_b32", d_out);
    run<6>("v_cmp_gt_f32 vcc", d_out);
    run<8>("v_mov_b32_dpp wave_shr", d_out);
    run<9>("v_cmp_e64+v_cndmask", d_out);
    run<10>("v_add_f32 dependent chain", d_out);
    return 0;
}
