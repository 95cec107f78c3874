// Micro-benchmark (round 5): what the gfx950 vector unit charges for the instruction classes of the exact Forward's
// `plus` (glibc expf in fp64, log1pf with reciprocal + Newton steps) at 1, 2 and 4 wavefronts per SIMD -- as four
// independent chains per wavefront (issue cost) and as ONE dependent chain (latency).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/_build/fwd_rates tools/ubench/fwd_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 2000;
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
// 64 instructions per loop iteration: four independent chains (registers %0..%3) or one chain (%0 only)
#define FOUR(op) R16(op(%0) op(%1) op(%2) op(%3))
#define ONE(op) R16(op(%0) op(%0) op(%0) op(%0))
#define D_FMA(r) "v_fma_f64 " #r ", " #r ", %4, %4\n"
#define D_MUL(r) "v_mul_f64 " #r ", " #r ", %4\n"
#define D_ADD(r) "v_add_f64 " #r ", " #r ", %4\n"
#define D_LSHLADD(r) "v_lshl_add_u64 " #r ", " #r ", 1, %4\n"
#define F_FMA(r) "v_fma_f32 " #r ", " #r ", %4, %4\n"
#define F_MUL(r) "v_mul_f32 " #r ", " #r ", %4\n"
#define F_ADD(r) "v_add_f32 " #r ", " #r ", %4\n"
#define F_RCP(r) "v_rcp_f32 " #r ", " #r "\n"
#define F_CND(r) "v_cndmask_b32 " #r ", " #r ", %4, vcc\n"
#define F_CND_S(r) "v_cndmask_b32 " #r ", " #r ", %4, s[10:11]\n"
#define F_CND_IMM(r) "v_cndmask_b32 " #r ", 1.0, " #r ", vcc\n"
#define F_CMP(r) "v_cmp_lt_f32 vcc, " #r ", %4\n"
#define F_CMP_S(r) "v_cmp_lt_f32 s[10:11], " #r ", %4\n"
#define F_CMPCND(r) "v_cmp_lt_f32 vcc, " #r ", %4\n v_cndmask_b32 " #r ", " #r ", %4, vcc\n"
#define F_CMPCND2(r) "v_cmp_lt_f32 vcc, " #r ", %4\n v_cndmask_b32 " #r ", " #r ", %4, vcc\n v_cndmask_b32 " #r ", " #r ", %4, vcc\n v_cndmask_b32 " #r ", " #r ", %4, vcc\n"
#define F_CMPADDCND(r) "v_cmp_lt_f32 vcc, " #r ", %4\n v_add_f32 " #r ", " #r ", %4\n v_add_f32 " #r ", " #r ", %4\n v_cndmask_b32 " #r ", " #r ", %4, vcc\n"
#define F_AND(r) "v_and_b32 " #r ", " #r ", %4\n"
#define F_LSHL(r) "v_lshlrev_b32 " #r ", 1, " #r "\n"
#define F_BFI(r) "v_bfi_b32 " #r ", %4, " #r ", %4\n"
#define F_DIVSCALE(r) "v_div_scale_f32 " #r ", vcc, " #r ", %4, " #r "\n"
#define F_FMAC(r) "v_fmac_f32 " #r ", %4, %4\n"
#define F_SUB(r) "v_sub_f32 " #r ", " #r ", %4\n"
#define F_MAX(r) "v_max_f32 " #r ", " #r ", %4\n"

template <int KIND, bool DEP>
__global__ void bench64(double* out, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, c0 = 1.0000001;
    for(int it = 0; it < ITER; ++it) {
#define CASE(k, OP)                                                                                                   \
    if(KIND == k) {                                                                                                   \
        if(DEP) asm volatile(ONE(OP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c0));                             \
        else asm volatile(FOUR(OP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c0));                               \
    }
        CASE(0, D_FMA) CASE(1, D_MUL) CASE(2, D_ADD) CASE(3, D_LSHLADD)
#undef CASE
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <int KIND, bool DEP>
__global__ void bench32(float* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, c0 = 1.0000001f;
    for(int it = 0; it < ITER; ++it) {
#define CASE(k, OP)                                                                                                   \
    if(KIND == k) {                                                                                                   \
        if(DEP) asm volatile(ONE(OP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c0) : "vcc", "s10", "s11");       \
        else asm volatile(FOUR(OP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c0) : "vcc", "s10", "s11");         \
    }
        CASE(0, F_FMA) CASE(1, F_MUL) CASE(2, F_ADD) CASE(3, F_RCP) CASE(4, F_CND) CASE(5, F_MAX)
        CASE(6, F_CND_S) CASE(7, F_CND_IMM) CASE(8, F_CMP) CASE(9, F_CMP_S) CASE(10, F_CMPCND) CASE(11, F_AND) CASE(12, F_LSHL)
        CASE(13, F_BFI) CASE(14, F_DIVSCALE) CASE(15, F_FMAC) CASE(16, F_SUB) CASE(17, F_CMPCND2) CASE(18, F_CMPADDCND)
#undef CASE
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
// conversions: f32 -> f64 -> f32 round trips (two instructions per link), four chains or one
template <bool DEP>
__global__ void bench_cvt(float* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    double t0, t1, t2, t3;
    for(int it = 0; it < ITER; ++it) {
#define LINK(a, t) "v_cvt_f64_f32 %" #t ", %" #a "\n v_cvt_f32_f64 %" #a ", %" #t "\n"
        if(DEP) asm volatile(R16(LINK(0, 4) LINK(0, 4)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3));
        else asm volatile(R4(R4(LINK(0, 4) LINK(1, 5)) ) R4(R4(LINK(2, 6) LINK(3, 7))) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3));
#undef LINK
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <typename K, typename T>
int time_kernel(const char* name, K kernel, T* d_out, T seed, double per_iter) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-34s", name);
    for(int wps : {1, 2, 4}) {
        hipLaunchKernelGGL(kernel, dim3(256 * wps), dim3(256), 0, 0, d_out, seed);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kernel, dim3(256 * wps), dim3(256), 0, 0, d_out, seed);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  w%d: %6.2f", wps, ms * 1e-3 * 2.4e9 / (per_iter * ITER * wps));
    }
    printf("   cycles per instruction and SIMD at 2.4 GHz\n");
    return 0;
}

int main() {
    double* d64;
    float* d32;
    CHECK(hipMalloc(&d64, sizeof(double) * 256 * 256 * 4));
    CHECK(hipMalloc(&d32, sizeof(float) * 256 * 256 * 4));
#define RUN64(k, name)                                                              \
    time_kernel(name " (4 chains)", bench64<k, false>, d64, 1.0, 64.0);             \
    time_kernel(name " (1 dependent chain)", bench64<k, true>, d64, 1.0, 64.0);
#define RUN32(k, name)                                                              \
    time_kernel(name " (4 chains)", bench32<k, false>, d32, 1.0f, 64.0);            \
    time_kernel(name " (1 dependent chain)", bench32<k, true>, d32, 1.0f, 64.0);
    RUN64(0, "v_fma_f64") RUN64(1, "v_mul_f64") RUN64(2, "v_add_f64") RUN64(3, "v_lshl_add_u64")
    RUN32(0, "v_fma_f32") RUN32(15, "v_fmac_f32") RUN32(1, "v_mul_f32") RUN32(2, "v_add_f32") RUN32(16, "v_sub_f32") RUN32(3, "v_rcp_f32") RUN32(5, "v_max_f32")
    RUN32(4, "v_cndmask_b32 vcc") RUN32(6, "v_cndmask_b32 s[10:11]") RUN32(7, "v_cndmask_b32 1.0,v,vcc") RUN32(8, "v_cmp_lt_f32 vcc") RUN32(9, "v_cmp_lt_f32 s[10:11]")
    time_kernel("v_cmp+v_cndmask pairs (4 chains)", bench32<10, false>, d32, 1.0f, 128.0);
    time_kernel("v_cmp + 3 v_cndmask vcc (4 chains)", bench32<17, false>, d32, 1.0f, 256.0);
    time_kernel("v_cmp, 2 v_add, v_cndmask (4 ch)", bench32<18, false>, d32, 1.0f, 256.0);
    RUN32(11, "v_and_b32") RUN32(12, "v_lshlrev_b32") RUN32(13, "v_bfi_b32") RUN32(14, "v_div_scale_f32")
    time_kernel("v_cvt_f64_f32+v_cvt_f32_f64 (4 ch)", bench_cvt<false>, d32, 1.0f, 64.0);
    time_kernel("v_cvt_f64_f32+v_cvt_f32_f64 (dep)", bench_cvt<true>, d32, 1.0f, 64.0);
    return 0;
}
