// Micro-benchmark round 2: candidate instructions for the flag path and mixes.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 2000;
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)

// each KIND issues 64 instructions per loop iteration, 4 independent chains
#define FOUR(op) R16(op(%0) op(%1) op(%2) op(%3))
#define OP_SUB(r) "v_sub_f32 " #r ", " #r ", %4\n"
#define OP_ALIGN(r) "v_alignbit_b32 " #r ", " #r ", %4, 31\n"
#define OP_ADDU(r) "v_add_u32 " #r ", " #r ", %4\n"
#define OP_LSHLADD(r) "v_lshl_add_u32 " #r ", " #r ", 1, %4\n"
#define OP_ANDOR(r) "v_and_or_b32 " #r ", " #r ", %4, %4\n"
#define OP_CND(r) "v_cndmask_b32 " #r ", " #r ", %4, vcc\n"
#define OP_MULC(r) "v_mul_f32 " #r ", " #r ", %4 clamp\n"
#define OP_FMA(r) "v_fma_f32 " #r ", " #r ", %4, %4\n"
#define OP_BFI(r) "v_bfi_b32 " #r ", %4, " #r ", %4\n"
#define OP_OR3(r) "v_or3_b32 " #r ", " #r ", %4, %4\n"
#define OP_MUL(r) "v_mul_f32 " #r ", " #r ", %4\n"
#define OP_ADDMAX(r) "v_add_f32 " #r ", " #r ", %4\n v_max_f32 " #r ", " #r ", %4\n"
#define OP_ADD3MAX(r) "v_add_f32 " #r ", " #r ", %4\n v_add_f32 " #r ", " #r ", %4\n v_add_f32 " #r ", " #r ", %4\n v_max_f32 " #r ", " #r ", %4\n"
#define OP_SUBC(r) "v_sub_f32 " #r ", " #r ", %4 clamp\n"
#define OP_ADDC(r) "v_addc_co_u32 " #r ", vcc, " #r ", " #r ", vcc\n"
#define OP_MIN(r) "v_min_f32 " #r ", " #r ", %4\n"
#define OP_OR(r) "v_or_b32 " #r ", " #r ", %4\n"
#define OP_LSHL(r) "v_lshlrev_b32 " #r ", 1, " #r "\n"
#define OP_ADDCO(r) "v_add_co_u32 " #r ", vcc, " #r ", %4\n"
#define OP_MAC(r) "v_fmac_f32 " #r ", %4, %4\n"
#define OP_CVT(r) "v_cvt_u32_f32 " #r ", " #r "\n"
#define OP_MOV(r) "v_mov_b32 " #r ", %4\n"
#define OP_SUBREV(r) "v_subrev_f32 " #r ", %4, " #r "\n"

template <int KIND>
__global__ void bench(float* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, c0 = 1e-7f;
    for (int it = 0; it < ITER; ++it) {
#define CASE(k, OP) if (KIND == k) asm volatile(FOUR(OP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c0) : "vcc");
        CASE(0, OP_SUB) CASE(1, OP_ALIGN) CASE(2, OP_ADDU) CASE(3, OP_LSHLADD) CASE(4, OP_ANDOR) CASE(5, OP_CND)
        CASE(6, OP_MULC) CASE(7, OP_FMA) CASE(8, OP_BFI) CASE(9, OP_OR3) CASE(10, OP_MUL)
        CASE(13, OP_SUBC) CASE(14, OP_ADDC) CASE(15, OP_MIN) CASE(16, OP_OR) CASE(17, OP_LSHL) CASE(18, OP_ADDCO)
        CASE(19, OP_MAC) CASE(20, OP_CVT) CASE(21, OP_MOV) CASE(22, OP_SUBREV)
        if (KIND == 11) asm volatile(R4(R4(OP_ADDMAX(%0) OP_ADDMAX(%1))) R4(R4(OP_ADDMAX(%2) OP_ADDMAX(%3))) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c0) : "vcc");
        if (KIND == 12) asm volatile(R4(OP_ADD3MAX(%0) OP_ADD3MAX(%1) OP_ADD3MAX(%2) OP_ADD3MAX(%3)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c0) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

// LDS gather: 16 ds_read_b32 per iteration from a 183x17-float table, row differs per lane
__global__ void lds_gather(float* out, int stride) {
    __shared__ float tab[183 * 20];
    for (int i = threadIdx.x; i < 183 * 20; i += blockDim.x) tab[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned row = (lane * 37 + 11) % 183;
    float acc = 0;
    for (int it = 0; it < ITER; ++it) {
        const char* base = (const char*)tab + row * stride * 4;
#pragma unroll
        for (int c = 0; c < 16; ++c) acc += *(const float*)(base + ((c * 7 + lane) & 3) * 4);
        row = (row + 3) % 183;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int KIND>
int run(const char* name, float* d_out, int per_iter = 64) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-26s", name);
    for (int wps : {1, 2, 4, 8}) {
        hipLaunchKernelGGL(bench<KIND>, dim3(256 * wps), dim3(256), 0, 0, d_out, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench<KIND>, dim3(256 * wps), dim3(256), 0, 0, d_out, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  w%d: %5.2f", wps, ms * 1e-3 * 2.4e9 / (double(per_iter) * ITER * wps));
    }
    printf("   cyc/instr/SIMD @2.4GHz\n");
    return 0;
}

int main() {
    float* d_out; CHECK(hipMalloc(&d_out, sizeof(float) * 256 * 256 * 8));
    run<0>("v_sub_f32", d_out); run<22>("v_subrev_f32", d_out); run<13>("v_sub_f32 clamp (VOP3)", d_out);
    run<10>("v_mul_f32", d_out); run<6>("v_mul_f32 clamp (VOP3)", d_out); run<7>("v_fma_f32", d_out); run<19>("v_fmac_f32", d_out);
    run<15>("v_min_f32", d_out); run<1>("v_alignbit_b32", d_out); run<2>("v_add_u32", d_out); run<18>("v_add_co_u32", d_out); run<14>("v_addc_co_u32", d_out);
    run<3>("v_lshl_add_u32", d_out); run<4>("v_and_or_b32", d_out); run<16>("v_or_b32", d_out); run<17>("v_lshlrev_b32", d_out);
    run<5>("v_cndmask_b32", d_out); run<8>("v_bfi_b32", d_out); run<9>("v_or3_b32", d_out); run<20>("v_cvt_u32_f32", d_out); run<21>("v_mov_b32", d_out);
    run<11>("mix add,max (1:1)", d_out); run<12>("mix add,add,add,max (3:1)", d_out);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int stride : {16, 17, 20}) {
        printf("lds gather stride %2d       ", stride);
        for (int wps : {1, 2, 4, 8}) {
            hipLaunchKernelGGL(lds_gather, dim3(256 * wps), dim3(256), 0, 0, d_out, stride);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(lds_gather, dim3(256 * wps), dim3(256), 0, 0, d_out, stride);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("  w%d: %5.2f", wps, ms * 1e-3 * 2.4e9 / (16.0 * ITER * wps * 4));
        }
        printf("   cyc/ds_read_b32/CU (4 SIMDs issuing)\n");
    }
    return 0;
}
