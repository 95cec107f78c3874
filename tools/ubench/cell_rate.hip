// Issue-rate micro-benchmark of the exact DP-cell instruction stream (no memory).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 400;

#define HEAD "v_add_f32 %[t0], %[diag], %[s]\n\t v_add_f32 %[t1], %[ge], %[zl]\n\t"
#define PEND "v_alignbit_b32 %[aB], %[aB], %[pend], 31\n\t"
#define ADD(d,a,b) "v_add_f32 " d ", " a ", " b "\n\t"
#define SUB(d,a,b) "v_sub_f32 " d ", " a ", " b "\n\t"
// variant selectors: MAXOP / ALNOP can be replaced by fast ops to see what they cost
#ifndef MAXOP
#define MAXOP "v_max_f32"
#endif
#ifndef ALNOP
#define ALN(acc,d) "v_alignbit_b32 " acc ", " acc ", " d ", 31\n\t"
#else
#define ALN(acc,d) ALNOP " " acc ", " acc ", " d "\n\t"
#endif
#define MAX(d,a,b) MAXOP " " d ", " a ", " b "\n\t"
#define BODY \
    ADD("%[t2]","%[gs]","%[zl]") ADD("%[t3]","%[go]","%[t0]") ADD("%[t0]","%[ng]","%[t0]") MAX("%[zl]","%[t3]","%[t1]") \
    ADD("%[t4]","%[ng]","%[t0]") ADD("%[t5]","%[gs]","%[y]") SUB("%[t1]","%[t1]","%[t3]") MAX("%[t3]","%[t4]","%[t5]") \
    ADD("%[pend]","%[ng]","%[t2]") ALN("%[aC]","%[t1]") SUB("%[t1]","%[t4]","%[t5]") MAX("%[x]","%[t3]","%[pend]") \
    SUB("%[t4]","%[t3]","%[pend]") ADD("%[t5]","%[go]","%[t0]") ALN("%[aA]","%[t1]") ADD("%[t1]","%[ge]","%[y]") \
    ADD("%[t3]","%[go]","%[t2]") MAX("%[t0]","%[t5]","%[t1]") SUB("%[t2]","%[t5]","%[t1]") ALN("%[aA]","%[t4]") \
    "v_add_u32 %[addr], %[lds], %[boff]\n\t" MAX("%[y]","%[t0]","%[t3]") SUB("%[pend]","%[t0]","%[t3]") ALN("%[aB]","%[t2]")

__global__ void bench(float* out, float seed, float ng, float gs, float go, float ge) {
    float X[16], Y[16];
    for (int c = 0; c < 16; ++c) { X[c] = seed + c + threadIdx.x; Y[c] = seed - c; }
    float zl = seed, pend = 0, diag = seed * 2, s = 0.5f;
    unsigned aA = 0, aB = 0, aC = 0, lds = threadIdx.x, boff = 4, addr_acc = 0;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float x_new, t0, t1, t2, t3, t4, t5; unsigned addr;
            asm volatile(HEAD PEND BODY
                : [x] "=&v"(x_new), [y] "+v"(Y[c]), [zl] "+v"(zl), [pend] "+v"(pend), [aA] "+v"(aA), [aB] "+v"(aB), [aC] "+v"(aC),
                  [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [addr] "=&v"(addr)
                : [diag] "v"(diag), [s] "v"(s), [lds] "v"(lds), [boff] "v"(boff), [ng] "s"(ng), [gs] "s"(gs), [go] "s"(go), [ge] "s"(ge));
            diag = X[c]; X[c] = x_new; addr_acc += addr;
        }
    }
    float r = zl + pend + diag;
    for (int c = 0; c < 16; ++c) r += X[c] + Y[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + aA + aB + aC + addr_acc;
}

int main() {
    float* d_out; CHECK(hipMalloc(&d_out, sizeof(float) * 256 * 256 * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {1, 2, 3, 4, 8}) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(bench, dim3(256 * wps), dim3(256), 0, 0, d_out, 1.0f, -0.001f, -1.79f, -6.9f, -0.18f);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        }
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        // 16 cells x 28 instructions (+1 add_u32 accumulate by the compiler) per iteration
        printf("waves/SIMD %d: %.3f ms -> %.1f cycles per cell per SIMD (@2.4 GHz), %.2f cyc/instr\n", wps, ms,
               ms * 1e-3 * 2.4e9 / (16.0 * ITER * wps), ms * 1e-3 * 2.4e9 / (16.0 * ITER * wps * 29));
    }
    return 0;
}
