// How accurate -- and how BIASED -- are the hardware transcendentals the fast Forward `plus` is made of?
//   c_fast(y) = log2(1 + exp2(y * log2e)) * ln2 + resid      (common.hpp: log_plus)
// against c(y) = log1p(exp(y)) in fp64, y in (-16, 0]; separately v_exp_f32 on [-23, 0] and v_log_f32 on [1, 2].
// A one-sided error accumulates linearly over the ~1 000 cells of a Forward path, a centred one like sqrt(n).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while(0)

__global__ void k_exp2(const float* in, float* out, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if(i < n) out[i] = __builtin_amdgcn_exp2f(in[i]); }
__global__ void k_log2(const float* in, float* out, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if(i < n) out[i] = __builtin_amdgcn_logf(in[i]); }
__global__ void k_c(const float* in, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) return;
    constexpr float kLog2e = 1.44269504088896340736f, kLn2 = 0.69314718055994530942f;
    const float t = in[i] * kLog2e;
    const float e = __builtin_amdgcn_exp2f(t);
    const float u = 1.0f + e;
    const float resid = e - (u - 1.0f);
    out[i] = __builtin_fmaf(__builtin_amdgcn_logf(u), kLn2, resid);
}

template <typename K, typename F>
int sweep(const char* name, K kern, float lo, float hi, F truth, bool relative, int bins = 0) {
    const int n = 1 << 24;
    std::vector<float> in(n), out(n);
    for(int i = 0; i < n; ++i) in[i] = lo + (hi - lo) * ((i + 0.5f) / n);
    float *d_in, *d_out;
    CHECK(hipMalloc(&d_in, n * 4)); CHECK(hipMalloc(&d_out, n * 4));
    CHECK(hipMemcpy(d_in, in.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kern, dim3(n / 256), dim3(256), 0, 0, d_in, d_out, n);
    CHECK(hipMemcpy(out.data(), d_out, n * 4, hipMemcpyDeviceToHost));
    double sum = 0, sq = 0, mx = 0, sum_rn = 0, sq_rn = 0;
    for(int i = 0; i < n; ++i) {
        const double t = truth(double(in[i]));
        const double scale = relative ? std::fabs(t) : 1.0;
        const double err = (double(out[i]) - t) / scale;
        const double err_rn = (double(float(t)) - t) / scale;  // what a correctly rounded result would leave
        sum += err; sq += err * err; mx = std::max(mx, std::fabs(err));
        sum_rn += err_rn; sq_rn += err_rn * err_rn;
    }
    if(bins > 0) {
        printf("%s, mean error per bin of the argument:\n   ", name);
        for(int b = 0; b < bins; ++b) {
            double sb = 0;
            const int per = n / bins;
            for(int i = b * per; i < (b + 1) * per; ++i) sb += (double(out[i]) - truth(double(in[i]))) / (relative ? std::fabs(truth(double(in[i]))) : 1.0);
            printf(" %+.2e", sb / per);
        }
        printf("\n");
    }
    printf("%-44s mean %+.3e  rms %.3e  max %.3e   (correctly rounded fp32: mean %+.3e rms %.3e)  [%s error]\n", name, sum / n, std::sqrt(sq / n), mx,
           sum_rn / n, std::sqrt(sq_rn / n), relative ? "relative" : "absolute");
    CHECK(hipFree(d_in)); CHECK(hipFree(d_out));
    return 0;
}

int main() {
    if(sweep("v_exp_f32(t), t in [-23, 0]", k_exp2, -23.0f, 0.0f, [](double t) { return std::exp2(t); }, true)) return 1;
    if(sweep("v_exp_f32(t), t in [-3, 0]", k_exp2, -3.0f, 0.0f, [](double t) { return std::exp2(t); }, true)) return 1;
    if(sweep("v_log_f32(u), u in [1, 2]", k_log2, 1.0f, 2.0f, [](double u) { return std::log2(u); }, false, 32)) return 1;
    if(sweep("v_log_f32(u), u in [1, 1.0625]", k_log2, 1.0f, 1.0625f, [](double u) { return std::log2(u); }, false, 16)) return 1;
    if(sweep("v_exp_f32(t), t in [-1, 0]", k_exp2, -1.0f, 0.0f, [](double t) { return std::exp2(t); }, true, 16)) return 1;
    if(sweep("c_fast(y), y in [-4, 0]", k_c, -4.0f, 0.0f, [](double y) { return std::log1p(std::exp(y)); }, false, 32)) return 1;
    for(float lo : {-16.0f, -8.0f, -4.0f, -2.0f, -1.0f, -0.5f}) {
        char name[64];
        snprintf(name, sizeof name, "c_fast(y) vs log1p(exp(y)), y in [%g, %g]", lo, lo / 2 > -0.3f ? 0.0f : lo / 2);
        if(sweep(name, k_c, lo, lo / 2 > -0.3f ? 0.0f : lo / 2, [](double y) { return std::log1p(std::exp(y)); }, false)) return 1;
    }
    return 0;
}
