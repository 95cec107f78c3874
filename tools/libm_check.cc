// Pin coati_amd/csrc/glibc_math.hpp against the libm of this machine: the header (the very source
// the GPU kernels compile) is built for the HOST and compared with expf / log1pf / logf on every
// float of the ranges the log-semiring path can produce.
//   g++ -O2 -std=c++17 -mfma -ffp-contract=off -o libm_check tools/libm_check.cc -lm
//   ./libm_check          exhaustive (about a minute)        ./libm_check quick   every 257th float
// Exit status 0 = no mismatch.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

#define COATI_MATH_FN static inline
#include "../coati_amd/csrc/glibc_math.hpp"

using namespace coati_hip_detail::libm;
static const uint64_t kTab[32] = {COATI_EXP2F_TABLE};

static uint32_t bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
static float from_bits(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

template <typename Ref, typename Mine>
static unsigned long sweep(const char* name, uint32_t lo, uint32_t hi, uint32_t stride, Ref ref, Mine mine) {
    unsigned long n = 0, bad = 0;
    for(uint64_t u = lo; u <= hi; u += stride) {
        const float x = from_bits(static_cast<uint32_t>(u));
        const float a = ref(x), b = mine(x);
        if(bits(a) != bits(b) && !(a != a && b != b)) {
            if(bad < 5) std::printf("  %s mismatch at x = %a: libm %a, restatement %a\n", name, x, a, b);
            ++bad;
        }
        ++n;
    }
    std::printf("%-7s %lu inputs, %lu mismatches\n", name, n, bad);
    return bad;
}

int main(int argc, char** argv) {
    const uint32_t stride = (argc > 1 && std::strcmp(argv[1], "quick") == 0) ? 257u : 1u;
    unsigned long bad = 0;
    // expf: -0.0 down to -104 (below that the result is +0), plus the far tail sampled
    bad += sweep("expf", bits(-0.0f), bits(-104.0f), stride, [](float x) { return ::expf(x); }, [](float x) { return expf_nonpos(x, kTab); });
    bad += sweep("expf<<", bits(-104.0f), bits(-3.4e38f), 4099u * stride, [](float x) { return ::expf(x); }, [](float x) { return expf_nonpos(x, kTab); });
    bad += sweep("log1pf", bits(0.0f), bits(1.0f), stride, [](float x) { return ::log1pf(x); }, [](float x) { return log1pf_unit(x); });
    bad += sweep("log1p/m", bits(0x1p-29f), bits(1.0f), stride, [](float x) { return ::log1pf(x); }, [](float x) { return log1pf_mid(x); });
    // the two-wide form (round 5, narrow Forward strips): both halves, the partner input running through the range the other way
    bad += sweep("l1p/x2a", bits(0x1p-29f), bits(1.0f), stride, [](float x) { return ::log1pf(x); }, [](float x) {
        float r0, r1;
        log1pf_mid_x2(x, from_bits(bits(0x1p-29f) + bits(1.0f) - bits(x)), r0, r1);
        return r0;
    });
    bad += sweep("l1p/x2b", bits(0x1p-29f), bits(1.0f), stride, [](float x) { return ::log1pf(x); }, [](float x) {
        float r0, r1;
        log1pf_mid_x2(from_bits(bits(0x1p-29f) + bits(1.0f) - bits(x)), x, r0, r1);
        return r1;
    });
    bad += sweep("logf", bits(0x1p-126f), bits(4.0f), stride, [](float x) { return ::logf(x); }, [](float x) { return logf_pos(x); });
    return bad == 0 ? 0 : 1;
}
