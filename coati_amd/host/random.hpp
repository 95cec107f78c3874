// Host side of the RNG that `coati sample` consumes: seeding only -- the draws
// themselves are made by the device walkers (coati_amd/csrc/sampleback.hip).
//
// Mirrors contrib/random/random.hpp of the reference (fragmites::random):
//   Lehmer64Fast (128-bit multiplicative congruential, state forced odd)   :80-136
//   SeedSeq<8> multilinear Weyl hash (hash A to seed, hash B to generate)   :334-398
//   string_seed_seq: decimal int32 strings as numbers, others FNV-hashed    :465-472,522-540
//   Random::Seed(seed_seq)                                                  :408-413
#ifndef COATI_AMD_HOST_RANDOM_HPP
#define COATI_AMD_HOST_RANDOM_HPP

#include <array>
#include <charconv>
#include <cstdint>
#include <cstring>
#include <string>
#include <string_view>
#include <vector>

namespace coati_amd {

class random_t {
   public:
    using state_type = unsigned __int128;

    random_t() { state_ = static_cast<state_type>(0x9f57c403d06c42fcULL) | 1; }  // default of random.hpp:99

    // rand.Seed(string_seed_seq(first, last))
    void seed(const std::vector<std::string>& seeds) {
        std::vector<uint32_t> user;
        for(const std::string& s : seeds) {
            std::string_view sv{s};
            int32_t v = 0;
            auto [p, ec] = std::from_chars(sv.data(), sv.data() + sv.size(), v, 10);
            if(ec == std::errc() && p == sv.data() + sv.size()) {
                user.push_back(static_cast<uint32_t>(v));
            } else {
                uint32_t h = 2166136261U;
                for(const char ch : sv) h = (h * 16777619U) ^ static_cast<uint32_t>(static_cast<int>(ch));
                user.push_back(h);
            }
        }
        std::array<uint32_t, 8> inner{};
        std::array<uint32_t, 4> outer{};
        weyl_hash(0x3423da0b87484307ULL, user.data(), user.size(), inner.data(), inner.size());
        weyl_hash(0xdf8b06c40fa44478ULL, inner.data(), inner.size(), outer.data(), outer.size());
        state_type st = 0;
        std::memcpy(&st, outer.data(), sizeof(st));
        state_ = st | 1;
    }

    uint64_t bits() {
        state_ *= static_cast<state_type>(0xda942042e4dd58b5ULL);
        return static_cast<uint64_t>(state_ >> 64);
    }
    float f24() { return static_cast<float>(static_cast<int64_t>(bits() >> 40)) / 16777216.0f; }

    uint64_t lo() const { return static_cast<uint64_t>(state_); }
    uint64_t hi() const { return static_cast<uint64_t>(state_ >> 64); }
    void set_state(uint64_t lo, uint64_t hi) { state_ = (static_cast<state_type>(hi) << 64) | lo; }

   private:
    static void weyl_hash(uint64_t init, const uint32_t* in, std::size_t n_in, uint32_t* out, std::size_t n_out) {
        constexpr uint64_t inc = 0x9e3779b97f4a7c15ULL;
        uint64_t w = init;
        for(std::size_t o = 0; o < n_out; ++o) {
            uint64_t sum = (w += inc);
            for(std::size_t q = 0; q < n_in; ++q) sum += (w += inc) * in[q];
            sum += (w += inc);
            out[o] = static_cast<uint32_t>(sum >> 32);
        }
    }
    state_type state_;
};

}  // namespace coati_amd
#endif
