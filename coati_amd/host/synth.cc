#include "synth.hpp"

#include <cmath>

#include "codon.hpp"

namespace coati_amd {

namespace {
struct splitmix64 {
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9e3779b97f4a7c15ULL);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    }
    double uniform() { return static_cast<double>(next() >> 11) * 0x1.0p-53; }     // [0,1)
    uint64_t below(uint64_t n) { return static_cast<uint64_t>(uniform() * n); }     // [0,n)
};
constexpr char kNt[] = "ACGT";
}  // namespace

void synth_pair(uint64_t index, const synth_params_t& prm, std::string& anc, std::string& des) {
    splitmix64 rng{prm.seed_base + index};
    anc.clear();
    anc.reserve(prm.n_codons * 3);
    for(uint32_t c = 0; c < prm.n_codons; ++c) {
        const int cod = cod61_to_64(static_cast<int>(rng.below(61)));
        anc.push_back(kNt[(cod >> 4) & 3]);
        anc.push_back(kNt[(cod >> 2) & 3]);
        anc.push_back(kNt[cod & 3]);
    }
    des = anc;
    for(char& ch : des) {
        if(rng.uniform() < prm.sub_rate) {
            const int cur = nt16(static_cast<unsigned char>(ch));
            ch = kNt[(cur + 1 + static_cast<int>(rng.below(3))) & 3];
        }
    }
    // Poisson(lambda) by multiplication of uniforms (Knuth)
    int events = 0;
    for(double prod = rng.uniform(), limit = std::exp(-prm.indel_lambda); prod > limit; prod *= rng.uniform()) ++events;
    const double q = 1.0 - 1.0 / prm.indel_mean_len;  // P(length > k) = q^k
    for(int e = 0; e < events; ++e) {
        const bool insertion = rng.uniform() < 0.5;
        const double u = 1.0 - rng.uniform();  // (0,1]
        const std::size_t len = 1 + static_cast<std::size_t>(std::floor(std::log(u) / std::log(q)));
        const std::size_t pos = static_cast<std::size_t>(rng.below(des.size() + 1));
        if(insertion) {
            std::string ins;
            for(std::size_t k = 0; k < len; ++k) ins.push_back(kNt[rng.below(4)]);
            des.insert(pos, ins);
        } else {
            des.erase(pos, len);  // clipped at the end of the sequence
        }
    }
    if(des.size() >= 3 && is_stop64(cod_int(std::string_view(des).substr(des.size() - 3)))) des.erase(des.size() - 3);
}

}  // namespace coati_amd
