// Deterministic synthetic workload of BASELINE.json configs 2/4/5 (SURVEY.md
// §8(d)): pair p is a function of (seed_base + p) only, so any rank can
// generate any shard.  Not part of the reference; a bench/test utility.
#ifndef COATI_AMD_HOST_SYNTH_HPP
#define COATI_AMD_HOST_SYNTH_HPP

#include <cstdint>
#include <string>

namespace coati_amd {

struct synth_params_t {
    uint64_t seed_base{0xC0A71};
    uint32_t n_codons{334};      // ancestor length in codons (1 002 nt)
    double sub_rate{0.05};       // per-site substitution probability
    double indel_lambda{2.0};    // Poisson mean of indel events per pair
    double indel_mean_len{6.0};  // geometric mean length of one indel
};

// ancestor: n_codons sense codons, uniform.  descendant: copy, per-site
// substitutions to a uniformly chosen other base, then Poisson(indel_lambda)
// indel events at uniform positions (insertion/deletion with probability 1/2,
// geometric length), then a terminal stop codon, if one was created, removed.
void synth_pair(uint64_t index, const synth_params_t& prm, std::string& anc, std::string& des);

}  // namespace coati_amd
#endif
