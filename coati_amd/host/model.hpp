// Marginal substitution models: codon P matrices and the 183x15 log-odds table
// the DP kernels consume.  Host-side, computed once per model.
//
// Mirrors (same names, argument meaning and errors):
//   mg94_p, gtr_q, marginal_p, ambiguous_sum_p, ambiguous_best_p
//                         src/lib/mutation_coati.cc:49-125,164-306,317-354
//   ecm_p                 src/lib/mutation_ecm.cc:151-184
//   set_subst             src/lib/utils.cc:595-620 (marginal branches)
#ifndef COATI_AMD_HOST_MODEL_HPP
#define COATI_AMD_HOST_MODEL_HPP

#include <array>
#include <cstddef>
#include <string>
#include <vector>

namespace coati_amd {

enum class AmbiguousNucs { SUM, BEST };   // structs.hpp:62
enum class MarginalSubst { SUM, MAX };    // structs.hpp:63

// aln.gap (structs.hpp:37-50)
struct gap_t {
    std::size_t len{1};
    float open{0.001};                     // NOLINT: double literal narrowed, as upstream
    float extend{1.0f - 1.0f / 6.0f};
};

// {no_gap, gap_stop, gap_open, gap_extend} = {log1pf(-g), log1pf(-e), logf(g), logf(e)}
// (align_pair.cc:66-69, semiring.hpp:117-120).  Throws std::invalid_argument
// unless 0 < g < 1 and 0 < e < 1 (upstream does not validate; the logs would
// be NaN or infinite).
std::array<float, 4> gap_log_consts(const gap_t& gap);

using matrix61_t = std::vector<float>;  // 61*61 row-major
using table_t = std::vector<float>;     // 183*15 row-major

constexpr std::size_t kTableRows = 183, kTableCols = 15;

// nucleotide GTR rate matrix (4x4 row-major)
std::array<float, 16> gtr_q(const std::array<float, 4>& pi, const std::array<float, 6>& sigma);

// MG94 codon substitution probabilities for branch length br_len.
matrix61_t mg94_p(float br_len, float omega, const std::array<float, 4>& nuc_freqs,
                  const std::array<float, 6>& sigma = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f});

// Empirical codon model (Kosiol et al. 2007) substitution probabilities.
matrix61_t ecm_p(float br_len, float omega);

// exp(Q * t) for a user-supplied 61x61 rate matrix (io.cc:48-88 `--sub`).
matrix61_t rate_matrix_p(const matrix61_t& Q, float br_len);

// 61x61 P -> 183x15 table: log(P(nuc | codon, phase) / pi[nuc]) for A,C,G,T and
// the 11 IUPAC ambiguity columns.
table_t marginal_p(const matrix61_t& P, const std::array<float, 4>& pi, AmbiguousNucs amb,
                   MarginalSubst msub);

// aln.{model, br_len, omega, pi, sigma, amb, sub} (structs.hpp:69-97)
struct model_params_t {
    std::string model{"mar-mg"};
    float br_len{0.0133};                  // NOLINT
    float omega{0.2};                      // NOLINT
    std::array<float, 4> pi{0.308, 0.185, 0.199, 0.308};  // NOLINT
    std::array<float, 6> sigma{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    AmbiguousNucs amb{AmbiguousNucs::SUM};
    MarginalSubst sub{MarginalSubst::SUM};
};

// set_subst for the marginal models: "mar-mg" or "mar-ecm"; anything else throws
// std::invalid_argument("Mutation model unknown.") -- the triplet/FST models
// are not served by this library.
table_t set_subst(const model_params_t& params);

// 61x61 matrix exponential used by the models (exposed for tests).
matrix61_t expm61(const matrix61_t& A);

}  // namespace coati_amd
#endif
