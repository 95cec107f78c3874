// The marginal workflow drivers and the pairwise-DP API they call, with the DP
// itself executed by libcoati_hip.so (include/coati_hip.h) on an MI355X.
//
// Mirrors (same names, argument meaning, error behaviour):
//   alignment_t                                   src/include/coati/structs.hpp:69-97
//   viterbi_mem, traceback_viterbi, forward,
//   sampleback                                     src/include/coati/align_pair.hpp:157-182
//   marg_alignment, alignment_score, marg_sample   src/include/coati/align_marginal.hpp:32-36
//                                                  src/lib/align_marginal.cc:44-88,373-473,536-594
// Extension (not in the reference, which aligns one pair per process):
//   marg_alignment_batch -- many pairs, one model, one GPU launch.
#ifndef COATI_AMD_HOST_ALIGN_HPP
#define COATI_AMD_HOST_ALIGN_HPP

#include <cstdint>
#include <iosfwd>
#include <string>
#include <vector>

#include "model.hpp"
#include "random.hpp"
#include "seq.hpp"

struct coati_hip_model;
struct coati_hip_batch;

namespace coati_amd {

class alignment_t {
   public:
    data_t data;
    std::string model{"mar-mg"};
    float br_len{0.0133};  // NOLINT
    float omega{0.2};      // NOLINT
    std::array<float, 4> pi{0.308, 0.185, 0.199, 0.308};  // NOLINT
    std::string refs;
    std::string tree;  // msa: path of the Newick guide tree
    bool rev{false};
    std::string rate;  // --sub: CSV rate matrix
    gap_t gap;
    std::array<float, 6> sigma{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    table_t subst_matrix;  // 183x15
    std::string output;
    bool score{false};
    AmbiguousNucs amb{AmbiguousNucs::SUM};
    MarginalSubst sub{MarginalSubst::SUM};
    int device{0};  // HIP device ordinal (not in the reference)
    // coati-sample --independent-streams (not in the reference): every sample draws from its own
    // jumped-ahead Lehmer stream; all walks run in parallel, sample 0 equals the reference's first
    bool independent_streams{false};
    // coati-sample --fast-forward: the Forward fill in the C ABI's tolerance mode (COATI_HIP_OPT_FORWARD_MODE: hardware exp2 / log2,
    // log-weights within 1e-5 relative of the reference's instead of bit-identical; 3.8x the fill rate)
    bool fast_forward{false};

    bool is_marginal() const { return model == "mar-mg" || model == "mar-ecm" || !rate.empty(); }
    std::string& seq(std::size_t i) { return data.seqs[i]; }
    std::string& name(std::size_t i) { return data.names[i]; }
};

// utils::set_subst (src/lib/utils.cc:595-620), marginal branches only
void set_subst(alignment_t& aln);

using seq_view_t = std::basic_string_view<unsigned char>;

// Viterbi work "matrices": in the reference three dense fp32 matrices; here the handle of
// the HBM-resident batch of one pair plus its results.
class align_pair_work_mem_t {
   public:
    align_pair_work_mem_t() = default;
    ~align_pair_work_mem_t();
    align_pair_work_mem_t(const align_pair_work_mem_t&) = delete;
    align_pair_work_mem_t& operator=(const align_pair_work_mem_t&) = delete;
    std::vector<uint8_t> ops;  // one op per alignment column
    float score{0.f};
    coati_hip_model* model{nullptr};
    coati_hip_batch* batch{nullptr};
};
// Forward work matrices (M/D/I resident in HBM until destruction).
class align_pair_work_t : public align_pair_work_mem_t {};

void viterbi_mem(align_pair_work_mem_t& work, const seq_view_t& a, const seq_view_t& b, const alignment_t& aln);
void traceback_viterbi(const align_pair_work_mem_t& work, const std::string& a, const std::string& b,
                       alignment_t& aln, std::size_t look_back);
void forward(align_pair_work_t& work, const seq_view_t& a, const seq_view_t& b, const alignment_t& aln);
void sampleback(const align_pair_work_t& work, const std::string& a, const std::string& b, alignment_t& aln,
                std::size_t look_back, random_t& rand);

bool marg_alignment(alignment_t& aln);
float alignment_score(alignment_t& aln, const table_t& p_marg);
void marg_sample(alignment_t& aln, std::size_t sample_size, random_t& rand);

// Batch extension: the input holds 2n sequences, consecutive ones form a pair
// (reference first unless rev); writes a JSON array of n alignments.
bool marg_alignment_batch(alignment_t& aln);
// A one-shot tool that exits right after the call says so: the model's cached HBM workspaces (several GB) are then
// left to the driver instead of being freed one by one (~65 ms of a 0.3 s run).  Off by default (libraries clean up).
void set_process_exits_after_call(bool on);
long batch_reader_first_difference(const std::string& path);  // (test hook, align.cc)
long batch_shard_first_difference(const alignment_t& aln, int world, int rank, uint64_t* s0, uint64_t* s1);  // (test hook)
// The same over several GPUs, ONE PROCESS PER GPU (this process is rank `rank` of `world` and drives
// aln.device): every rank reads the input, rank 0 computes the model and broadcasts it (ncclBroadcast),
// each rank aligns its shard of coati_hip_shard_bounds, the results are gathered to rank 0 over RCCL
// and rank 0 writes the output (include/coati_hip_dist.h; libcoati_hip_dist.so is loaded on demand).
// id_file: where rank 0 leaves the rendezvous id for the others.
bool marg_alignment_batch_dist(alignment_t& aln, int rank, int world, const std::string& id_file);

// The pairwise step of `coati msa` (align_leafs, src/lib/align_msa.cc:285-318) for all leaves at
// once: every leaf is aligned to `ref_seq` with the substitution table of ITS branch length
// (`input` supplies model, omega, pi, sigma, gaps; its br_len is overwritten per leaf, as the
// reference does).  One multi-table model, one launch.  Returns, per leaf, the aligned pair
// (seqs[0] = reference row, seqs[1] = leaf row) and the score.  Sequences are taken as they are
// (the reference's loop does not trim stop codons either).
std::vector<data_t> align_leafs(alignment_t& input, const std::string& ref_seq, const std::vector<std::string>& leaf_seqs,
                                const std::vector<float>& br_lens);

// `coati msa` (ref_indel_alignment, src/lib/align_msa.cc:45-120): every leaf of the guide tree is
// aligned to the reference (one batched GPU launch, align_leafs), then the pairwise alignments are
// merged up the tree re-rooted at the reference's parent (merge_alignments, align_msa.cc:337-374;
// insertions.hpp).  Reads input.data.path and input.tree, writes input.output.
bool ref_indel_alignment(alignment_t& input);

}  // namespace coati_amd
#endif
