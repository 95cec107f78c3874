// Merging pairwise (reference, leaf) alignments into one MSA: bookkeeping of insertion columns.
// Host only.
//
// Mirrors (same names, argument meaning, error messages):
//   insertion_data_t, insertion_vector                       src/include/coati/insertions.hpp:39-58
//   insertion_flags, merge_indels, add_closed_ins,
//   check_all_open, find_open_ins, add_gap                   src/lib/insertions.cc:38-438
// The reference keeps the per-column flags in an Eigen sparse vector of length 2*len; here they are a
// plain vector<int> of the same length (0 = no insertion, kOpen, kClosed).
#ifndef COATI_AMD_HOST_INSERTIONS_HPP
#define COATI_AMD_HOST_INSERTIONS_HPP

#include <cstdint>
#include <string>
#include <string_view>
#include <vector>

namespace coati_amd {

constexpr int kOpen = 111;    // 'o': an insertion (w.r.t. the reference) other sequences may still share
constexpr int kClosed = 99;   // 'c': a column that no further insertion may join

using flag_vector = std::vector<int>;

struct insertion_data_t {
    std::vector<std::string> sequences;  // rows aligned so far (same length)
    std::vector<std::string> names;
    flag_vector insertions;              // per column; length 2 * len so that merging never overflows

    insertion_data_t() = default;
    insertion_data_t(const std::string& s, const std::string& n, flag_vector f)
        : sequences(1, s), names(1, n), insertions{std::move(f)} {}
    insertion_data_t(std::vector<std::string> s, std::vector<std::string> n, flag_vector f)
        : sequences{std::move(s)}, names{std::move(n)}, insertions{std::move(f)} {}
};
using insertion_vector = std::vector<insertion_data_t>;

// kOpen at every column where `ref` has a gap (ref and seq: one pairwise alignment).
flag_vector insertion_flags(std::string_view ref, std::string_view seq);
// Merge the children of one tree node, column by column, into `merged_data`.
void merge_indels(insertion_vector& ins_data, insertion_data_t& merged_data);
uint64_t add_closed_ins(insertion_vector& ins_data, std::size_t pos);
bool check_all_open(insertion_vector& ins_data, std::size_t pos);
std::vector<std::size_t> find_open_ins(insertion_vector& ins_data, std::size_t pos);
void add_gap(insertion_vector& ins_data, const std::vector<std::size_t>& seq_indexes, std::size_t pos);

}  // namespace coati_amd
#endif
