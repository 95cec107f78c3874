// Sequence preparation for the marginal aligner (host side).
//
// Mirrors:
//   data_t                         src/include/coati/data.hpp:44-77
//   marginal_seq_encoding          src/lib/utils.cc:496-528
//   order_ref, process_marginal    src/lib/utils.cc:789-838
//   trim_end_stops                 src/lib/utils.cc:945-967
//   restore_end_stops              src/lib/utils.cc:1044-1063
#ifndef COATI_AMD_HOST_SEQ_HPP
#define COATI_AMD_HOST_SEQ_HPP

#include <cstdint>
#include <string>
#include <string_view>
#include <vector>

#include "model.hpp"

namespace coati_amd {

struct data_t {
    std::string path;
    std::vector<std::string> names;
    std::vector<std::string> seqs;
    float score{0.f};
    std::vector<std::string> stops;  // trimmed terminal stop codons, one per sequence

    std::size_t size() const;  // throws if names/seqs disagree
};

using encoded_t = std::basic_string<unsigned char>;

// anc -> codon61*3+phase in [0,183); des -> nt16 code ('-' -> 15, any other
// character -> 16, exactly as upstream).  Throws std::invalid_argument for
// ambiguous nucleotides or a stop codon in the ancestor (upstream messages).
std::vector<encoded_t> marginal_seq_encoding(std::string_view anc, std::string_view des);
// the two halves of it, writing into caller storage (anc.size() resp. des.size() bytes); same errors
void encode_ancestor(std::string_view anc, unsigned char* out);
void encode_descendant(std::string_view des, unsigned char* out);

// Upstream lets descendant codes 15 ('-') and 16 (invalid) index past the 15
// table columns (only a debug assert, matrix.hpp:73-80).  The drivers here call
// this instead and fail with std::invalid_argument.
void check_descendant_codes(const encoded_t& des);

void trim_end_stops(data_t& data);
void restore_end_stops(data_t& data, const gap_t& gap);

// Put the reference first: `refs` is compared with the full sequence name.
void order_ref(data_t& data, const std::string& refs, bool rev);

// Exactly two sequences; optional reordering; length checks; trim terminal stops.
void process_marginal(data_t& data, const gap_t& gap, const std::string& refs, bool rev);

// ops (one byte per column: 0 M, 1 D, 2 I) -> the two gapped strings of traceback<>.
void ops_to_alignment(const uint8_t* ops, std::size_t n_ops, std::string_view anc, std::string_view des,
                      std::string& out_anc, std::string& out_des);

}  // namespace coati_amd
#endif
