// `coati format`: convert between FASTA/PHYLIP/JSON, extract and/or reorder
// sequences, pad after insertions so that the reference keeps its reading phase.  Pure host work;
// never touches the GPU.
//
// Mirrors (same names, argument meaning, error messages):
//   format_t                          src/include/coati/structs.hpp:105-111
//   format_sequences, extract_seqs    src/lib/format.cc:41-128
//   set_options_format                src/lib/utils.cc:435-451
#ifndef COATI_AMD_HOST_FORMAT_HPP
#define COATI_AMD_HOST_FORMAT_HPP

#include <string>
#include <vector>

#include "seq.hpp"

namespace coati_amd {

struct format_t {
    bool preserve_phase{false};      // pad after insertions (gaps in the first sequence)
    std::string padding{"?"};        // padding character(s)
    std::vector<std::string> names;  // sequences to keep, by name, in this order
    std::vector<std::size_t> pos;    // ... or by 1-based position
};

// Keep only the sequences named by format.names or format.pos, in that order.
void extract_seqs(format_t& format, data_t& data);
// extract_seqs, then (preserve_phase) after every run of '-' in the first sequence whose length is
// not a multiple of 3 insert padding into ALL sequences so that the next codon starts in frame.
void format_data(format_t& format, data_t& data);
// format_data + write_output(data, output).  Returns EXIT_SUCCESS.
int format_sequences(format_t& format, data_t& data, const std::string& output);

struct format_args_t {
    format_t format;
    std::string input;
    std::string output;
    bool help{false};
};
format_args_t parse_arguments_format(int argc, const char* const* argv);
std::string usage_format();

}  // namespace coati_amd
#endif
