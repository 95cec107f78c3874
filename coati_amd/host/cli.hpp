// Command-line surface of `coati alignpair` / `coati sample` / `coati msa`
// (set_options_alignpair / set_options_sample / set_options_msa, src/lib/utils.cc:93-161,224-268,328-380):
// same flags, defaults and value checks, parsed by a small hand-written parser
// (the reference uses the vendored CLI11).
#ifndef COATI_AMD_HOST_CLI_HPP
#define COATI_AMD_HOST_CLI_HPP

#include <string>
#include <vector>

#include "align.hpp"

namespace coati_amd {

struct args_t {
    alignment_t aln;
    std::size_t sample_size{1};
    std::vector<std::string> seeds{{""}};  // structs.hpp:120: the default is one empty seed string
    bool batch{false};                    // extension: consecutive sequence pairs
    std::vector<int> devices;             // extension: --devices 0,1,... = one process per listed GPU (--batch only)
    int dist_rank{-1}, dist_world{0};     // (internal: what the --devices launcher passes to its children)
    std::string dist_id;                  // (internal: rendezvous file)
    bool help{false};
};

enum class verb_t { alignpair, sample, msa };

// Throws std::invalid_argument with a CLI-style message on bad usage.
args_t parse_arguments(verb_t verb, int argc, const char* const* argv);
std::string usage(verb_t verb);

}  // namespace coati_amd
#endif
