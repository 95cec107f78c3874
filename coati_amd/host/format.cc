#include "format.hpp"

#include <algorithm>
#include <cstdlib>
#include <stdexcept>

#include "io.hpp"

namespace coati_amd {

void extract_seqs(format_t& format, data_t& data) {
    // names are translated to 1-based positions first (format.cc:94-107)
    for(const std::string& want : format.names) {
        const auto it = std::find(data.names.cbegin(), data.names.cend(), want);
        if(it == data.names.cend()) throw std::invalid_argument("Sequence " + want + " not found.");
        format.pos.push_back(static_cast<std::size_t>(it - data.names.cbegin()) + 1);
    }
    if(format.pos.empty()) return;
    for(const std::size_t p : format.pos)
        if(p == 0 || p > data.size()) throw std::invalid_argument("Positions of seqs to extract are of out range");
    std::vector<std::string> names, seqs;
    for(const std::size_t p : format.pos) {
        names.push_back(data.names[p - 1]);
        seqs.push_back(data.seqs[p - 1]);
    }
    data.names = std::move(names);
    data.seqs = std::move(seqs);
}

void format_data(format_t& format, data_t& data) {
    if(!format.names.empty() || !format.pos.empty()) extract_seqs(format, data);
    if(!format.preserve_phase) return;
    if(format.padding == "-") throw std::invalid_argument("Invalid padding character " + format.padding + " .");
    if(data.seqs.empty()) return;
    // format.cc:51-71.  A run of `len` gaps in the first sequence with len % 3 == 1 gets the first
    // padding character twice, len % 3 == 2 gets the first two padding characters once (with the
    // default one-character padding: 2 resp. 1 characters, so that run + padding is a multiple of 3).
    std::size_t pos = data.seqs[0].find('-');
    while(pos != std::string::npos) {
        std::size_t len = 0;
        while(pos < data.seqs[0].size() && data.seqs[0][pos] == '-') {
            ++pos;
            ++len;
        }
        len %= 3;
        if(len != 0) {
            const std::string piece = format.padding.substr(0, std::min(len, format.padding.size()));
            const std::string pad = len == 1 ? piece + piece : piece;
            for(std::string& s : data.seqs) s.insert(std::min(pos, s.size()), pad);
        }
        pos = data.seqs[0].find('-', pos);
    }
}

int format_sequences(format_t& format, data_t& data, const std::string& output) {
    format_data(format, data);
    write_output(data, output);
    return EXIT_SUCCESS;
}

std::string usage_format() {
    return "coati format - convert between formats, extract and/or reoder sequences\n"
           "Usage: coati-format [OPTIONS] input\n"
           "  input                       Input file (FASTA/PHYLIP/JSON accepted)\n"
           "  -o,--output TEXT            Alignment output file\n"
           "  -p,--preserve-phase         Preserve phase\n"
           "  -c,--padding TEXT           Padding char to format preserve phase (needs -p)\n"
           "  -s,--cut-seqs TEXT ...      Name of sequences to extract\n"
           "  -x,--cut-pos UINT ...       Position of sequences to extract (1 based; excludes -s)\n";
}

format_args_t parse_arguments_format(int argc, const char* const* argv) {
    format_args_t args;
    bool have_input = false, have_padding = false;
    auto need = [&](int& i, const std::string& flag) -> std::string {
        if(i + 1 >= argc) throw std::invalid_argument(flag + ": 1 required TEXT missing");
        return argv[++i];
    };
    auto is_value = [&](int i) { return i < argc && (argv[i][0] != '-' || argv[i][1] == '\0'); };
    for(int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if(a == "-h" || a == "--help") {
            args.help = true;
        } else if(a == "-o" || a == "--output") {
            args.output = need(i, a);
        } else if(a == "-p" || a == "--preserve-phase") {
            args.format.preserve_phase = true;
        } else if(a == "-c" || a == "--padding") {
            args.format.padding = need(i, a);
            have_padding = true;
        } else if(a == "-s" || a == "--cut-seqs") {
            if(!is_value(i + 1)) throw std::invalid_argument(a + ": At least 1 required");
            while(is_value(i + 1)) args.format.names.emplace_back(argv[++i]);
        } else if(a == "-x" || a == "--cut-pos") {
            if(!is_value(i + 1)) throw std::invalid_argument(a + ": At least 1 required");
            while(is_value(i + 1)) {
                const std::string v = argv[++i];
                char* end = nullptr;
                const long long n = std::strtoll(v.c_str(), &end, 10);
                if(end == v.c_str() || *end != '\0' || n < 0) throw std::invalid_argument(a + ": Value " + v + " could not be converted");
                args.format.pos.push_back(static_cast<std::size_t>(n));
            }
        } else if(!a.empty() && a[0] == '-' && a != "-") {
            throw std::invalid_argument("The following argument was not expected: " + a);
        } else if(!have_input) {
            args.input = a;
            have_input = true;
        } else {
            throw std::invalid_argument("The following argument was not expected: " + a);
        }
    }
    if(args.help) return args;
    if(!have_input) throw std::invalid_argument("input is required");
    if(have_padding && !args.format.preserve_phase) throw std::invalid_argument("--padding requires --preserve-phase");
    if(!args.format.names.empty() && !args.format.pos.empty()) throw std::invalid_argument("--cut-pos excludes --cut-seqs");
    return args;
}

}  // namespace coati_amd
