#include "cli.hpp"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

namespace coati_amd {

namespace {
float positive_number(const std::string& flag, const std::string& v) {
    std::size_t used = 0;
    float x = 0.f;
    try {
        x = std::stof(v, &used);
    } catch(const std::exception&) {
        throw std::invalid_argument(flag + ": Value " + v + " could not be converted");
    }
    if(used != v.size()) throw std::invalid_argument(flag + ": Value " + v + " could not be converted");
    if(!(x > 0)) throw std::invalid_argument(flag + ": Number less or equal to 0: " + v);  // CLI::PositiveNumber
    return x;
}
float number(const std::string& flag, const std::string& v) {
    std::size_t used = 0;
    float x = 0.f;
    try {
        x = std::stof(v, &used);
    } catch(const std::exception&) {
        throw std::invalid_argument(flag + ": Value " + v + " could not be converted");
    }
    if(used != v.size()) throw std::invalid_argument(flag + ": Value " + v + " could not be converted");
    return x;
}
std::string upper(std::string s) {
    for(char& c : s) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
    return s;
}
}  // namespace

std::string usage(verb_t verb) {
    if(verb == verb_t::msa)
        return "coati msa - multiple sequence alignment of nucleotide sequences\n"
               "Usage: coati-msa [OPTIONS] input tree reference\n"
               "  input                       Input file (FASTA/PHYLIP/JSON accepted)\n"
               "  tree                        Newick phylogenetic tree\n"
               "  reference                   Name of reference sequence\n"
               "  -m,--model TEXT             Substitution model (mar-mg mar-ecm)\n"
               "  -o,--output TEXT            Alignment output file\n"
               "  -g,--gap-open FLOAT         Gap opening score\n"
               "  -e,--gap-extend FLOAT       Gap extension score\n"
               "  -w,--omega FLOAT            Nonsynonymous-synonymous bias\n"
               "  -p,--pi FLOAT x 4           Nucleotide frequencies (A C G T)\n"
               "  -k,--gap-len UINT           Gap unit length\n"
               "  -x,--sigma FLOAT x 6        GTR sigma parameters (AC AG AT CG CT GT)\n"
               "  -a,--ambiguous SUM|BEST     Ambiguous nucleotides model\n"
               "  --device INT                HIP device ordinal (default 0)\n";
    std::string u = verb == verb_t::alignpair ? "coati alignpair - pairwise alignment of nucleotide sequences\n"
                                                "Usage: coati-alignpair [OPTIONS] input\n"
                                              : "coati sample - align two sequences and sample alignments\n"
                                                "Usage: coati-sample [OPTIONS] input\n";
    u += "  input                       Input file (FASTA/PHYLIP/JSON accepted)\n"
         "  -m,--model TEXT             Substitution model (mar-mg mar-ecm)\n"
         "  --sub TEXT                  File with branch lengths and codon subst matrix\n"
         "  -t,--time FLOAT             Evolutionary time/branch length\n"
         "  -o,--output TEXT            Alignment output file\n"
         "  -g,--gap-open FLOAT         Gap opening score\n"
         "  -e,--gap-extend FLOAT       Gap extension score\n"
         "  -w,--omega FLOAT            Nonsynonymous-synonymous bias\n"
         "  -p,--pi FLOAT x 4           Nucleotide frequencies (A C G T)\n"
         "  -k,--gap-len UINT           Gap unit length\n"
         "  -x,--sigma FLOAT x 6        GTR sigma parameters (AC AG AT CG CT GT)\n"
         "  -a,--ambiguous SUM|BEST     Ambiguous nucleotides model\n"
         "  --device INT                HIP device ordinal (default 0)\n";
    if(verb == verb_t::alignpair)
        u += "  --marginal-sub SUM|MAX      Marginal substitution option\n"
             "  -r,--ref TEXT               Name of reference sequence (default: 1st seq)\n"
             "  -v,--rev-ref                Use 2nd seq as reference (default: 1st seq)\n"
             "  -s,--score                  Score input alignment and exit\n"
             "  --batch                     Input holds 2n sequences: align consecutive pairs (JSON array out)\n"
             "  --devices LIST              With --batch: one process per listed HIP device (e.g. 0,1,2,3), pairs sharded\n"
             "                              by DP cells, model broadcast and results gathered over RCCL\n";
    else
        u += "  -n,--sample-size UINT       Sample size\n"
             "  --independent-streams       Every sample from its own jumped-ahead RNG stream (all walks in parallel;\n"
             "                              sample 1 equals the default mode's, the rest are equivalent, not identical)\n"
             "  --fast-forward              Forward fill with the GPU's exp2/log2 instructions: log-weights within 1e-5 relative\n"
             "                              of the default's (which are the CPU reference's bits), 3.8x the fill rate\n"
             "  -s,--seed TEXT ...          Space separated list of seed(s) used for sampling\n";
    return u;
}

args_t parse_arguments(verb_t verb, int argc, const char* const* argv) {
    args_t args;
    alignment_t& aln = args.aln;
    bool have_input = false, have_model = false, have_ref = false, seeds_given = false;
    int msa_positionals = 0;  // msa: input, tree, reference
    auto need = [&](int& i, const std::string& flag) -> std::string {
        if(i + 1 >= argc) throw std::invalid_argument(flag + ": 1 required TEXT missing");
        return argv[++i];
    };
    for(int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if(a == "-h" || a == "--help") {
            args.help = true;
        } else if(a == "-m" || a == "--model") {
            aln.model = need(i, a);
            have_model = true;
        } else if(a == "--sub") {
            aln.rate = need(i, a);
        } else if(a == "-t" || a == "--time") {
            aln.br_len = positive_number(a, need(i, a));
        } else if(a == "-o" || a == "--output") {
            aln.output = need(i, a);
        } else if(a == "-g" || a == "--gap-open") {
            aln.gap.open = positive_number(a, need(i, a));
        } else if(a == "-e" || a == "--gap-extend") {
            aln.gap.extend = positive_number(a, need(i, a));
        } else if(a == "-w" || a == "--omega") {
            aln.omega = positive_number(a, need(i, a));
        } else if(a == "-p" || a == "--pi") {
            for(int q = 0; q < 4; ++q) aln.pi[q] = number(a, need(i, a));
        } else if(a == "-x" || a == "--sigma") {
            for(int q = 0; q < 6; ++q) aln.sigma[q] = number(a, need(i, a));
        } else if(a == "-k" || a == "--gap-len") {
            const std::string v = need(i, a);
            char* end = nullptr;
            const long long n = std::strtoll(v.c_str(), &end, 10);
            if(end == v.c_str() || *end != '\0') throw std::invalid_argument(a + ": Value " + v + " could not be converted");
            if(n < 1) throw std::invalid_argument(a + ": Gap unit length must be positive.");  // upstream: unchecked underflow
            aln.gap.len = static_cast<std::size_t>(n);
        } else if(a == "-a" || a == "--ambiguous") {
            const std::string v = upper(need(i, a));
            if(v == "SUM" || v == "0")
                aln.amb = AmbiguousNucs::SUM;
            else if(v == "BEST" || v == "1")
                aln.amb = AmbiguousNucs::BEST;
            else
                throw std::invalid_argument(a + ": Check " + v + " value in {SUM->0,BEST->1} OR {0,1} FAILED");
        } else if(a == "--device") {
            aln.device = std::atoi(need(i, a).c_str());
        } else if(a == "-b" || a == "--base-error") {
            (void)positive_number(a, need(i, a));  // only used by the FST models
        } else if(verb == verb_t::alignpair && a == "--marginal-sub") {
            const std::string v = upper(need(i, a));
            if(v == "SUM" || v == "0")
                aln.sub = MarginalSubst::SUM;
            else if(v == "MAX" || v == "1")
                aln.sub = MarginalSubst::MAX;
            else
                throw std::invalid_argument(a + ": Check " + v + " value in {SUM->0,MAX->1} OR {0,1} FAILED");
        } else if(verb == verb_t::alignpair && (a == "-r" || a == "--ref")) {
            aln.refs = need(i, a);
            have_ref = true;
        } else if(verb == verb_t::alignpair && (a == "-v" || a == "--rev-ref")) {
            aln.rev = true;
        } else if(verb == verb_t::alignpair && (a == "-s" || a == "--score")) {
            aln.score = true;
        } else if(verb == verb_t::alignpair && a == "--batch") {
            args.batch = true;
        } else if(verb == verb_t::alignpair && a == "--devices") {
            const std::string v = need(i, a);
            std::size_t at = 0;
            while(at <= v.size()) {
                const std::size_t comma = std::min(v.find(',', at), v.size());
                const std::string tok = v.substr(at, comma - at);
                char* end = nullptr;
                const long d = std::strtol(tok.c_str(), &end, 10);
                if(tok.empty() || *end != '\0' || d < 0) throw std::invalid_argument(a + ": Value " + v + " could not be converted");
                args.devices.push_back(static_cast<int>(d));
                at = comma + 1;
            }
        } else if(verb == verb_t::alignpair && a == "--dist-rank") {
            args.dist_rank = std::atoi(need(i, a).c_str());
        } else if(verb == verb_t::alignpair && a == "--dist-world") {
            args.dist_world = std::atoi(need(i, a).c_str());
        } else if(verb == verb_t::alignpair && a == "--dist-id") {
            args.dist_id = need(i, a);
        } else if(verb == verb_t::sample && a == "--independent-streams") {
            args.aln.independent_streams = true;
        } else if(verb == verb_t::sample && a == "--fast-forward") {
            args.aln.fast_forward = true;
        } else if(verb == verb_t::sample && (a == "-n" || a == "--sample-size")) {
            const std::string v = need(i, a);
            char* end = nullptr;
            const long long n = std::strtoll(v.c_str(), &end, 10);
            if(end == v.c_str() || *end != '\0' || n < 0) throw std::invalid_argument(a + ": Value " + v + " could not be converted");
            args.sample_size = static_cast<std::size_t>(n);
        } else if(verb == verb_t::sample && (a == "-s" || a == "--seed")) {
            if(!seeds_given) args.seeds.clear();
            seeds_given = true;
            while(i + 1 < argc && (argv[i + 1][0] != '-' || std::isdigit(static_cast<unsigned char>(argv[i + 1][1])))) args.seeds.emplace_back(argv[++i]);
            if(args.seeds.empty()) throw std::invalid_argument(a + ": At least 1 required");
        } else if(!a.empty() && a[0] == '-' && a != "-") {
            throw std::invalid_argument("The following argument was not expected: " + a);
        } else if(!have_input) {
            aln.data.path = a;
            have_input = true;
            msa_positionals = 1;
        } else if(verb == verb_t::msa && msa_positionals == 1) {
            aln.tree = a;
            msa_positionals = 2;
        } else if(verb == verb_t::msa && msa_positionals == 2) {
            aln.refs = a;
            msa_positionals = 3;
        } else {
            throw std::invalid_argument("The following argument was not expected: " + a);
        }
    }
    if(args.help) return args;
    if(!have_input) throw std::invalid_argument("input is required");
    if(verb == verb_t::msa) {
        if(msa_positionals < 2) throw std::invalid_argument("tree is required");
        if(msa_positionals < 3) throw std::invalid_argument("reference is required");
        if(FILE* f = std::fopen(aln.tree.c_str(), "r"))
            std::fclose(f);
        else
            throw std::invalid_argument("tree: File does not exist: " + aln.tree);  // CLI::ExistingFile
    }
    if(!args.devices.empty() && !args.batch) throw std::invalid_argument("--devices requires --batch");
    if(have_model && !aln.rate.empty()) throw std::invalid_argument("--sub excludes --model");
    if(have_ref && aln.rev) throw std::invalid_argument("--rev-ref excludes --ref");
    return args;
}

}  // namespace coati_amd
