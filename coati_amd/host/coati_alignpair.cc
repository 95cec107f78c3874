// coati-alignpair: the `coati alignpair` verb (src/coati-alignpair.cc:30-50) for the
// marginal models, with the DP on an MI355X.
#include <cstdlib>
#include <iostream>

#include "cli.hpp"

int main(int argc, char* argv[]) {
    using namespace coati_amd;
    args_t args;
    try {
        args = parse_arguments(verb_t::alignpair, argc, argv);
    } catch(const std::exception& e) {
        std::cerr << e.what() << "\nRun with --help for more information." << std::endl;
        return 106;  // CLI11's exit code for parse errors is non-zero; the exact value is not relied upon
    }
    if(args.help) {
        std::cout << usage(verb_t::alignpair);
        return EXIT_SUCCESS;
    }
    try {
        if(!args.aln.is_marginal()) {
            // the triplet/FST models (tri-mg, tri-ecm, dna) are a different algorithm and not served here
            throw std::invalid_argument("Mutation model unknown.");
        }
        const bool ok = args.batch ? marg_alignment_batch(args.aln) : marg_alignment(args.aln);
        return ok ? EXIT_SUCCESS : EXIT_FAILURE;
    } catch(const std::exception& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
}
