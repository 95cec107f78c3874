// coati-msa: the `coati msa` verb (src/coati-msa.cc:27-44) for the marginal models; the pairwise
// leaf alignments run on an MI355X in one batch.
#include <cstdlib>
#include <iostream>

#include "cli.hpp"

int main(int argc, char* argv[]) {
    using namespace coati_amd;
    args_t args;
    try {
        args = parse_arguments(verb_t::msa, argc, argv);
    } catch(const std::exception& e) {
        std::cerr << e.what() << "\nRun with --help for more information." << std::endl;
        return 106;
    }
    if(args.help) {
        std::cout << usage(verb_t::msa);
        return EXIT_SUCCESS;
    }
    try {
        return ref_indel_alignment(args.aln) ? EXIT_SUCCESS : EXIT_FAILURE;
    } catch(const std::exception& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
    }
    return EXIT_FAILURE;
}
