// coati-sample: the `coati sample` verb (src/coati-sample.cc:30-61).
#include <cstdlib>
#include <iostream>

#include "cli.hpp"

int main(int argc, char* argv[]) {
    using namespace coati_amd;
    args_t args;
    try {
        args = parse_arguments(verb_t::sample, argc, argv);
    } catch(const std::exception& e) {
        std::cerr << e.what() << "\nRun with --help for more information." << std::endl;
        return 106;
    }
    if(args.help) {
        std::cout << usage(verb_t::sample);
        return EXIT_SUCCESS;
    }
    if(!args.aln.is_marginal()) {
        std::cerr << "ERROR: Sampling only available with models mar-mg or mar-ecm." << std::endl;
        return EXIT_FAILURE;
    }
    // args.seeds defaults to {""} so, as upstream (structs.hpp:120, coati-sample.cc:46-50), the
    // string seeding is always taken: without -s every run uses the hash of the empty string.
    random_t rand;
    rand.seed(args.seeds);
    try {
        marg_sample(args.aln, args.sample_size, rand);
    } catch(const std::exception& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
