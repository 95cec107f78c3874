// coati-format: the `coati format` verb (src/coati-format.cc:27-52).  Host only.
#include <cstdlib>
#include <iostream>

#include "format.hpp"
#include "io.hpp"

int main(int argc, char* argv[]) {
    using namespace coati_amd;
    format_args_t args;
    try {
        args = parse_arguments_format(argc, argv);
    } catch(const std::exception& e) {
        std::cerr << e.what() << "\nRun with --help for more information." << std::endl;
        return 106;
    }
    if(args.help) {
        std::cout << usage_format();
        return EXIT_SUCCESS;
    }
    try {
        data_t data = read_input(args.input);
        return format_sequences(args.format, data, args.output);
    } catch(const std::exception& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
    }
    return EXIT_FAILURE;
}
