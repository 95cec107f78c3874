// coati-genseed: the `coati genseed` verb (src/coati-genseed.cc:27-52): seed the generator from
// the strings given (or from the machine's entropy when there are none) and print its state as four
// base-58 words (encode_seed, contrib/random/random.hpp:416-441) -- a string `coati sample -s`
// accepts.  Host only.
#include <cstdint>
#include <iostream>
#include <random>
#include <string>
#include <vector>

#include "random.hpp"

namespace {
std::string base58_word(uint32_t u) {
    static const char alphabet[] = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz";
    std::string word(6, alphabet[0]);
    for(int i = 0; i < 6 && u != 0; ++i, u /= 58) word[5 - i] = alphabet[u % 58];
    return word;
}
}  // namespace

int main(int argc, char* argv[]) {
    std::vector<std::string> seeds(argv + 1, argv + argc);
    if(seeds.empty()) {  // (the reference mixes several entropy sources; any unpredictable words do)
        std::random_device entropy;
        for(int i = 0; i < 4; ++i) seeds.push_back(std::to_string(static_cast<int32_t>(entropy())));
    }
    coati_amd::random_t rand;
    rand.seed(seeds);
    const uint32_t words[4] = {static_cast<uint32_t>(rand.lo()), static_cast<uint32_t>(rand.lo() >> 32),
                               static_cast<uint32_t>(rand.hi()), static_cast<uint32_t>(rand.hi() >> 32)};
    std::string out = base58_word(words[0]);
    for(int i = 1; i < 4; ++i) out += "-" + base58_word(words[i]);
    std::cout << out << std::endl;
    return 0;
}
