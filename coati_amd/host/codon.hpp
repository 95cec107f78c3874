// Codon / nucleotide bookkeeping for the marginal models.
//
// Restates (from the published definitions, not from the reference's text):
//   nt16_table          src/include/coati/utils.hpp:54-61
//   amino_group         src/include/coati/utils.hpp:66-70
//   cod_int             src/lib/utils.cc:72-85
//   cod64_to_61 / _64   src/lib/utils.cc:1144-1165,1195-1211
//   get_nuc             src/lib/utils.cc:738-749
#ifndef COATI_AMD_HOST_CODON_HPP
#define COATI_AMD_HOST_CODON_HPP

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <string_view>

namespace coati_amd {

// IUPAC nucleotide codes in table-column order; '-' is 15, anything else 16.
inline constexpr std::string_view kNt16 = "ACGTRYMKSWBDHVN-";

// ASCII -> nt16 code (A C G T/U R Y M K S W B D H V N '-' -> 0..15, other -> 16).
inline uint8_t nt16(unsigned char ch) {
    static const std::array<uint8_t, 256> lut = [] {
        std::array<uint8_t, 256> t{};
        t.fill(16);
        for(size_t k = 0; k < kNt16.size(); ++k) {
            const unsigned char up = static_cast<unsigned char>(kNt16[k]);
            t[up] = static_cast<uint8_t>(k);
            if(up >= 'A' && up <= 'Z') t[up - 'A' + 'a'] = static_cast<uint8_t>(k);
        }
        t['U'] = t['u'] = 3;
        return t;
    }();
    return lut[ch];
}

inline bool is_stop64(int cod) { return cod == 48 || cod == 50 || cod == 56; }  // TAA TAG TGA

// Position of a codon in AAA..TTT order (0..63); -1 if any of the three
// characters is not one of ACGTU (either case).
inline int cod_int(std::string_view codon) {
    if(codon.size() < 3) return -1;
    int v = 0;
    for(int k = 0; k < 3; ++k) {
        const uint8_t c = nt16(static_cast<unsigned char>(codon[k]));
        if(c > 3) return -1;
        v = (v << 2) | c;
    }
    return v;
}

// 64-codon index -> 61 sense-codon index.
inline int cod64_to_61(int cod) {
    if(cod < 0 || cod > 63) throw std::out_of_range("Codon index " + std::to_string(cod) + " is out of range [0-63].");
    if(is_stop64(cod)) throw std::invalid_argument("Stop codon not expected in cod64_to_61");
    return cod - (cod > 48) - (cod > 50) - (cod > 56);
}

// 61 sense-codon index -> 64-codon index.
inline int cod61_to_64(int cod) {
    if(cod < 0 || cod > 60) throw std::out_of_range("Codon index " + std::to_string(cod) + " is out of range [0-60].");
    int c = cod;
    if(c >= 48) ++c;  // skip TAA
    if(c >= 50) ++c;  // skip TAG
    if(c >= 56) ++c;  // skip TGA
    return c;
}

// Nucleotide (0..3) at position pos (0..2) of sense codon `cod`.
inline uint8_t get_nuc(int cod, int pos) {
    const int c = cod61_to_64(cod);
    return static_cast<uint8_t>((c >> (4 - 2 * pos)) & 3);
}

// Amino acid (one-letter, standard genetic code) of each of the 61 sense codons;
// two codons are "in the same group" iff they are synonymous.
inline char amino_acid61(int cod) {
    static constexpr std::string_view code64 =
        "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";
    return code64[static_cast<size_t>(cod61_to_64(cod))];
}

}  // namespace coati_amd
#endif
