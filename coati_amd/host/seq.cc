#include "seq.hpp"

#include <cmath>
#include <stdexcept>

#include "codon.hpp"

namespace coati_amd {

std::size_t data_t::size() const {
    if(names.size() != seqs.size()) throw std::invalid_argument("Different number of sequences and names.");
    return names.size();
}

void encode_ancestor(std::string_view anc, unsigned char* out) {
    for(std::size_t i = 0; i < anc.size(); i += 3) {
        const int cod = cod_int(anc.substr(i, 3));  // (-1 also for a trailing partial codon)
        if(cod == -1) throw std::invalid_argument("Ambiguous nucleotides in ancestor/reference.");
        if(is_stop64(cod)) throw std::invalid_argument("Early stop codon in ancestor/reference.");
        const int base = cod64_to_61(cod) * 3;
        for(int phase = 0; phase < 3; ++phase) out[i + static_cast<std::size_t>(phase)] = static_cast<unsigned char>(base + phase);
    }
}

void encode_descendant(std::string_view des, unsigned char* out) {
    for(std::size_t i = 0; i < des.size(); ++i) out[i] = nt16(static_cast<unsigned char>(des[i]));
}

std::vector<encoded_t> marginal_seq_encoding(std::string_view anc, std::string_view des) {
    std::vector<encoded_t> ret(2);
    ret[0].resize(anc.size());
    ret[1].resize(des.size());
    encode_ancestor(anc, ret[0].data());
    encode_descendant(des, ret[1].data());
    return ret;
}

void check_descendant_codes(const encoded_t& des) {
    for(const unsigned char code : des)
        if(code >= kTableCols) throw std::invalid_argument("Invalid character in descendant sequence.");
}

void trim_end_stops(data_t& data) {
    for(std::size_t i = 0; i < data.size(); ++i) {
        std::string& seq = data.seqs[i];
        const std::size_t len = seq.size();
        if(len >= 3 && is_stop64(cod_int(std::string_view(seq).substr(len - 3)))) {
            data.stops.emplace_back(seq.substr(len - 3));
            seq.erase(len - 3);
        } else {
            data.stops.emplace_back("");
        }
    }
}

void restore_end_stops(data_t& data, const gap_t& gap) {
    if(data.stops.size() != 2) throw std::runtime_error("Error restoring end stop codons.");
    const float gap_score = ::logf(gap.open * gap.extend * gap.extend);
    if(data.stops[0].size() == data.stops[1].size()) {  // both or neither had a stop
        data.seqs[0].append(data.stops[0]);
        data.seqs[1].append(data.stops[1]);
    } else if(data.stops[0].empty()) {  // only the descendant
        data.seqs[0].append("---");
        data.seqs[1].append(data.stops[1]);
        data.score += gap_score;
    } else if(data.stops[1].empty()) {  // only the ancestor
        data.seqs[0].append(data.stops[0]);
        data.seqs[1].append("---");
        data.score += gap_score;
    }
}

void order_ref(data_t& data, const std::string& refs, bool rev) {
    if(data.names[0] == refs) return;
    if(data.names[1] == refs || rev) {
        std::swap(data.names[0], data.names[1]);
        std::swap(data.seqs[0], data.seqs[1]);
        return;
    }
    throw std::invalid_argument("Name of reference sequence not found.");
}

void process_marginal(data_t& data, const gap_t& gap, const std::string& refs, bool rev) {
    if(data.size() != 2) throw std::invalid_argument("Exactly two sequences required.");
    if(!refs.empty() || rev) order_ref(data, refs, rev);
    const std::size_t len_a = data.seqs[0].size(), len_b = data.seqs[1].size();
    if(len_a % 3 != 0 || len_a % gap.len != 0)
        throw std::invalid_argument("Length of reference sequence must be multiple of 3 and gap unit length.");
    if(len_b % gap.len != 0)
        throw std::invalid_argument("Length of descendant sequence must be multiple of gap unit length.");
    trim_end_stops(data);
}

void ops_to_alignment(const uint8_t* ops, std::size_t n_ops, std::string_view anc, std::string_view des,
                      std::string& out_anc, std::string& out_des) {
    out_anc.clear();
    out_des.clear();
    out_anc.reserve(n_ops);
    out_des.reserve(n_ops);
    std::size_t pa = 0, pb = 0;
    for(std::size_t t = 0; t < n_ops; ++t) {
        if(ops[t] == 2) {
            if(pb >= des.size()) throw std::runtime_error("alignment ops overrun the descendant");
            out_anc.push_back('-');
            out_des.push_back(des[pb++]);
        } else if(ops[t] == 1) {
            if(pa >= anc.size()) throw std::runtime_error("alignment ops overrun the ancestor");
            out_anc.push_back(anc[pa++]);
            out_des.push_back('-');
        } else {
            if(pa >= anc.size() || pb >= des.size()) throw std::runtime_error("alignment ops overrun the sequences");
            out_anc.push_back(anc[pa++]);
            out_des.push_back(des[pb++]);
        }
    }
    if(pa != anc.size() || pb != des.size()) throw std::runtime_error("alignment ops do not consume the sequences");
}

}  // namespace coati_amd
