#include "insertions.hpp"

#include <algorithm>
#include <stdexcept>

namespace coati_amd {

namespace {
int flag_at(const insertion_data_t& d, std::size_t pos) { return pos < d.insertions.size() ? d.insertions[pos] : 0; }
}  // namespace

flag_vector insertion_flags(std::string_view ref, std::string_view seq) {
    if(ref.length() != seq.length())
        throw std::runtime_error("Opening insertion flags failed, length of sequences is different.");
    flag_vector flags(2 * seq.length(), 0);
    for(std::size_t i = 0; i < ref.length(); ++i)
        if(ref[i] == '-') flags[i] = kOpen;
    return flags;
}

// A column of gaps for every set that is NOT in `seq_indexes` (their flags shift right and the new
// column is closed); the sets in `seq_indexes` keep their residue there and the column is closed.
void add_gap(insertion_vector& ins_data, const std::vector<std::size_t>& seq_indexes, std::size_t pos) {
    for(std::size_t set = 0; set < ins_data.size(); ++set) {
        insertion_data_t& d = ins_data[set];
        const bool member = std::find(seq_indexes.begin(), seq_indexes.end(), set) != seq_indexes.end();
        if(member) {
            if(pos < d.insertions.size()) d.insertions[pos] = kClosed;
            continue;
        }
        for(std::string& row : d.sequences) row.insert(std::min(pos, row.size()), 1, '-');
        if(pos < d.insertions.size()) {
            d.insertions.insert(d.insertions.begin() + static_cast<std::ptrdiff_t>(pos), kClosed);
            d.insertions.pop_back();  // (the length stays 2 * len)
        }
    }
}

// Every closed insertion at `pos` (and the closed ones that follow it in the same set) gets its own
// column of gaps in all other sets.  Returns how many were handled.
uint64_t add_closed_ins(insertion_vector& ins_data, std::size_t pos) {
    uint64_t handled = 0;
    std::size_t set = 0;
    while(set < ins_data.size()) {
        if(flag_at(ins_data[set], pos) == kClosed) {
            add_gap(ins_data, {set}, pos);
            ++pos;  // the next column of the same set; later sets are looked at from there on
            ++handled;
        } else {
            ++set;
        }
    }
    return handled;
}

bool check_all_open(insertion_vector& ins_data, std::size_t pos) {
    char nuc = '0';
    for(const insertion_data_t& d : ins_data) {
        const std::string& row = d.sequences[0];
        if(pos > row.length()) return false;
        const char here = pos < row.length() ? row[pos] : '\0';
        if(nuc == '0') nuc = here;
        if(flag_at(d, pos) != kOpen || here != nuc) return false;
    }
    return true;
}

std::vector<std::size_t> find_open_ins(insertion_vector& ins_data, std::size_t pos) {
    std::vector<std::size_t> sets;
    char nuc = '0';
    for(std::size_t set = 0; set < ins_data.size(); ++set) {
        if(flag_at(ins_data[set], pos) != kOpen) continue;
        const std::string& row = ins_data[set].sequences[0];
        if(pos > row.length()) continue;
        const char here = pos < row.length() ? row[pos] : '\0';
        if(nuc == '0') {
            nuc = here;
            sets.push_back(set);
        } else if(here == nuc) {
            sets.push_back(set);
        }
    }
    return sets;
}

void merge_indels(insertion_vector& ins_data, insertion_data_t& merged_data) {
    if(ins_data.size() < 2) throw std::runtime_error("Merging indels of only 1 sequence.");
    uint64_t pending = 0, done = 0;
    for(const insertion_data_t& d : ins_data)
        pending += static_cast<uint64_t>(std::count_if(d.insertions.begin(), d.insertions.end(), [](int f) { return f != 0; }));
    for(std::size_t pos = 0; done < pending; ++pos) {
        // (i) closed insertions: a column of gaps in every other set
        done += add_closed_ins(ins_data, pos);
        // (ii) every set has an open insertion of the same residue here: it stays one open column
        if(check_all_open(ins_data, pos)) {
            done += ins_data.size();
            continue;
        }
        // (iii) the open insertions that share the first one's residue share a column, now closed
        const std::vector<std::size_t> sets = find_open_ins(ins_data, pos);
        if(!sets.empty()) {
            add_gap(ins_data, sets, pos);
            done += sets.size();
        }
    }
    for(const insertion_data_t& d : ins_data)
        for(std::size_t i = 0; i < d.sequences.size(); ++i) {
            merged_data.sequences.push_back(d.sequences[i]);
            merged_data.names.push_back(d.names[i]);
        }
    merged_data.insertions = ins_data[0].insertions;  // (identical in every set by now)
}

}  // namespace coati_amd
