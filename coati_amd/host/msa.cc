// `coati msa`: progressive merge of reference-anchored pairwise alignments.
// The pairwise step runs on the GPU in one batch (align.cc: align_leafs); everything here is host
// bookkeeping.  Mirrors ref_indel_alignment / merge_alignments of src/lib/align_msa.cc:45-120,337-374.
#include <algorithm>
#include <stdexcept>

#include "align.hpp"
#include "insertions.hpp"
#include "io.hpp"
#include "tree.hpp"

namespace coati_amd {

namespace {

// Inner nodes are merged as soon as all their children are done; a node with one child passes that
// child's block up unchanged.
void merge_alignments(std::vector<bool>& visited, const tree::tree_t& tree, insertion_vector& nodes_ins,
                      const std::vector<std::size_t>& inodes) {
    while(std::any_of(visited.begin(), visited.end(), [](bool done) { return !done; })) {
        bool progressed = false;
        for(const std::size_t n : inodes) {
            if(visited[n]) continue;
            const auto& kids = tree[n].children;
            if(std::any_of(kids.begin(), kids.end(), [&](std::size_t c) { return !visited[c]; })) continue;
            visited[n] = true;
            progressed = true;
            if(kids.size() == 1) {
                nodes_ins[n] = nodes_ins[kids[0]];
                continue;
            }
            insertion_vector blocks;
            blocks.reserve(kids.size());
            for(const std::size_t c : kids) blocks.push_back(nodes_ins[c]);
            nodes_ins[n] = insertion_data_t();
            merge_indels(blocks, nodes_ins[n]);
        }
        if(!progressed) throw std::runtime_error("Guide tree cannot be merged (inner node without children).");
    }
}

}  // namespace

bool ref_indel_alignment(alignment_t& input) {
    if(!input.is_marginal()) throw std::invalid_argument("MSA only supports marginal models.");
    input.data = read_input(input.data.path);
    if(input.data.size() < 3) throw std::invalid_argument("At least three sequences required.");

    std::string newick = tree::read_newick(input.tree);
    tree::tree_t tree = tree::parse_newick(newick);
    tree::reroot(tree, input.refs);
    const std::size_t ref_pos = tree::find_node(tree, input.refs);
    const std::string ref_seq = tree::find_seq(input.refs, input.data);

    insertion_vector nodes_ins(tree.size());
    nodes_ins[ref_pos] = insertion_data_t(ref_seq, input.refs, flag_vector(2 * ref_seq.length(), 0));

    // pairwise step (align_leafs, align_msa.cc:285-318): all leaves but the reference, each with the
    // substitution table of its own distance to the reference -- one batched launch
    std::vector<std::size_t> leaves;
    std::vector<std::string> leaf_seqs;
    std::vector<float> br_lens;
    for(std::size_t n = 0; n < tree.size(); ++n) {
        if(!tree[n].is_leaf || tree[n].label == input.refs) continue;
        leaves.push_back(n);
        br_lens.push_back(tree::distance_ref(tree, ref_pos, n));
        leaf_seqs.push_back(tree::find_seq(tree[n].label, input.data));
    }
    const std::vector<data_t> pairwise = align_leafs(input, ref_seq, leaf_seqs, br_lens);
    for(std::size_t q = 0; q < leaves.size(); ++q)
        nodes_ins[leaves[q]] = insertion_data_t(pairwise[q].seqs[1], tree[leaves[q]].label,
                                                insertion_flags(pairwise[q].seqs[0], pairwise[q].seqs[1]));

    std::vector<std::size_t> inodes;
    std::vector<bool> visited(tree.size(), false);
    for(std::size_t n = 0; n < tree.size(); ++n) {
        if(tree[n].is_leaf)
            visited[n] = true;
        else
            inodes.push_back(n);
    }
    for(std::size_t n = 0; n < tree.size(); ++n)
        if(tree[n].parent != n) tree[tree[n].parent].children.push_back(n);

    merge_alignments(visited, tree, nodes_ins, inodes);

    // rows in the order of the input file
    const insertion_data_t& root = nodes_ins[tree[ref_pos].parent];
    data_t out;
    for(const std::string& name : input.data.names) {
        const auto it = std::find(root.names.begin(), root.names.end(), name);
        if(it == root.names.end()) throw std::invalid_argument("Sequence " + name + " is not a leaf of the tree.");
        out.names.push_back(name);
        out.seqs.push_back(root.sequences[static_cast<std::size_t>(it - root.names.begin())]);
    }
    write_output(out, input.output);
    return true;
}

}  // namespace coati_amd
