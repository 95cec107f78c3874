// Guide tree of `coati msa`: Newick text -> node vector, re-rooting at the
// reference's parent, distances.  Host only.
//
// Mirrors (same names, argument meaning, error messages):
//   node_t, tree_t                                      src/include/coati/tree.hpp:35-52
//   read_newick, parse_newick, find_seq, find_node,
//   reroot, distance_ref                                src/lib/tree.cc:118-451
// The reference parses with a boost::spirit grammar (tree.cc:40-107); this is a small
// recursive-descent parser for the same language:
//   tree  := node [';']        node := leaf | inode
//   leaf  := label [':' float]
//   inode := '(' node (',' node)* ')' [label] [':' float]
//   label := one or more of  - 0-9 A-Z a-z / % _ .
#ifndef COATI_AMD_HOST_TREE_HPP
#define COATI_AMD_HOST_TREE_HPP

#include <string>
#include <string_view>
#include <vector>

#include "seq.hpp"

namespace coati_amd::tree {

struct node_t {
    std::string label;
    float length{0.f};  // branch to the parent
    bool is_leaf{false};
    std::size_t parent{0};  // index in the tree; the root is its own parent
    std::vector<std::size_t> children;

    node_t(std::string name, float len, bool leaf = false, std::size_t ancestor = 0)
        : label{std::move(name)}, length{len}, is_leaf{leaf}, parent{ancestor} {}
};
using tree_t = std::vector<node_t>;  // pre-order: a node precedes its descendants, node 0 is the root

std::string read_newick(const std::string& tree_file);
tree_t parse_newick(std::string& content);  // strips tabs, newlines and blanks from `content` first
std::string find_seq(std::string_view name, const data_t& data);
std::size_t find_node(const tree_t& tree, std::string_view name);
// Make the PARENT of the node called `label` the root (parent == self, length 0), reversing the
// parent links (and moving the branch lengths) along the path to the old root.
void reroot(tree_t& tree, std::string_view label);
// Branch-length distance node -> root plus the reference's own branch.
float distance_ref(const tree_t& tree, std::size_t ref, std::size_t node);

}  // namespace coati_amd::tree
#endif
