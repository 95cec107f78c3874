#include "tree.hpp"

#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <stdexcept>

namespace coati_amd::tree {

std::string read_newick(const std::string& tree_file) {
    std::ifstream in(tree_file);
    if(!in.good()) throw std::invalid_argument("Error opening " + tree_file + ".");
    std::string content((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    if(content.empty()) throw std::invalid_argument("Reading tree failed, file is empty!");
    return content;
}

namespace {

struct parser_t {
    const std::string& s;
    std::size_t at{0};
    tree_t out;

    static bool label_char(char c) {
        return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '-' || c == '/' || c == '%' ||
               c == '_' || c == '.';
    }
    std::string label() {
        const std::size_t from = at;
        while(at < s.size() && label_char(s[at])) ++at;
        return s.substr(from, at - from);
    }
    bool length(float& len) {  // [':' float]; false = syntax error after ':'
        len = 0.f;
        if(at >= s.size() || s[at] != ':') return true;
        const char* begin = s.c_str() + at + 1;
        char* end = nullptr;
        const float v = std::strtof(begin, &end);
        if(end == begin) return false;
        len = v;
        at += 1 + static_cast<std::size_t>(end - begin);
        return true;
    }
    // parses one node (and its subtree) as a child of `parent`; returns false on a syntax error
    bool node(std::size_t parent) {
        if(at < s.size() && s[at] == '(') {
            const std::size_t self = out.size();
            out.emplace_back("", 0.f, false, self == 0 ? 0 : parent);
            ++at;
            for(;;) {
                if(!node(self)) return false;
                if(at < s.size() && s[at] == ',') {
                    ++at;
                    continue;
                }
                break;
            }
            if(at >= s.size() || s[at] != ')') return false;
            ++at;
            out[self].label = label();  // optional
            float len = 0.f;
            if(!length(len)) return false;
            out[self].length = len;
            return true;
        }
        std::string name = label();
        if(name.empty()) return false;  // a leaf needs a label
        float len = 0.f;
        if(!length(len)) return false;
        const std::size_t self = out.size();
        out.emplace_back(std::move(name), len, true, self == 0 ? 0 : parent);
        return true;
    }
};

}  // namespace

tree_t parse_newick(std::string& content) {
    content.erase(std::remove_if(content.begin(), content.end(), [](char c) { return c == '\t' || c == '\n' || c == ' '; }),
                  content.end());
    parser_t p{content, 0, {}};
    bool ok = !content.empty() && p.node(0);
    if(ok && p.at < content.size() && content[p.at] == ';') ++p.at;
    if(!ok || p.at != content.size()) throw std::runtime_error("Parsing content of newick tree failed.");
    return std::move(p.out);
}

std::string find_seq(std::string_view name, const data_t& data) {
    const auto it = std::find(data.names.cbegin(), data.names.cend(), name);
    if(it == data.names.cend()) throw std::invalid_argument("Sequence " + std::string(name) + " not found.");
    return data.seqs[static_cast<std::size_t>(it - data.names.cbegin())];
}

std::size_t find_node(const tree_t& tree, std::string_view name) {
    for(std::size_t i = 0; i < tree.size(); ++i)
        if(tree[i].label == name) return i;
    throw std::invalid_argument("Node " + std::string(name) + " not found.");
}

void reroot(tree_t& tree, std::string_view label) {
    const std::size_t new_root = tree[find_node(tree, label)].parent;
    // the path new_root -> ... -> old root
    std::vector<std::size_t> path;
    for(std::size_t n = new_root;; n = tree[n].parent) {
        path.push_back(n);
        if(tree[n].parent == n) break;
    }
    // every node on the path becomes the child of the node before it and inherits that branch
    for(std::size_t i = path.size() - 1; i > 0; --i) {
        tree[path[i]].parent = path[i - 1];
        tree[path[i]].length = tree[path[i - 1]].length;
    }
    tree[new_root].parent = new_root;
    tree[new_root].length = 0.f;
}

float distance_ref(const tree_t& tree, std::size_t ref, std::size_t node) {
    float d = 0.f;
    for(; tree[node].parent != node; node = tree[node].parent) d += tree[node].length;
    return d + tree[ref].length;
}

}  // namespace coati_amd::tree
