#include <vector>
#include <algorithm>
#include <mutex>
#include <thread>
// C entry points of libcoati_host.so used by the Python tests and bench.py to
// reach the C++ host layer (model construction, sequence preparation, the
// synthetic workload).  Errors: non-zero return + coati_host_last_error().
#include <cstring>
#include <exception>
#include <fstream>
#include <stdexcept>
#include <string>

#include "align.hpp"
#include "insertions.hpp"
#include "tree.hpp"
#include "io.hpp"
#include "model.hpp"
#include "random.hpp"
#include "seq.hpp"
#include "synth.hpp"

namespace {
thread_local std::string g_err;
template <class F>
int guarded(F&& f) {
    try {
        f();
        return 0;
    } catch(const std::invalid_argument& e) {
        g_err = e.what();
        return 1;
    } catch(const std::out_of_range& e) {
        g_err = e.what();
        return 2;
    } catch(const std::exception& e) {
        g_err = e.what();
        return 3;
    }
}
}  // namespace

extern "C" {

const char* coati_host_last_error(void) { return g_err.c_str(); }

int coati_host_gap_consts(float gap_open, float gap_extend, float out[4]) {
    return guarded([&] {
        coati_amd::gap_t g;
        g.open = gap_open;
        g.extend = gap_extend;
        const auto c = coati_amd::gap_log_consts(g);
        std::memcpy(out, c.data(), sizeof(float) * 4);
    });
}

int coati_host_mg94_p(float br_len, float omega, const float pi[4], const float sigma[6], float out[3721]) {
    return guarded([&] {
        std::array<float, 4> p{pi[0], pi[1], pi[2], pi[3]};
        std::array<float, 6> s{0, 0, 0, 0, 0, 0};
        if(sigma != nullptr) std::memcpy(s.data(), sigma, sizeof(float) * 6);
        const auto P = coati_amd::mg94_p(br_len, omega, p, s);
        std::memcpy(out, P.data(), sizeof(float) * 3721);
    });
}

int coati_host_ecm_p(float br_len, float omega, float out[3721]) {
    return guarded([&] {
        const auto P = coati_amd::ecm_p(br_len, omega);
        std::memcpy(out, P.data(), sizeof(float) * 3721);
    });
}

int coati_host_gtr_q(const float pi[4], const float sigma[6], float out[16]) {
    return guarded([&] {
        const auto q = coati_amd::gtr_q({pi[0], pi[1], pi[2], pi[3]},
                                        {sigma[0], sigma[1], sigma[2], sigma[3], sigma[4], sigma[5]});
        std::memcpy(out, q.data(), sizeof(float) * 16);
    });
}

int coati_host_marginal_p(const float P[3721], const float pi[4], int amb_best, int sub_max, float out[2745]) {
    return guarded([&] {
        coati_amd::matrix61_t m(P, P + 3721);
        const auto t = coati_amd::marginal_p(m, {pi[0], pi[1], pi[2], pi[3]},
                                             amb_best ? coati_amd::AmbiguousNucs::BEST : coati_amd::AmbiguousNucs::SUM,
                                             sub_max ? coati_amd::MarginalSubst::MAX : coati_amd::MarginalSubst::SUM);
        std::memcpy(out, t.data(), sizeof(float) * 2745);
    });
}

// set_subst: model "mar-mg" | "mar-ecm"
int coati_host_set_subst(const char* model, float br_len, float omega, const float pi[4], const float sigma[6],
                         int amb_best, int sub_max, float out[2745]) {
    return guarded([&] {
        coati_amd::model_params_t prm;
        prm.model = model;
        prm.br_len = br_len;
        prm.omega = omega;
        if(pi != nullptr) prm.pi = {pi[0], pi[1], pi[2], pi[3]};
        if(sigma != nullptr) std::memcpy(prm.sigma.data(), sigma, sizeof(float) * 6);
        prm.amb = amb_best ? coati_amd::AmbiguousNucs::BEST : coati_amd::AmbiguousNucs::SUM;
        prm.sub = sub_max ? coati_amd::MarginalSubst::MAX : coati_amd::MarginalSubst::SUM;
        const auto t = coati_amd::set_subst(prm);
        std::memcpy(out, t.data(), sizeof(float) * 2745);
    });
}

// marginal_seq_encoding: a_out needs strlen(anc) bytes, b_out strlen(des)
int coati_host_encode(const char* anc, const char* des, unsigned char* a_out, unsigned char* b_out) {
    return guarded([&] {
        const auto enc = coati_amd::marginal_seq_encoding(anc, des);
        std::memcpy(a_out, enc[0].data(), enc[0].size());
        std::memcpy(b_out, enc[1].data(), enc[1].size());
    });
}

// trim_end_stops + restore_end_stops round trip on two sequences: writes the
// trimmed sequences / stops (buffers of capacity cap) and, given an alignment of
// the trimmed sequences in aln_a/aln_b (in/out) and score (in/out), restores.
int coati_host_trim_end_stops(const char* s0, const char* s1, char* t0, char* t1, char* stop0, char* stop1,
                              unsigned long long cap) {
    return guarded([&] {
        coati_amd::data_t d;
        d.names = {"a", "b"};
        d.seqs = {s0, s1};
        coati_amd::trim_end_stops(d);
        if(d.seqs[0].size() + 1 > cap || d.seqs[1].size() + 1 > cap) throw std::invalid_argument("buffer too small");
        std::strcpy(t0, d.seqs[0].c_str());
        std::strcpy(t1, d.seqs[1].c_str());
        std::strcpy(stop0, d.stops[0].c_str());
        std::strcpy(stop1, d.stops[1].c_str());
    });
}

int coati_host_restore_end_stops(char* aln0, char* aln1, const char* stop0, const char* stop1, float gap_open,
                                 float gap_extend, float* score, unsigned long long cap) {
    return guarded([&] {
        coati_amd::data_t d;
        d.names = {"a", "b"};
        d.seqs = {aln0, aln1};
        d.stops = {stop0, stop1};
        d.score = *score;
        coati_amd::gap_t g;
        g.open = gap_open;
        g.extend = gap_extend;
        coati_amd::restore_end_stops(d, g);
        if(d.seqs[0].size() + 1 > cap || d.seqs[1].size() + 1 > cap) throw std::invalid_argument("buffer too small");
        std::strcpy(aln0, d.seqs[0].c_str());
        std::strcpy(aln1, d.seqs[1].c_str());
        *score = d.score;
    });
}

// string_seed_seq + Random::Seed: n seed strings -> Lehmer state (lo, hi)
int coati_host_rng_seed(const char* const* seeds, int n, unsigned long long out[2]) {
    return guarded([&] {
        std::vector<std::string> v;
        for(int q = 0; q < n; ++q) v.emplace_back(seeds[q]);
        coati_amd::random_t r;
        r.seed(v);
        out[0] = r.lo();
        out[1] = r.hi();
    });
}

// n f24() draws from state (lo, hi); the state is advanced in place
int coati_host_rng_f24(unsigned long long state[2], int n, float* out) {
    return guarded([&] {
        coati_amd::random_t r;
        r.set_state(state[0], state[1]);
        for(int q = 0; q < n; ++q) out[q] = r.f24();
        state[0] = r.lo();
        state[1] = r.hi();
    });
}

// extract_file_type (buffers of capacity cap)
int coati_host_extract_file_type(const char* path, char* out_path, char* out_ext, unsigned long long cap) {
    return guarded([&] {
        const auto ft = coati_amd::extract_file_type(path);
        if(ft.path.size() + 1 > cap || ft.type_ext.size() + 1 > cap) throw std::invalid_argument("buffer too small");
        std::strcpy(out_path, ft.path.c_str());
        std::strcpy(out_ext, ft.type_ext.c_str());
    });
}

// read_input(in) -> write_output(out): format conversion through the readers/writers
int coati_host_convert(const char* in_path, const char* out_path, float score) {
    return guarded([&] {
        coati_amd::data_t d = coati_amd::read_input(in_path);
        if(score == score) d.score = score;  // NaN = keep what the reader found
        coati_amd::write_output(d, out_path);
    });
}

// element-of-array JSON writer used by `sample` and the batch extension
int coati_host_write_json_array(const char* in_path, const char* out_path, unsigned count) {
    return guarded([&] {
        const coati_amd::data_t d = coati_amd::read_input(in_path);
        std::ofstream out(out_path);
        for(unsigned i = 0; i < count; ++i) coati_amd::write_json(d, out, i, count);
    });
}

int coati_host_json_number(float v, char* out, unsigned long long cap) {
    return guarded([&] {
        const std::string s = coati_amd::json_number(v);
        if(s.size() + 1 > cap) throw std::invalid_argument("buffer too small");
        std::strcpy(out, s.c_str());
    });
}

// alignment_score of an aligned pair under a marginal model (mar-mg / mar-ecm)
int coati_host_alignment_score(const char* aln_anc, const char* aln_des, const char* model, float gap_open,
                               float gap_extend, unsigned gap_len, float* score) {
    return guarded([&] {
        coati_amd::alignment_t aln;
        aln.model = model;
        aln.gap.open = gap_open;
        aln.gap.extend = gap_extend;
        aln.gap.len = gap_len;
        aln.data.names = {"A", "B"};
        aln.data.seqs = {aln_anc, aln_des};
        coati_amd::set_subst(aln);
        *score = coati_amd::alignment_score(aln, aln.subst_matrix);
    });
}

int coati_host_parse_matrix_csv(const char* path, float out[3721]) {
    return guarded([&] {
        const auto P = coati_amd::parse_matrix_csv(path);
        std::memcpy(out, P.data(), sizeof(float) * 3721);
    });
}

// Encoded synthetic pairs [first, first+n): offsets have n+1 entries; call with
// a_cat == NULL to size the buffers (offsets are filled either way).
int coati_host_batch_reader_check(const char* path, long* first_difference) {
    return guarded([&] { *first_difference = coati_amd::batch_reader_first_difference(path); });
}

// The host side of one rank of `coati-alignpair --batch --devices` without a device: shard plan from the index, the
// shard parsed + encoded by the block pipeline's stage A, compared with the generic reader (align.cc).
int coati_host_batch_shard_check(const char* path, int world, int rank, unsigned long long* s0, unsigned long long* s1, long* first_difference) {
    return guarded([&] {
        coati_amd::alignment_t aln;
        aln.data.path = path;
        uint64_t a = 0, b = 0;
        *first_difference = coati_amd::batch_shard_first_difference(aln, world, rank, &a, &b);
        *s0 = a, *s1 = b;
    });
}

int coati_host_synth_encoded(unsigned long long first, unsigned long long n, unsigned long long seed_base,
                             unsigned n_codons, unsigned char* a_cat, unsigned long long* a_off,
                             unsigned char* b_cat, unsigned long long* b_off) {
    return guarded([&] {
        coati_amd::synth_params_t prm;
        prm.seed_base = seed_base;
        prm.n_codons = n_codons;
        // pair p is a function of its index: lengths first (parallel), prefix sums, then the codes in place
        const unsigned n_threads = static_cast<unsigned>(std::min<unsigned long long>(
            std::max<unsigned long long>(1, n / 512), std::min(32u, std::max(1u, std::thread::hardware_concurrency()))));
        auto run = [&](auto&& body) {
            std::vector<std::thread> pool;
            std::exception_ptr failure;
            std::mutex lock;
            for(unsigned t = 0; t < n_threads; ++t)
                pool.emplace_back([&, t] {
                    try {
                        body(n * t / n_threads, n * (t + 1) / n_threads);
                    } catch(...) {
                        std::lock_guard<std::mutex> hold(lock);
                        if(!failure) failure = std::current_exception();
                    }
                });
            for(auto& th : pool) th.join();
            if(failure) std::rethrow_exception(failure);
        };
        a_off[0] = b_off[0] = 0;
        run([&](unsigned long long lo, unsigned long long hi) {
            std::string anc, des;
            for(unsigned long long p = lo; p < hi; ++p) {
                coati_amd::synth_pair(first + p, prm, anc, des);
                a_off[p + 1] = anc.size();
                b_off[p + 1] = des.size();
            }
        });
        for(unsigned long long p = 0; p < n; ++p) {
            a_off[p + 1] += a_off[p];
            b_off[p + 1] += b_off[p];
        }
        if(a_cat == nullptr) return;
        run([&](unsigned long long lo, unsigned long long hi) {
            std::string anc, des;
            for(unsigned long long p = lo; p < hi; ++p) {
                coati_amd::synth_pair(first + p, prm, anc, des);
                const auto enc = coati_amd::marginal_seq_encoding(anc, des);
                std::memcpy(a_cat + a_off[p], enc[0].data(), enc[0].size());
                std::memcpy(b_cat + b_off[p], enc[1].data(), enc[1].size());
            }
        });
    });
}

// Raw text of synthetic pair `index` (buffers of capacity cap each, NUL-terminated).
int coati_host_synth_raw(unsigned long long index, unsigned long long seed_base, unsigned n_codons, char* anc,
                         char* des, unsigned long long cap) {
    return guarded([&] {
        coati_amd::synth_params_t prm;
        prm.seed_base = seed_base;
        prm.n_codons = n_codons;
        std::string a, d;
        coati_amd::synth_pair(index, prm, a, d);
        if(a.size() + 1 > cap || d.size() + 1 > cap) throw std::invalid_argument("buffer too small");
        std::memcpy(anc, a.c_str(), a.size() + 1);
        std::memcpy(des, d.c_str(), d.size() + 1);
    });
}

// align_leafs for tests/bindings: leaves are NUL-terminated strings; outputs are written as
// 2 strings per leaf into `out` (slot = len(ref) + longest leaf + 1 bytes each).
int coati_host_align_leafs(const char* model, float omega, float gap_open, float gap_extend, unsigned gap_len,
                           const char* ref_seq, const char* const* leaves, const float* br_lens, unsigned n_leaves,
                           char* out, unsigned long long slot, float* scores) {
    return guarded([&] {
        coati_amd::alignment_t aln;
        aln.model = model;
        aln.omega = omega;
        aln.gap.open = gap_open;
        aln.gap.extend = gap_extend;
        aln.gap.len = gap_len;
        std::vector<std::string> ls(leaves, leaves + n_leaves);
        std::vector<float> bl(br_lens, br_lens + n_leaves);
        const auto res = coati_amd::align_leafs(aln, ref_seq, ls, bl);
        for(unsigned p = 0; p < n_leaves; ++p) {
            if(res[p].seqs[0].size() + 1 > slot) throw std::invalid_argument("align_leafs: output slot too small");
            std::strcpy(out + (2ull * p) * slot, res[p].seqs[0].c_str());
            std::strcpy(out + (2ull * p + 1) * slot, res[p].seqs[1].c_str());
            scores[p] = res[p].score;
        }
    });
}

// ---- msa host pieces, for the unit tests (text in, text out) ---------------------------------
// Parsed (optionally re-rooted at reroot_label's parent) guide tree, one node per line:
//   index <TAB> label <TAB> length <TAB> is_leaf <TAB> parent
int coati_host_newick(const char* text, const char* reroot_label, char* out, unsigned long long cap) {
    return guarded([&] {
        std::string content(text);
        coati_amd::tree::tree_t tree = coati_amd::tree::parse_newick(content);
        if(reroot_label != nullptr && reroot_label[0] != '\0') coati_amd::tree::reroot(tree, reroot_label);
        std::string res;
        for(std::size_t i = 0; i < tree.size(); ++i) {
            char buf[64];
            std::snprintf(buf, sizeof buf, "%.9g", static_cast<double>(tree[i].length));
            res += std::to_string(i) + "\t" + tree[i].label + "\t" + buf + "\t" + (tree[i].is_leaf ? "1" : "0") + "\t" +
                   std::to_string(tree[i].parent) + "\n";
        }
        if(res.size() + 1 > cap) throw std::invalid_argument("newick: output buffer too small");
        std::memcpy(out, res.c_str(), res.size() + 1);
    });
}

int coati_host_tree_distance(const char* text, const char* ref_label, const char* node_label, int do_reroot, float* out) {
    return guarded([&] {
        std::string content(text);
        coati_amd::tree::tree_t tree = coati_amd::tree::parse_newick(content);
        if(do_reroot) coati_amd::tree::reroot(tree, ref_label);
        *out = coati_amd::tree::distance_ref(tree, coati_amd::tree::find_node(tree, ref_label),
                                             coati_amd::tree::find_node(tree, node_label));
    });
}

// merge_indels on sets given one per line as  names;seqs;flag_length;pos=flag,pos=flag,...
// (names and seqs comma separated).  Output: one line in the same format (flag_length omitted).
int coati_host_merge_indels(const char* spec, char* out, unsigned long long cap) {
    return guarded([&] {
        auto split = [](const std::string& s, char sep) {
            std::vector<std::string> parts;
            std::size_t from = 0;
            for(;;) {
                const std::size_t at = s.find(sep, from);
                parts.push_back(s.substr(from, at == std::string::npos ? std::string::npos : at - from));
                if(at == std::string::npos) break;
                from = at + 1;
            }
            return parts;
        };
        coati_amd::insertion_vector sets;
        for(const std::string& line : split(spec, '\n')) {
            if(line.empty()) continue;
            const auto f = split(line, ';');
            if(f.size() != 4) throw std::invalid_argument("merge_indels: bad set line");
            coati_amd::flag_vector flags(static_cast<std::size_t>(std::stoul(f[2])), 0);
            if(!f[3].empty())
                for(const std::string& kv : split(f[3], ',')) {
                    const auto pv = split(kv, '=');
                    flags.at(std::stoul(pv.at(0))) = std::stoi(pv.at(1));
                }
            sets.emplace_back(split(f[1], ','), split(f[0], ','), flags);
        }
        coati_amd::insertion_data_t merged;
        coati_amd::merge_indels(sets, merged);
        std::string res;
        for(std::size_t i = 0; i < merged.names.size(); ++i) res += (i ? "," : "") + merged.names[i];
        res += ";";
        for(std::size_t i = 0; i < merged.sequences.size(); ++i) res += (i ? "," : "") + merged.sequences[i];
        res += ";";
        bool first = true;
        for(std::size_t i = 0; i < merged.insertions.size(); ++i)
            if(merged.insertions[i] != 0) {
                res += (first ? "" : ",") + std::to_string(i) + "=" + std::to_string(merged.insertions[i]);
                first = false;
            }
        if(res.size() + 1 > cap) throw std::invalid_argument("merge_indels: output buffer too small");
        std::memcpy(out, res.c_str(), res.size() + 1);
    });
}

}  // extern "C"
