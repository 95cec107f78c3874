#include "align.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <exception>
#include <fstream>
#include <iostream>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "coati_hip.h"
#include "codon.hpp"
#include "io.hpp"

namespace coati_amd {

namespace {
[[noreturn]] void throw_hip(int rc) {
    const std::string msg = coati_hip_last_error();
    if(rc == COATI_HIP_ENOMEM) throw std::bad_alloc();
    if(rc == COATI_HIP_EINVAL) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}
void hip_check(int rc) {
    if(rc != COATI_HIP_OK) throw_hip(rc);
}
// COATI_HOST_TIMING=1: wall time of the stages of a driver on stderr
struct host_timer {
    const char* what;
    bool on;
    std::chrono::steady_clock::time_point t0, prev;
    explicit host_timer(const char* w) : what(w), on(std::getenv("COATI_HOST_TIMING") != nullptr) { t0 = prev = std::chrono::steady_clock::now(); }
    void stage(const char* name) {
        if(!on) return;
        const auto t = std::chrono::steady_clock::now();
        std::cerr << what << ": " << name << " " << std::chrono::duration<double, std::milli>(t - prev).count() << " ms (total "
                  << std::chrono::duration<double, std::milli>(t - t0).count() << ")\n";
        prev = t;
    }
};

// The kernels mark a pair they could not finish (a strip hand-off that timed out) with a NaN score
// instead of hanging the GPU: that is an error of the run, not a result to print as `null`.
void check_scores(const std::vector<float>& scores) {
    for(std::size_t p = 0; p < scores.size(); ++p)
        if(std::isnan(scores[p])) throw std::runtime_error("The device did not finish pair " + std::to_string(p) + " (strip hand-off timed out).");
}

coati_hip_model* make_model(const alignment_t& aln) {
    if(aln.gap.len < 1) throw std::invalid_argument("Gap unit length must be positive.");
    if(aln.subst_matrix.size() != kTableRows * kTableCols) throw std::invalid_argument("Substitution matrix not set.");
    const auto k = gap_log_consts(aln.gap);
    coati_hip_model* m = nullptr;
    hip_check(coati_hip_model_create(aln.subst_matrix.data(), k[0], k[1], k[2], k[3], static_cast<int>(aln.gap.len),
                                     aln.device, &m));
    return m;
}
void release(align_pair_work_mem_t& w) {
    if(w.batch != nullptr) coati_hip_batch_destroy(w.batch);
    if(w.model != nullptr) coati_hip_model_destroy(w.model);
    w.batch = nullptr;
    w.model = nullptr;
}
void make_pair_batch(align_pair_work_mem_t& work, const seq_view_t& a, const seq_view_t& b, const alignment_t& aln) {
    release(work);
    work.model = make_model(aln);
    const uint64_t a_off[2] = {0, a.size()}, b_off[2] = {0, b.size()};
    hip_check(coati_hip_batch_create(work.model, 1, a.data(), a_off, b.data(), b_off, &work.batch));
}
}  // namespace

align_pair_work_mem_t::~align_pair_work_mem_t() { release(*this); }

namespace {
// fn(i) for i in [0, n) on up to 16 host threads (the per-pair host work either side of a batched
// launch: tables, encoding, gapped strings).  The exception of the lowest failing index is rethrown on
// the caller's thread.
template <typename Fn>
void parallel_for(std::size_t n, std::size_t min_per_thread, Fn&& fn) {
    const std::size_t want = std::max<std::size_t>(1, n / std::max<std::size_t>(1, min_per_thread));
    const std::size_t n_threads = std::min<std::size_t>({want, std::max(1u, std::thread::hardware_concurrency()), 16});
    if(n_threads <= 1) {
        for(std::size_t i = 0; i < n; ++i) fn(i);
        return;
    }
    const std::size_t chunk = std::max<std::size_t>(1, std::min<std::size_t>(min_per_thread, n / (4 * n_threads) + 1));
    std::atomic<std::size_t> next{0};
    std::exception_ptr failure;
    std::size_t failure_at = n;  // the LOWEST failing index wins, as in a serial loop: chunks are handed
    std::mutex failure_lock;     // out in ascending order and a started chunk always runs to its end
    auto worker = [&]() {
        for(;;) {
            const std::size_t lo = next.fetch_add(chunk);
            if(lo >= n) return;
            std::size_t i = lo;
            try {
                for(; i < std::min(n, lo + chunk); ++i) fn(i);
            } catch(...) {
                std::lock_guard<std::mutex> hold(failure_lock);
                if(i < failure_at) {
                    failure_at = i;
                    failure = std::current_exception();
                }
                next.store(n);
                return;
            }
        }
    };
    std::vector<std::thread> pool;
    for(std::size_t w = 1; w < n_threads; ++w) pool.emplace_back(worker);
    worker();
    for(auto& th : pool) th.join();
    if(failure) std::rethrow_exception(failure);
}
}  // namespace

void set_subst(alignment_t& aln) {
    if(!aln.rate.empty()) {
        aln.model = "user_marg_model";
        aln.subst_matrix = marginal_p(parse_matrix_csv(aln.rate), aln.pi, aln.amb, aln.sub);
        return;
    }
    model_params_t prm;
    prm.model = aln.model;
    prm.br_len = aln.br_len;
    prm.omega = aln.omega;
    prm.pi = aln.pi;
    prm.sigma = aln.sigma;
    prm.amb = aln.amb;
    prm.sub = aln.sub;
    aln.subst_matrix = set_subst(prm);
}

void viterbi_mem(align_pair_work_mem_t& work, const seq_view_t& a, const seq_view_t& b, const alignment_t& aln) {
    make_pair_batch(work, a, b, aln);
    hip_check(coati_hip_viterbi_launch(work.batch));
    work.ops.assign(a.size() + b.size() + 1, 0);
    uint64_t off = 0;
    uint32_t len = 0;
    hip_check(coati_hip_viterbi_fetch(work.batch, &work.score, work.ops.data(), a.size() + b.size(), &off, &len));
    check_scores({work.score});
    work.ops.erase(work.ops.begin(), work.ops.begin() + static_cast<std::ptrdiff_t>(off));
    work.ops.resize(len);
}

void traceback_viterbi(const align_pair_work_mem_t& work, const std::string& a, const std::string& b,
                       alignment_t& aln, std::size_t /*look_back*/) {
    aln.data.seqs.assign(2, std::string());
    ops_to_alignment(work.ops.data(), work.ops.size(), a, b, aln.data.seqs[0], aln.data.seqs[1]);
    aln.data.score = work.score;
}

void forward(align_pair_work_t& work, const seq_view_t& a, const seq_view_t& b, const alignment_t& aln) {
    make_pair_batch(work, a, b, aln);
    hip_check(coati_hip_forward_launch(work.batch));
}

void sampleback(const align_pair_work_t& work, const std::string& a, const std::string& b, alignment_t& aln,
                std::size_t /*look_back*/, random_t& rand) {
    if(work.batch == nullptr) throw std::runtime_error("sampleback: forward() was not run.");
    const uint64_t state_in[2] = {rand.lo(), rand.hi()};
    uint64_t state_out[2] = {0, 0};
    std::vector<uint8_t> ops(a.size() + b.size() + 1);
    uint64_t off = 0;
    uint32_t len = 0;
    float lw = 0.f;
    hip_check(coati_hip_sampleback(work.batch, 1, state_in, /*independent_streams=*/0, &lw, ops.data(),
                                   a.size() + b.size(), &off, &len, state_out));
    rand.set_state(state_out[0], state_out[1]);  // the stream continues where the device walker stopped
    aln.data.seqs.assign(2, std::string());
    ops_to_alignment(ops.data() + off, len, a, b, aln.data.seqs[0], aln.data.seqs[1]);
    aln.data.score = lw;
}

bool marg_alignment(alignment_t& aln) {
    const bool timing = std::getenv("COATI_HOST_TIMING") != nullptr;  // stage times on stderr
    auto t_prev = std::chrono::steady_clock::now();
    auto stage = [&](const char* what) {
        if(!timing) return;
        const auto t = std::chrono::steady_clock::now();
        std::cerr << "marg_alignment: " << what << " " << std::chrono::duration<double, std::milli>(t - t_prev).count() << " ms\n";
        t_prev = t;
    };
    aln.data = read_input(aln.data.path);
    set_subst(aln);
    stage("input + model");
    if(aln.score) {
        std::cout << alignment_score(aln, aln.subst_matrix) << std::endl;
        return true;
    }
    process_marginal(aln.data, aln.gap, aln.refs, aln.rev);
    const std::string anc = aln.seq(0), des = aln.seq(1);
    const auto seq_pair = marginal_seq_encoding(anc, des);
    check_descendant_codes(seq_pair[1]);
    align_pair_work_mem_t work;
    try {
        viterbi_mem(work, seq_pair[0], seq_pair[1], aln);
    } catch(const std::bad_alloc&) {
        std::cerr << "ERROR: sequences to align exceed available memory." << std::endl;
        return false;  // (upstream returns EXIT_FAILURE from a bool function, i.e. true: align_marginal.cc:72-75)
    }
    stage("viterbi_mem (HIP runtime start, upload, fill + traceback)");
    traceback_viterbi(work, anc, des, aln, aln.gap.len);
    restore_end_stops(aln.data, aln.gap);
    write_output(aln.data, aln.output);
    stage("fetch + output");
    return true;
}

std::vector<data_t> align_leafs(alignment_t& input, const std::string& ref_seq, const std::vector<std::string>& leaf_seqs,
                                const std::vector<float>& br_lens) {
    if(leaf_seqs.size() != br_lens.size()) throw std::invalid_argument("One branch length per leaf is required.");
    const std::size_t n = leaf_seqs.size();
    // COATI_HOST_TIMING=1: stage times on stderr (tables / encode / device / strings)
    const bool timing = std::getenv("COATI_HOST_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_prev = now();
    auto stage = [&](const char* what) {
        if(!timing) return;
        const auto t = now();
        std::cerr << "align_leafs: " << what << " " << std::chrono::duration<double, std::milli>(t - t_prev).count() << " ms\n";
        t_prev = t;
    };
    std::vector<data_t> out(n);
    if(n == 0) return out;
    // one table per distinct branch length (set_subst per leaf, align_msa.cc:297-301); a table costs
    // ~3 ms of host time (61x61 matrix exponential), so distinct lengths are built on several threads
    std::vector<float> distinct;
    std::vector<uint32_t> table_index(n);
    for(std::size_t p = 0; p < n; ++p) {
        std::size_t t = 0;
        while(t < distinct.size() && distinct[t] != br_lens[p]) ++t;
        if(t == distinct.size()) distinct.push_back(br_lens[p]);
        table_index[p] = static_cast<uint32_t>(t);
    }
    constexpr std::size_t kTable = 183 * 15;
    std::vector<float> tables(distinct.size() * kTable);
    auto store = [&](std::size_t t, const table_t& tab) { std::copy(tab.begin(), tab.end(), tables.begin() + t * kTable); };
    if(!input.rate.empty()) {  // user matrix: the file fixes the branch length, every leaf shares one table
        set_subst(input);
        for(std::size_t t = 0; t < distinct.size(); ++t) store(t, input.subst_matrix);
    } else {
        model_params_t prm;
        prm.model = input.model;
        prm.omega = input.omega;
        prm.pi = input.pi;
        prm.sigma = input.sigma;
        prm.amb = input.amb;
        prm.sub = input.sub;
        parallel_for(distinct.size(), 1, [&](std::size_t t) {
            model_params_t mine = prm;
            mine.br_len = distinct[t];
            store(t, set_subst(mine));
        });
        // leave `input` as the serial loop of the reference does: the last leaf's model
        input.br_len = br_lens.back();
        const table_t last(tables.begin() + table_index[n - 1] * kTable, tables.begin() + (table_index[n - 1] + 1) * kTable);
        input.subst_matrix = last;
    }
    stage("tables");
    const encoded_t ref_codes = marginal_seq_encoding(ref_seq, std::string_view())[0];  // the same ancestor for every leaf
    std::vector<uint64_t> a_off(n + 1, 0), b_off(n + 1, 0);
    for(std::size_t p = 0; p < n; ++p) {
        a_off[p + 1] = a_off[p] + ref_codes.size();
        b_off[p + 1] = b_off[p] + leaf_seqs[p].size();
    }
    std::vector<unsigned char> a_cat(a_off[n]), b_cat(b_off[n]);
    parallel_for(n, 256, [&](std::size_t p) {
        std::copy(ref_codes.begin(), ref_codes.end(), a_cat.begin() + static_cast<std::ptrdiff_t>(a_off[p]));
        unsigned char* des = b_cat.data() + b_off[p];
        encode_descendant(leaf_seqs[p], des);
        for(std::size_t i = 0; i < leaf_seqs[p].size(); ++i)
            if(des[i] >= kTableCols) throw std::invalid_argument("Invalid character in descendant sequence.");
    });
    if(input.gap.len < 1) throw std::invalid_argument("Gap unit length must be positive.");
    stage("encode");
    const auto k = gap_log_consts(input.gap);
    coati_hip_model* model = nullptr;
    hip_check(coati_hip_model_create_tables(tables.data(), static_cast<uint32_t>(distinct.size()), k[0], k[1], k[2], k[3],
                                            static_cast<int>(input.gap.len), input.device, &model));
    coati_hip_batch* batch = nullptr;
    int rc = coati_hip_batch_create_tables(model, n, a_cat.data(), a_off.data(), b_cat.data(), b_off.data(),
                                           table_index.data(), &batch);
    std::vector<float> scores(n);
    std::vector<uint8_t> ops(a_cat.size() + b_cat.size() + 1);
    std::vector<uint64_t> off(n);
    std::vector<uint32_t> len(n);
    if(rc == COATI_HIP_OK) rc = coati_hip_viterbi_launch(batch);
    if(rc == COATI_HIP_OK)
        rc = coati_hip_viterbi_fetch(batch, scores.data(), ops.data(), a_cat.size() + b_cat.size(), off.data(), len.data());
    if(batch != nullptr) coati_hip_batch_destroy(batch);
    coati_hip_model_destroy(model);
    hip_check(rc);
    check_scores(scores);  // (after the handles are gone: it throws)
    stage("device (upload, plan, launch, fetch)");
    parallel_for(n, 256, [&](std::size_t p) {
        out[p].seqs.assign(2, std::string());
        ops_to_alignment(ops.data() + off[p], len[p], ref_seq, leaf_seqs[p], out[p].seqs[0], out[p].seqs[1]);
        out[p].score = scores[p];
    });
    stage("gapped strings");
    return out;
}

namespace {
// The encoded pairs of a --batch input and what is needed to print them again.
struct batch_input_t {
    std::size_t n{0};
    std::vector<data_t> pairs;
    std::vector<std::string> ancs, dess;
    std::vector<uint64_t> a_off, b_off;
    std::vector<unsigned char> a_cat, b_cat;
};
batch_input_t read_batch_input(alignment_t& aln) {
    data_t all = read_input(aln.data.path);
    if(all.size() == 0 || all.size() % 2 != 0) throw std::invalid_argument("Batch input needs an even number of sequences.");
    batch_input_t in;
    const std::size_t n = in.n = all.size() / 2;
    in.pairs.resize(n), in.ancs.resize(n), in.dess.resize(n);
    parallel_for(n, 256, [&](std::size_t p) {
        in.pairs[p].names = {all.names[2 * p], all.names[2 * p + 1]};
        in.pairs[p].seqs = {all.seqs[2 * p], all.seqs[2 * p + 1]};
        process_marginal(in.pairs[p], aln.gap, std::string(), aln.rev);
        in.ancs[p] = in.pairs[p].seqs[0];
        in.dess[p] = in.pairs[p].seqs[1];
    });
    in.a_off.assign(n + 1, 0), in.b_off.assign(n + 1, 0);
    for(std::size_t p = 0; p < n; ++p) {
        in.a_off[p + 1] = in.a_off[p] + in.ancs[p].size();
        in.b_off[p + 1] = in.b_off[p] + in.dess[p].size();
    }
    in.a_cat.resize(in.a_off[n]), in.b_cat.resize(in.b_off[n]);
    parallel_for(n, 256, [&](std::size_t p) {
        encode_ancestor(in.ancs[p], in.a_cat.data() + in.a_off[p]);
        unsigned char* des = in.b_cat.data() + in.b_off[p];
        encode_descendant(in.dess[p], des);
        for(std::size_t i = 0; i < in.dess[p].size(); ++i)
            if(des[i] >= kTableCols) throw std::invalid_argument("Invalid character in descendant sequence.");
    });
    return in;
}
void write_batch_output(alignment_t& aln, batch_input_t& in, const std::vector<float>& scores, const std::vector<uint8_t>& ops,
                        const std::vector<uint64_t>& off, const std::vector<uint32_t>& len) {
    std::ofstream file;
    std::ostream* out = &std::cout;
    if(!(aln.output.empty() || aln.output == "-")) {
        file.open(extract_file_type(aln.output).path);
        if(!file) throw std::invalid_argument("Opening output file " + aln.output + " failed.");
        out = &file;
    }
    // gapped strings for all pairs in parallel, then the (ordered) JSON stream
    parallel_for(in.n, 64, [&](std::size_t p) {
        in.pairs[p].seqs.assign(2, std::string());
        ops_to_alignment(ops.data() + off[p], len[p], in.ancs[p], in.dess[p], in.pairs[p].seqs[0], in.pairs[p].seqs[1]);
        in.pairs[p].score = scores[p];
        restore_end_stops(in.pairs[p], aln.gap);
    });
    for(std::size_t p = 0; p < in.n; ++p) write_json(in.pairs[p], *out, p, in.n);
}
}  // namespace

bool marg_alignment_batch(alignment_t& aln) {
    host_timer tm("alignpair --batch");
    batch_input_t in = read_batch_input(aln);
    tm.stage("read + encode");
    set_subst(aln);
    coati_hip_model* model = make_model(aln);
    tm.stage("model");
    const std::size_t n = in.n;
    std::vector<float> scores(n);
    std::vector<uint8_t> ops(in.a_cat.size() + in.b_cat.size() + 1);
    std::vector<uint64_t> off(n);
    std::vector<uint32_t> len(n);
    const int rc = coati_hip_viterbi_batch(model, n, in.a_cat.data(), in.a_off.data(), in.b_cat.data(), in.b_off.data(),
                                           scores.data(), ops.data(), in.a_cat.size() + in.b_cat.size(), off.data(), len.data());
    coati_hip_model_destroy(model);
    hip_check(rc);
    check_scores(scores);
    tm.stage("device (upload, kernels, download)");
    write_batch_output(aln, in, scores, ops, off, len);
    tm.stage("gapped strings + output");
    return true;
}

namespace {
// libcoati_hip_dist.so (the RCCL layer) is loaded on demand, so that the single-GPU tools run on a box
// without librccl.
struct dist_api_t {
    void* handle{nullptr};
    const char* (*last_error)(){nullptr};
    int (*unique_id)(void*){nullptr};
    int (*init)(const void*, int, int, int, void**){nullptr};
    void (*destroy)(void*){nullptr};
    int (*broadcast_model)(void*, int, float*, uint32_t, uint32_t*, float*, int*){nullptr};
    int (*viterbi)(void*, int, coati_hip_model*, uint64_t, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, float*,
                   uint8_t*, uint64_t, uint64_t*, uint32_t*){nullptr};
};
dist_api_t load_dist_api() {
    dist_api_t api;
    // next to libcoati_host.so (same directory as this library)
    std::string dir;
    Dl_info info;
    if(dladdr(reinterpret_cast<void*>(&load_dist_api), &info) != 0 && info.dli_fname != nullptr) {
        dir = info.dli_fname;
        const std::size_t slash = dir.rfind('/');
        dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
    }
    for(const std::string& path : {dir + "libcoati_hip_dist.so", std::string("libcoati_hip_dist.so")}) {
        api.handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if(api.handle != nullptr) break;
    }
    if(api.handle == nullptr) throw std::runtime_error(std::string("--devices needs libcoati_hip_dist.so (make dist): ") + dlerror());
    auto sym = [&](const char* name) {
        void* p = dlsym(api.handle, name);
        if(p == nullptr) throw std::runtime_error(std::string("libcoati_hip_dist.so lacks ") + name);
        return p;
    };
    api.last_error = reinterpret_cast<decltype(api.last_error)>(sym("coati_hip_dist_last_error"));
    api.unique_id = reinterpret_cast<decltype(api.unique_id)>(sym("coati_hip_dist_unique_id"));
    api.init = reinterpret_cast<decltype(api.init)>(sym("coati_hip_dist_init"));
    api.destroy = reinterpret_cast<decltype(api.destroy)>(sym("coati_hip_dist_destroy"));
    api.broadcast_model = reinterpret_cast<decltype(api.broadcast_model)>(sym("coati_hip_dist_broadcast_model"));
    api.viterbi = reinterpret_cast<decltype(api.viterbi)>(sym("coati_hip_dist_viterbi"));
    return api;
}
}  // namespace

bool marg_alignment_batch_dist(alignment_t& aln, int rank, int world, const std::string& id_file) {
    if(world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("--devices: bad rank / world");
    host_timer tm(rank == 0 ? "alignpair --batch --devices (rank 0)" : "alignpair --batch --devices");
    const dist_api_t api = load_dist_api();
    auto dist_check = [&](int rc) {
        if(rc != 0) throw std::runtime_error(api.last_error());
    };
    batch_input_t in = read_batch_input(aln);  // every rank reads the same input: only results cross the links
    tm.stage("read + encode");
    // ---- rendezvous: rank 0 leaves the id in a file (written under another name, then renamed)
    unsigned char id[128];
    if(rank == 0) {
        dist_check(api.unique_id(id));
        const std::string tmp = id_file + ".tmp";
        std::ofstream f(tmp, std::ios::binary);
        f.write(reinterpret_cast<const char*>(id), sizeof id);
        f.close();
        if(!f || std::rename(tmp.c_str(), id_file.c_str()) != 0) throw std::runtime_error("--devices: cannot write " + id_file);
    } else {
        bool got = false;
        for(int tries = 0; tries < 6000 && !got; ++tries) {  // up to a minute
            std::ifstream f(id_file, std::ios::binary);
            if(f && f.read(reinterpret_cast<char*>(id), sizeof id) && f.gcount() == static_cast<std::streamsize>(sizeof id)) got = true;
            if(!got) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        if(!got) throw std::runtime_error("--devices: rank 0 never published the rendezvous id (" + id_file + ")");
    }
    void* comm = nullptr;
    dist_check(api.init(id, world, rank, aln.device, &comm));
    struct comm_guard {
        const dist_api_t& api;
        void* c;
        ~comm_guard() { api.destroy(c); }
    } guard{api, comm};
    tm.stage("communicator");
    // ---- the model: computed on rank 0 only, broadcast, so every rank scores with the same bits
    std::vector<float> table(kTableRows * kTableCols, 0.f);
    uint32_t n_tables = 1;
    float consts[4] = {0, 0, 0, 0};
    int gap_len = static_cast<int>(aln.gap.len);
    if(rank == 0) {
        set_subst(aln);
        std::copy(aln.subst_matrix.data(), aln.subst_matrix.data() + table.size(), table.begin());
        const auto k = gap_log_consts(aln.gap);
        std::copy(k.begin(), k.end(), consts);
    }
    dist_check(api.broadcast_model(comm, 0, table.data(), 1, &n_tables, consts, &gap_len));
    coati_hip_model* model = nullptr;
    hip_check(coati_hip_model_create(table.data(), consts[0], consts[1], consts[2], consts[3], gap_len, aln.device, &model));
    tm.stage("model broadcast");
    const std::size_t n = in.n;
    std::vector<float> scores(rank == 0 ? n : 0);
    std::vector<uint8_t> ops(rank == 0 ? in.a_cat.size() + in.b_cat.size() + 1 : 1);
    std::vector<uint64_t> off(rank == 0 ? n : 0);
    std::vector<uint32_t> len(rank == 0 ? n : 0);
    const int rc = api.viterbi(comm, 0, model, n, in.a_cat.data(), in.a_off.data(), in.b_cat.data(), in.b_off.data(), scores.data(),
                               ops.data(), in.a_cat.size() + in.b_cat.size(), off.data(), len.data());
    coati_hip_model_destroy(model);
    dist_check(rc);
    // a pair whose strip hand-off timed out comes back with a NaN score (never a hang): an error here, as in the
    // single-GPU driver -- not `"score": null` next to a garbage alignment
    if(rank == 0) check_scores(scores);
    tm.stage("sharded Viterbi + gather");
    if(rank == 0) {
        write_batch_output(aln, in, scores, ops, off, len);
        tm.stage("gapped strings + output");
    }
    return true;
}

namespace {
// utils::process_alignment (src/lib/utils.cc:847-938): validate an input alignment, blank terminal
// stop codons, return the expanded CIGAR and leave the ungapped sequences in data.seqs.
std::string process_alignment(alignment_t& aln) {
    data_t& data = aln.data;
    if(data.size() != 2) throw std::invalid_argument("Exactly two sequences required.");
    if(!aln.refs.empty() || aln.rev) order_ref(data, aln.refs, aln.rev);
    const std::size_t len = data.seqs[0].length();
    if(len != data.seqs[1].length())
        throw std::invalid_argument("For alignment scoring both sequences must have equal length.");
    for(std::size_t i = 0; i < 2; ++i) {
        std::string& seq = data.seqs[i];
        const std::size_t p3 = seq.find_last_not_of('-');
        std::size_t p2 = std::string::npos, p1 = std::string::npos;
        if(p3 != std::string::npos && p3 >= 2) p2 = seq.find_last_not_of('-', p3 - 1);
        if(p2 != std::string::npos && p2 >= 1) p1 = seq.find_last_not_of('-', p2 - 1);
        if(p1 == std::string::npos) {
            data.stops.emplace_back("");
            continue;
        }
        const char cod[4] = {seq[p1], seq[p2], seq[p3], '\0'};
        if(is_stop64(cod_int(cod))) {
            data.stops.emplace_back(cod);
            seq[p1] = seq[p2] = seq[p3] = '-';
        } else {
            data.stops.emplace_back("");
        }
    }
    std::string cigar;
    cigar.reserve(len);
    for(std::size_t i = 0; i < len; ++i) {
        const char a = data.seqs[0][i], b = data.seqs[1][i];
        if(a != '-' && b != '-')
            cigar.push_back('M');
        else if(a != '-')
            cigar.push_back('D');
        else if(b != '-')
            cigar.push_back('I');
    }
    for(std::string& s : data.seqs) s.erase(std::remove(s.begin(), s.end(), '-'), s.end());
    const std::size_t len_a = data.seqs[0].length(), len_b = data.seqs[1].length();
    if(len_a % 3 != 0 || len_a % aln.gap.len != 0)
        throw std::invalid_argument("Length of reference sequence must be multiple of 3 and gap unit length.");
    if(len_b % aln.gap.len != 0)
        throw std::invalid_argument("Length of descendant sequence must be multiple of gap unit length.");
    return cigar;
}
}  // namespace

float alignment_score(alignment_t& aln, const table_t& p_marg) {
    const std::string cigar = process_alignment(aln);
    const auto enc = marginal_seq_encoding(aln.data.seqs[0], aln.data.seqs[1]);
    check_descendant_codes(enc[1]);
    const auto k = gap_log_consts(aln.gap);
    const float no_gap = k[0], gap_stop = k[1], gap_open = k[2], gap_extend = k[3];
    auto subst = [&](std::size_t pa, std::size_t pb) { return p_marg[enc[0][pa] * kTableCols + enc[1][pb]]; };
    auto pw = [](float x, std::size_t n) { return x * static_cast<float>(n); };
    // one closed gap run (align_marginal.cc:421-437,450-463); `last` adds the trailing no_gap of the both-kinds case
    auto close_gap = [&](float score, std::size_t nins, std::size_t ndel, bool last) {
        if(nins == 0) return (((score + no_gap) + gap_open) + pw(gap_extend, ndel - 1)) + gap_stop;
        if(ndel == 0) return (((score + gap_open) + pw(gap_extend, nins - 1)) + gap_stop) + no_gap;
        float s = (((score + gap_open) + gap_open) + pw(gap_extend, nins + ndel - 2)) + gap_stop;
        s = s + gap_stop;
        return last ? s + no_gap : s;
    };
    bool in_gap = false;
    float score = 0.f;
    std::size_t nins = 0, ndel = 0, apos = 0, bpos = 0;
    for(const char op : cigar) {
        if(op == 'I') {
            ++nins;
            ++bpos;
            in_gap = true;
        } else if(op == 'D') {
            ++ndel;
            ++apos;
            in_gap = true;
        } else if(!in_gap) {
            score = ((score + no_gap) + no_gap) + subst(apos, bpos);
            ++apos;
            ++bpos;
        } else {
            score = close_gap(score, nins, ndel, false);
            score = score + subst(apos, bpos);
            nins = ndel = 0;
            in_gap = false;
            ++apos;
            ++bpos;
        }
    }
    score = in_gap ? close_gap(score, nins, ndel, true) : (score + no_gap) + no_gap;
    aln.data.score = score;
    restore_end_stops(aln.data, aln.gap);
    return aln.data.score;
}

void marg_sample(alignment_t& aln, std::size_t sample_size, random_t& rand) {
    aln.data = read_input(aln.data.path);
    if(aln.data.size() != 2) throw std::invalid_argument("Exactly two sequences required.");
    std::ofstream file;
    std::ostream* out = &std::cout;
    if(!(aln.output.empty() || aln.output == "-")) {
        file.open(aln.output);
        if(!file) throw std::invalid_argument("Opening output file " + aln.output + " failed.");
        out = &file;
    }
    const std::size_t len_a = aln.seq(0).length();
    if(len_a % 3 != 0 || len_a % aln.gap.len != 0)
        throw std::invalid_argument("Length of reference sequence must be multiple of 3.");
    if(aln.seq(1).length() % aln.gap.len != 0)
        throw std::invalid_argument("Length of descendant sequence must be multiple of " + std::to_string(aln.gap.len) + ".");
    trim_end_stops(aln.data);
    const std::string anc = aln.seq(0), des = aln.seq(1);
    const auto seq_pair = marginal_seq_encoding(anc, des);
    check_descendant_codes(seq_pair[1]);
    set_subst(aln);
    align_pair_work_t work;
    forward(work, seq_pair[0], seq_pair[1], aln);
    if(sample_size == 0) return;
    // all samples in one device call: the walker draws them one after the other from `rand`'s
    // stream, exactly as the loop of align_marginal.cc:590-593 does
    const uint64_t state_in[2] = {rand.lo(), rand.hi()};
    uint64_t state_out[2] = {0, 0};
    const std::size_t width = anc.size() + des.size();
    std::vector<uint8_t> ops(sample_size * width + 1);
    std::vector<uint64_t> off(sample_size);
    std::vector<uint32_t> len(sample_size);
    std::vector<float> lw(sample_size);
    hip_check(coati_hip_sampleback(work.batch, static_cast<uint32_t>(sample_size), state_in,
                                   aln.independent_streams ? 1 : 0, lw.data(), ops.data(), sample_size * width, off.data(),
                                   len.data(), state_out));
    if(!aln.independent_streams) rand.set_state(state_out[0], state_out[1]);
    for(std::size_t i = 0; i < sample_size; ++i) {
        aln.data.seqs.assign(2, std::string());
        ops_to_alignment(ops.data() + off[i], len[i], anc, des, aln.data.seqs[0], aln.data.seqs[1]);
        aln.data.score = lw[i];
        restore_end_stops(aln.data, aln.gap);
        write_json(aln.data, *out, i, sample_size);
    }
}

}  // namespace coati_amd
