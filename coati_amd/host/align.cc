#include "align.hpp"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <exception>
#include <fstream>
#include <future>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

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
    if(aln.fast_forward) hip_check(coati_hip_model_set_option(m, COATI_HIP_OPT_FORWARD_MODE, COATI_HIP_FORWARD_TOLERANCE));
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
    // (at most 48: the block pipeline runs two or three of these loops at once on a 256-thread host)
    const std::size_t n_threads = std::min<std::size_t>({want, std::max(1u, std::thread::hardware_concurrency()), 48});
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
// ---------------------------------------------------------------------------------------------------------
// --batch: the reference aligns ONE pair per process (utils.cc:809-812); here a file of 2n sequences is n pairs.
// The driver is a three-stage pipeline over BLOCKS of pairs, so that a file of any size streams through bounded
// memory and the host work either side of the GPU overlaps it:
//   A  parse + process_marginal + encode   (block k+1, on the thread pool)
//   B  coati_hip_viterbi_batch             (block k,   this thread; the library overlaps H2D / kernel / D2H inside)
//   C  gapped strings + JSON text + write  (block k-1, on the thread pool; blocks leave in order)
// The HIP runtime and the model are brought up on a helper thread while block 0 is parsed.
// Input semantics are read_fasta's / process_marginal's / write_json's, byte for byte (tests: the CLI suite
// compares this driver's output with per-pair runs).
// ---------------------------------------------------------------------------------------------------------
constexpr std::size_t kBatchBlockPairs = 32768;

// A FASTA file in memory and where its records start ('>' at the beginning of a line; io.cc:read_fasta).  The file is
// MAPPED, not read: every rank of a multi-GPU run indexes the whole file (the index is the shard plan) but touches the
// sequence bytes of its own shard only, and the single-GPU driver saves the copy of a multi-GB input.
struct fasta_index_t {
    const char* data{nullptr};
    std::size_t size{0};
    std::vector<std::size_t> starts;  // offsets of the '>' of every record, then size
    fasta_index_t() = default;
    fasta_index_t(const fasta_index_t&) = delete;
    fasta_index_t& operator=(const fasta_index_t&) = delete;
    ~fasta_index_t() {
        if(data != nullptr && size != 0) munmap(const_cast<char*>(data), size);
    }
};
// false: not a plain FASTA file (stdin, "fmt:path" of another format, .phy, .json): the generic reader takes it
bool load_fasta(const std::string& path, fasta_index_t& fi) {
    const file_type_t type = path.empty() ? file_type_t{"-", ".json"} : extract_file_type(path);
    if(type.path.empty() || type.path == "-" || !(type.type_ext == ".fa" || type.type_ext == ".fasta")) return false;
    const int fd = open(type.path.c_str(), O_RDONLY | O_CLOEXEC);
    if(fd < 0) throw std::invalid_argument("Opening input file " + path + " failed.");
    struct stat st;
    if(fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {  // (not a regular file: a pipe, a device)
        close(fd);
        return false;
    }
    fi.size = static_cast<std::size_t>(st.st_size);
    if(fi.size != 0) {
        void* m = mmap(nullptr, fi.size, PROT_READ, MAP_PRIVATE, fd, 0);
        if(m == MAP_FAILED) {
            close(fd);
            fi.size = 0;
            throw std::invalid_argument("Reading input file " + path + " failed.");
        }
        fi.data = static_cast<const char*>(m);
        (void)madvise(m, fi.size, MADV_WILLNEED);
    }
    close(fd);
    // record starts, found in parallel: every thread scans a slice for "\n>" (and the file's first byte)
    const std::size_t n = fi.size;
    const std::size_t slices = std::max<std::size_t>(1, std::min<std::size_t>(64, n >> 20));
    std::vector<std::vector<std::size_t>> found(slices);
    parallel_for(slices, 1, [&](std::size_t k) {
        const std::size_t lo = n * k / slices, hi = n * (k + 1) / slices;
        const char* base = fi.data;
        for(const char* p = base + lo; p < base + hi;) {
            p = static_cast<const char*>(std::memchr(p, '>', static_cast<std::size_t>(base + hi - p)));
            if(p == nullptr) break;
            if(p == base || p[-1] == '\n') found[k].push_back(static_cast<std::size_t>(p - base));
            ++p;
        }
    });
    for(const auto& f : found) fi.starts.insert(fi.starts.end(), f.begin(), f.end());
    fi.starts.push_back(n);
    return true;
}
// record `rec` as read_fasta reads it: the name is the rest of the '>' line (a trailing '\r' dropped), the sequence
// the following lines without white space; empty lines and lines starting with ';' are skipped
void fasta_record(const fasta_index_t& fi, std::size_t rec, std::string& name, std::string& seq) {
    const char* p = fi.data + fi.starts[rec] + 1;
    const char* const end = fi.data + fi.starts[rec + 1];
    const char* eol = static_cast<const char*>(std::memchr(p, '\n', static_cast<std::size_t>(end - p)));
    if(eol == nullptr) eol = end;
    name.assign(p, eol);
    if(!name.empty() && name.back() == '\r') name.pop_back();
    if(name.empty()) throw std::invalid_argument("Input fasta file contains a sequence without a name.");
    seq.clear();
    seq.reserve(static_cast<std::size_t>(end - eol));
    for(p = eol < end ? eol + 1 : end; p < end;) {
        eol = static_cast<const char*>(std::memchr(p, '\n', static_cast<std::size_t>(end - p)));
        if(eol == nullptr) eol = end;
        if(p != eol && *p != ';')
            for(const char* q = p; q < eol; ++q)
                if(std::isspace(static_cast<unsigned char>(*q)) == 0) seq.push_back(*q);
        p = eol < end ? eol + 1 : end;
    }
}

// One block of pairs on its way through the pipeline.
struct batch_block_t {
    std::size_t p0{0}, n{0};
    std::vector<data_t> pairs;  // processed (stops trimmed and remembered); seqs[0] = ancestor, seqs[1] = descendant
    std::vector<uint64_t> a_off, b_off;
    std::vector<unsigned char> a_cat, b_cat;
    std::vector<float> scores;
    std::vector<uint8_t> ops;
    std::vector<uint64_t> off;
    std::vector<uint32_t> len;
};
// Where the pairs come from: the indexed FASTA file, or (other formats, stdin) everything the generic reader returned.
struct batch_source_t {
    bool fast{false};
    fasta_index_t fasta;
    data_t all;
    std::size_t n_pairs{0};
};
std::unique_ptr<batch_source_t> open_batch_source(const alignment_t& aln) {
    auto src = std::make_unique<batch_source_t>();
    src->fast = load_fasta(aln.data.path, src->fasta);
    const std::size_t n_seqs = src->fast ? src->fasta.starts.size() - 1 : (src->all = read_input(aln.data.path)).size();
    if(n_seqs == 0 || n_seqs % 2 != 0) throw std::invalid_argument("Batch input needs an even number of sequences.");
    src->n_pairs = n_seqs / 2;
    return src;
}
// What decides the shards of a multi-GPU run: per pair the sizes of its two records (names and line breaks
// included -- an estimate of len_a x len_b that needs no parsing; every rank derives the same bounds from the same
// file).  Offsets as coati_hip_shard_bounds takes them.
void batch_source_weights(const batch_source_t& src, std::vector<uint64_t>& a_off, std::vector<uint64_t>& b_off) {
    a_off.assign(src.n_pairs + 1, 0), b_off.assign(src.n_pairs + 1, 0);
    for(std::size_t p = 0; p < src.n_pairs; ++p) {
        const uint64_t wa = src.fast ? src.fasta.starts[2 * p + 1] - src.fasta.starts[2 * p] : src.all.seqs[2 * p].size();
        const uint64_t wb = src.fast ? src.fasta.starts[2 * p + 2] - src.fasta.starts[2 * p + 1] : src.all.seqs[2 * p + 1].size();
        a_off[p + 1] = a_off[p] + wa, b_off[p + 1] = b_off[p] + wb;
    }
}
// stage A, first half: the pairs [p0, p0 + n) parsed and processed; offsets filled in
void batch_block_parse(const alignment_t& aln, const batch_source_t& src, std::size_t p0, std::size_t n, batch_block_t& blk) {
    blk.p0 = p0;
    blk.n = n;
    blk.pairs.assign(n, data_t());
    parallel_for(n, 64, [&](std::size_t i) {
        data_t& d = blk.pairs[i];
        d.names.assign(2, std::string());
        d.seqs.assign(2, std::string());
        for(std::size_t s = 0; s < 2; ++s) {
            const std::size_t rec = 2 * (p0 + i) + s;
            if(src.fast) {
                fasta_record(src.fasta, rec, d.names[s], d.seqs[s]);
            } else {
                d.names[s] = src.all.names[rec];
                d.seqs[s] = src.all.seqs[rec];
            }
        }
        process_marginal(d, aln.gap, std::string(), aln.rev);
    });
    blk.a_off.assign(n + 1, 0), blk.b_off.assign(n + 1, 0);
    for(std::size_t i = 0; i < n; ++i) {
        blk.a_off[i + 1] = blk.a_off[i] + blk.pairs[i].seqs[0].size();
        blk.b_off[i + 1] = blk.b_off[i] + blk.pairs[i].seqs[1].size();
    }
}
// stage A, second half: the pairs [i0, i1) of the block encoded (a rank of a multi-GPU run encodes its shard only)
void batch_block_encode(batch_block_t& blk, std::size_t i0, std::size_t i1) {
    // (a_cat / b_cat hold the bytes [a_off[i0], a_off[i1]) / [b_off[i0], b_off[i1]) of the block's concatenations)
    const uint64_t a0 = blk.a_off[i0], b0 = blk.b_off[i0];
    blk.a_cat.resize(blk.a_off[i1] - a0), blk.b_cat.resize(blk.b_off[i1] - b0);
    parallel_for(i1 - i0, 64, [&](std::size_t k) {
        const std::size_t i = i0 + k;
        encode_ancestor(blk.pairs[i].seqs[0], blk.a_cat.data() + (blk.a_off[i] - a0));
        unsigned char* des = blk.b_cat.data() + (blk.b_off[i] - b0);
        const std::string& d = blk.pairs[i].seqs[1];
        encode_descendant(d, des);
        for(std::size_t c = 0; c < d.size(); ++c)
            if(des[c] >= kTableCols) throw std::invalid_argument("Invalid character in descendant sequence.");
    });
}
// stage C: the block's alignments as JSON text (write_json's bytes for elements p0 .. p0+n of an array of n_total), in order
void batch_block_write(const alignment_t& aln, batch_block_t& blk, std::size_t n_total, std::ostream& out) {
    constexpr std::size_t kSub = 128;  // pairs per text piece
    const std::size_t pieces = (blk.n + kSub - 1) / kSub;
    std::vector<std::string> text(pieces);
    parallel_for(pieces, 1, [&](std::size_t k) {
        std::ostringstream os;
        for(std::size_t i = k * kSub; i < std::min(blk.n, (k + 1) * kSub); ++i) {
            data_t& d = blk.pairs[i];
            const std::string anc = std::move(d.seqs[0]), des = std::move(d.seqs[1]);
            d.seqs.assign(2, std::string());
            ops_to_alignment(blk.ops.data() + blk.off[i], blk.len[i], anc, des, d.seqs[0], d.seqs[1]);
            d.score = blk.scores[i];
            restore_end_stops(d, aln.gap);
            write_json(d, os, blk.p0 + i, n_total);
            d = data_t();  // (a block's strings are not kept once they are text)
        }
        text[k] = os.str();
    });
    for(const std::string& t : text) out.write(t.data(), static_cast<std::streamsize>(t.size()));
}
void check_block_scores(const batch_block_t& blk) {
    for(std::size_t i = 0; i < blk.n; ++i)
        if(std::isnan(blk.scores[i]))
            throw std::runtime_error("The device did not finish pair " + std::to_string(blk.p0 + i) + " (strip hand-off timed out).");
}
bool g_fast_exit = false;
}  // namespace

void set_process_exits_after_call(bool on) { g_fast_exit = on; }

// Test hook: the block driver's reader against the generic one (read_input) on the same file.  0 = the same names
// and sequences in the same order; k + 1 = the first difference is record k; -1 = the fast reader does not take the file.
long batch_reader_first_difference(const std::string& path) {
    fasta_index_t fi;
    if(!load_fasta(path, fi)) return -1;
    const data_t all = read_input(path);
    const std::size_t n = fi.starts.size() - 1;
    std::string name, seq;
    for(std::size_t r = 0; r < std::max(n, all.size()); ++r) {
        if(r >= n || r >= all.size()) return static_cast<long>(r) + 1;
        fasta_record(fi, r, name, seq);
        if(name != all.names[r] || seq != all.seqs[r]) return static_cast<long>(r) + 1;
    }
    return 0;
}

namespace {
// The block pipeline over the pairs [p_begin, p_end) of `src`: the JSON text of elements p_begin .. p_end of an array of
// n_total leaves through `out`, blocks in order.  `model` is asked for when the first block is ready for the device
// (the single-GPU driver brings the HIP runtime up on a helper thread meanwhile).
template <typename GetModel>
void run_batch_pipeline(const alignment_t& aln, const batch_source_t& src, std::size_t p_begin, std::size_t p_end, std::size_t n_total,
                        GetModel&& model, std::ostream& out, host_timer& tm) {
    const std::size_t n_mine = p_end - p_begin;
    const std::size_t n_blocks = (n_mine + kBatchBlockPairs - 1) / kBatchBlockPairs;
    if(n_blocks == 0) return;
    auto stage_a = [&](std::size_t k) {
        auto blk = std::make_unique<batch_block_t>();
        const std::size_t p0 = p_begin + k * kBatchBlockPairs;
        batch_block_parse(aln, src, p0, std::min(kBatchBlockPairs, p_end - p0), *blk);
        batch_block_encode(*blk, 0, blk->n);
        return blk;
    };
    std::future<std::unique_ptr<batch_block_t>> next = std::async(std::launch::async, stage_a, std::size_t{0});
    std::future<void> writing;
    coati_hip_model* m = nullptr;
    try {
        for(std::size_t k = 0; k < n_blocks; ++k) {
            std::unique_ptr<batch_block_t> blk = next.get();
            if(k + 1 < n_blocks) next = std::async(std::launch::async, stage_a, k + 1);
            if(k == 0) {
                tm.stage("parse + encode (first block)");
                m = model();
                tm.stage("model (overlapped with the input)");
            }
            batch_block_t& b = *blk;
            b.scores.resize(b.n), b.off.resize(b.n), b.len.resize(b.n);
            b.ops.resize(b.a_cat.size() + b.b_cat.size() + 1);
            hip_check(coati_hip_viterbi_batch(m, b.n, b.a_cat.data(), b.a_off.data(), b.b_cat.data(), b.b_off.data(), b.scores.data(), b.ops.data(),
                                              b.a_cat.size() + b.b_cat.size(), b.off.data(), b.len.data()));
            check_block_scores(b);
            if(k == 0) tm.stage("device (first block: workspaces, upload, kernels, download)");
            if(writing.valid()) writing.get();  // (blocks leave in order; a failed write of the previous block stops the run here)
            std::shared_ptr<batch_block_t> held = std::move(blk);
            writing = std::async(std::launch::async, [&aln, held, n_total, &out]() {
                batch_block_write(aln, *held, n_total, out);
                if(!out) throw std::runtime_error("Writing output failed.");
            });
        }
        if(writing.valid()) writing.get();
    } catch(...) {
        // (the helper threads hold references to this frame: let them finish before it unwinds)
        if(next.valid()) next.wait();
        if(writing.valid()) writing.wait();
        throw;
    }
    out.flush();
    if(!out) throw std::runtime_error("Writing output failed.");
    tm.stage(n_blocks > 1 ? "remaining blocks (parse, device, output overlapped)" : "gapped strings + output");
}
}  // namespace

// Test hook for the multi-GPU driver's host side (no device needed): rank `rank` of `world` plans its shard from the
// index of `path` and parses + encodes ONLY that shard through the block pipeline's stage A; the result is compared,
// pair by pair, with the generic path (read_input of the whole file -> process_marginal -> marginal_seq_encoding).
// Returns 0 if names, processed sequences, stop codons and codes are identical, k + 1 if pair k of the input differs;
// *s0 / *s1 = the shard.  Every rank derives the same bounds: the caller checks that the shards tile the input.
long batch_shard_first_difference(const alignment_t& aln_in, int world, int rank, uint64_t* s0_out, uint64_t* s1_out) {
    alignment_t aln = aln_in;
    const std::unique_ptr<batch_source_t> src = open_batch_source(aln);
    std::vector<uint64_t> wa, wb, bounds(static_cast<std::size_t>(world) + 1, 0);
    batch_source_weights(*src, wa, wb);
    hip_check(coati_hip_shard_bounds(src->n_pairs, wa.data(), wb.data(), world, bounds.data()));
    const std::size_t s0 = bounds[static_cast<std::size_t>(rank)], s1 = bounds[static_cast<std::size_t>(rank) + 1];
    if(s0_out != nullptr) *s0_out = s0;
    if(s1_out != nullptr) *s1_out = s1;
    const data_t all = read_input(aln.data.path);
    // small blocks, so that a shard is several of them
    constexpr std::size_t kBlock = 7;
    for(std::size_t p0 = s0; p0 < s1; p0 += kBlock) {
        batch_block_t blk;
        batch_block_parse(aln, *src, p0, std::min(kBlock, s1 - p0), blk);
        batch_block_encode(blk, 0, blk.n);
        for(std::size_t i = 0; i < blk.n; ++i) {
            const std::size_t p = p0 + i;
            data_t d;
            d.names = {all.names[2 * p], all.names[2 * p + 1]};
            d.seqs = {all.seqs[2 * p], all.seqs[2 * p + 1]};
            process_marginal(d, aln.gap, std::string(), aln.rev);
            const auto enc = marginal_seq_encoding(d.seqs[0], d.seqs[1]);
            const data_t& g = blk.pairs[i];
            bool same = g.names == d.names && g.seqs == d.seqs && g.stops == d.stops;
            same = same && blk.a_off[i + 1] - blk.a_off[i] == enc[0].size() && blk.b_off[i + 1] - blk.b_off[i] == enc[1].size();
            same = same && std::equal(enc[0].begin(), enc[0].end(), blk.a_cat.begin() + static_cast<std::ptrdiff_t>(blk.a_off[i]));
            same = same && std::equal(enc[1].begin(), enc[1].end(), blk.b_cat.begin() + static_cast<std::ptrdiff_t>(blk.b_off[i]));
            if(!same) return static_cast<long>(p) + 1;
        }
    }
    return 0;
}

bool marg_alignment_batch(alignment_t& aln) {
    host_timer tm("alignpair --batch");
    set_subst(aln);
    // the HIP runtime, the model and the pipeline's workspaces (coati_hip_model_prepare, sized from the input's index:
    // pairs and the mean record size) come up on a helper thread while the input is read, parsed and encoded
    std::promise<std::array<uint64_t, 3>> hint_promise;
    std::shared_future<std::array<uint64_t, 3>> hint = hint_promise.get_future().share();
    auto model_ready = std::async(std::launch::async, [&aln, hint]() {
        coati_hip_model* m = make_model(aln);
        try {
            const std::array<uint64_t, 3> h = hint.get();  // (pairs, len_a, len_b; an exception here = the input failed: nothing to prepare)
            (void)coati_hip_model_prepare(m, h[0], h[1], h[2]);
        } catch(...) {
        }
        return m;
    });
    struct model_guard {
        std::future<coati_hip_model*>& f;
        coati_hip_model* m{nullptr};
        ~model_guard() {
            if(m == nullptr && f.valid()) {
                try {
                    m = f.get();
                } catch(...) {
                }
            }
            // (a process that exits right after this call leaves the GBs of cached workspaces to the driver: freeing
            // them one by one costs the one-shot tool ~65 ms)
            if(m != nullptr && !g_fast_exit) coati_hip_model_destroy(m);
        }
    } guard{model_ready};
    // (declared AFTER model_guard, so destroyed BEFORE it: on unwind -- missing file, odd number of records -- the helper
    // thread must be released from hint.get() before ~model_guard waits for it; the other order deadlocks on a GPU box)
    struct hint_guard {  // (whatever happens below, the helper thread is not left waiting)
        std::promise<std::array<uint64_t, 3>>& p;
        bool set{false};
        ~hint_guard() {
            if(!set) p.set_exception(std::make_exception_ptr(std::runtime_error("no input")));
        }
    } hguard{hint_promise};
    const std::unique_ptr<batch_source_t> src = open_batch_source(aln);
    {
        // mean sequence length from the record sizes (names and line breaks make it a slight overestimate: sizes things only)
        std::vector<uint64_t> wa, wb;
        batch_source_weights(*src, wa, wb);
        const uint64_t n = std::max<uint64_t>(src->n_pairs, 1);
        uint64_t la = 0;  // the LONGEST ancestor record sizes the wavefronts' checkpoint slots (a slot too small is re-allocated by the call)
        for(std::size_t p = 0; p < src->n_pairs; ++p) la = std::max(la, wa[p + 1] - wa[p]);
        const uint64_t lb = wb[src->n_pairs] / n;
        hint_promise.set_value({static_cast<uint64_t>(src->n_pairs), la - la % 3, lb});
        hguard.set = true;
    }
    tm.stage("read + index");
    std::ofstream file;
    std::ostream* out = &std::cout;
    if(!(aln.output.empty() || aln.output == "-")) {
        file.open(extract_file_type(aln.output).path);
        if(!file) throw std::invalid_argument("Opening output file " + aln.output + " failed.");
        out = &file;
    }
    run_batch_pipeline(aln, *src, 0, src->n_pairs, src->n_pairs, [&]() { return guard.m = model_ready.get(); }, *out, tm);
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
    int (*allreduce_f64)(void*, int, double*, uint32_t){nullptr};
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
    api.allreduce_f64 = reinterpret_cast<decltype(api.allreduce_f64)>(sym("coati_hip_dist_allreduce_f64"));
    return api;
}
// `bytes` bytes of file `from` appended into `to_fd` at offset `at` (copy_file_range: the kernel moves the pages; plain
// read / write where that is not available, e.g. across file systems)
void copy_into(const std::string& from, int to_fd, uint64_t at, uint64_t bytes) {
    const int in = open(from.c_str(), O_RDONLY | O_CLOEXEC);
    if(in < 0) throw std::runtime_error("--devices: cannot reopen " + from);
    struct closer {
        int fd;
        ~closer() { close(fd); }
    } guard{in};
    loff_t off_in = 0, off_out = static_cast<loff_t>(at);
    uint64_t left = bytes;
    while(left > 0) {
        const ssize_t moved = copy_file_range(in, &off_in, to_fd, &off_out, static_cast<std::size_t>(std::min<uint64_t>(left, 1u << 30)), 0);
        if(moved > 0) {
            left -= static_cast<uint64_t>(moved);
            continue;
        }
        // fall back: read + pwrite from where the kernel copy stopped
        std::vector<char> buf(8u << 20);
        while(left > 0) {
            const ssize_t got = pread(in, buf.data(), static_cast<std::size_t>(std::min<uint64_t>(left, buf.size())), off_in);
            if(got <= 0) throw std::runtime_error("--devices: reading " + from + " failed");
            for(ssize_t done = 0; done < got;) {
                const ssize_t put = pwrite(to_fd, buf.data() + done, static_cast<std::size_t>(got - done), off_out + done);
                if(put <= 0) throw std::runtime_error("Writing output failed.");
                done += put;
            }
            off_in += got, off_out += got, left -= static_cast<uint64_t>(got);
        }
    }
}
}  // namespace

// One rank of `coati-alignpair --batch --devices ...` (one process per GPU).  Nothing of the data path crosses a link:
//   * every rank maps and indexes the input (the record sizes are the shard plan: coati_hip_shard_bounds on them, the
//     same bounds everywhere) but parses, encodes, aligns and FORMATS only its own shard, through the same three-stage
//     block pipeline as the single-GPU driver;
//   * the model is computed on rank 0 and broadcast over RCCL (every rank scores with the same bits);
//   * rank 0 streams its slice of the JSON array straight into the output; the other ranks stream theirs into part
//     files, the slice sizes are exchanged (one small all-reduce), and every rank moves its part to its place in the
//     output (parallel kernel-side copies).  With the output on stdout rank 0 forwards the parts in order.
// What the reference does for one pair per process: src/lib/align_marginal.cc:44-88, src/lib/utils.cc:809-812.
bool marg_alignment_batch_dist(alignment_t& aln, int rank, int world, const std::string& id_file) {
    if(world < 1 || world > 64 || rank < 0 || rank >= world) throw std::invalid_argument("--devices: bad rank / world (at most 64 devices)");
    host_timer tm(rank == 0 ? "alignpair --batch --devices (rank 0)" : "alignpair --batch --devices");
    const dist_api_t api = load_dist_api();
    auto dist_check = [&](int rc) {
        if(rc != 0) throw std::runtime_error(api.last_error());
    };
    // ---- rendezvous + communicator on a helper thread (RCCL's bring-up takes longer than indexing the input)
    auto comm_ready = std::async(std::launch::async, [&]() -> void* {
        unsigned char id[128];
        if(rank == 0) {  // rank 0 leaves the id in a file (written under another name, then renamed)
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
        // RCCL prints its version banner on stdout when a communicator is created; rank 0's stdout is the JSON stream
        // when no -o is given: send fd 1 to stderr for the duration (nothing else writes to stdout before the pipeline)
        struct stdout_to_stderr {
            int saved;
            stdout_to_stderr() {
                std::cout.flush();
                std::fflush(stdout);
                saved = dup(STDOUT_FILENO);
                if(saved >= 0) dup2(STDERR_FILENO, STDOUT_FILENO);
            }
            ~stdout_to_stderr() {
                std::fflush(stdout);
                if(saved >= 0) {
                    dup2(saved, STDOUT_FILENO);
                    close(saved);
                }
            }
        } quiet;
        dist_check(api.init(id, world, rank, aln.device, &comm));
        return comm;
    });
    struct comm_guard {
        const dist_api_t& api;
        std::future<void*>& f;
        void* c{nullptr};
        ~comm_guard() {
            if(c == nullptr && f.valid()) {
                try {
                    c = f.get();
                } catch(...) {
                }
            }
            if(c != nullptr) api.destroy(c);
        }
    } guard{api, comm_ready};
    // ---- the shard plan from the index alone
    const std::unique_ptr<batch_source_t> src = open_batch_source(aln);
    const std::size_t n_total = src->n_pairs;
    std::vector<uint64_t> wa, wb, bounds(static_cast<std::size_t>(world) + 1, 0);
    batch_source_weights(*src, wa, wb);
    hip_check(coati_hip_shard_bounds(n_total, wa.data(), wb.data(), world, bounds.data()));
    const std::size_t s0 = bounds[static_cast<std::size_t>(rank)], s1 = bounds[static_cast<std::size_t>(rank) + 1];
    tm.stage("index + shard plan");
    void* comm = guard.c = comm_ready.get();
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
    struct model_guard {
        coati_hip_model* m;
        ~model_guard() {
            if(!g_fast_exit) coati_hip_model_destroy(m);
        }
    } mguard{model};
    tm.stage("model broadcast");
    // ---- this rank's slice of the output
    // -o: the result is ASSEMBLED in <output>.tmp and renamed onto the output path by rank 0 only after every rank has
    // said that its slice is in place -- a run that fails half way (a bad pair on some rank, a NaN score) leaves no
    // truncated JSON under the user's name.  Rank 0 streams its slice (offset 0) straight into <output>.tmp, the others
    // into <output>.partN, which they copy to their place and remove.  What a rank leaves behind when it is STOPPED
    // (the launcher's SIGTERM after another rank failed) the launcher removes (coati_alignpair.cc: launch_ranks).
    const bool to_stdout = aln.output.empty() || aln.output == "-";
    const std::string final_path = to_stdout ? std::string() : extract_file_type(aln.output).path;
    const std::string tmp_path = final_path + ".tmp";
    const std::string part_path = (to_stdout ? id_file : final_path) + ".part" + std::to_string(rank);
    struct scratch_guard {  // (an exception on this rank: its own scratch files go)
        std::vector<std::string> paths;
        ~scratch_guard() {
            for(const std::string& p : paths) std::remove(p.c_str());
        }
    } scratch;
    std::ofstream file;
    std::ostream* out = &std::cout;
    if(rank != 0 || !to_stdout) {
        const std::string& path = rank == 0 ? tmp_path : part_path;
        file.open(path, std::ios::binary | std::ios::trunc);
        if(!file) throw std::invalid_argument("Opening output file " + (rank == 0 ? final_path : path) + " failed.");
        scratch.paths.push_back(path);
        out = &file;
    }
    run_batch_pipeline(aln, *src, s0, s1, n_total, [&]() { return model; }, *out, tm);
    const uint64_t my_bytes = out == &std::cout ? 0 : static_cast<uint64_t>(file.tellp());
    if(file.is_open()) {
        file.close();
        if(!file) throw std::runtime_error("Writing output failed.");
    }
    // ---- slice sizes of all ranks (exact in a double below 2^53), then every part to its place
    std::vector<double> sizes(static_cast<std::size_t>(world), 0.0);
    sizes[static_cast<std::size_t>(rank)] = static_cast<double>(my_bytes);
    dist_check(api.allreduce_f64(comm, 0, sizes.data(), static_cast<uint32_t>(world)));
    tm.stage("slice sizes exchanged");
    if(to_stdout) {
        if(rank == 0) {
            std::vector<char> buf(8u << 20);
            for(int r = 1; r < world; ++r) {
                const std::string from = id_file + ".part" + std::to_string(r);
                std::ifstream in(from, std::ios::binary);
                if(!in) throw std::runtime_error("--devices: cannot read " + from);
                while(in) {
                    in.read(buf.data(), static_cast<std::streamsize>(buf.size()));
                    std::cout.write(buf.data(), in.gcount());
                }
                std::remove(from.c_str());
            }
            std::cout.flush();
            if(!std::cout) throw std::runtime_error("Writing output failed.");
        }
    } else if(rank != 0) {
        uint64_t at = 0;
        for(int r = 0; r < rank; ++r) at += static_cast<uint64_t>(sizes[static_cast<std::size_t>(r)]);
        const int fd = open(tmp_path.c_str(), O_WRONLY | O_CLOEXEC);  // (rank 0 created it before its pipeline started)
        if(fd < 0) throw std::runtime_error("Opening output file " + final_path + " failed.");
        try {
            copy_into(part_path, fd, at, my_bytes);
        } catch(...) {
            close(fd);
            throw;
        }
        if(close(fd) != 0) throw std::runtime_error("Writing output failed.");
    }
    // (all parts are in place when every rank has left this collective)
    double done = 1.0;
    dist_check(api.allreduce_f64(comm, 0, &done, 1));
    if(!to_stdout && rank == 0) {
        if(std::rename(tmp_path.c_str(), final_path.c_str()) != 0) throw std::runtime_error("Writing output failed.");
        scratch.paths.clear();  // (it IS the output now)
    }
    tm.stage("output assembled");
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
    // (the sampler's device blocks and page-locked records while the Forward kernel runs: this process makes ONE sampleback call)
    hip_check(coati_hip_sampleback_prepare(work.batch, static_cast<uint32_t>(sample_size), aln.independent_streams ? 1 : 0));
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
