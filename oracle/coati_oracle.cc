// TEST INFRASTRUCTURE ONLY -- see coati_oracle.h for the rules and the parity
// status ("PINNED").  Plain scalar C++; every routine names the reference lines
// it restates.  Deliberately keeps the reference's data layout (dense row-major
// fp32 matrices, filled with `lowest` first) so that it doubles as the timed CPU
// baseline ("port") of bench.py.
#include "coati_oracle.h"

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <string_view>
#include <thread>
#include <vector>

namespace {

constexpr float kLowest = std::numeric_limits<float>::lowest();  // semiring.hpp:83,113
constexpr int kTableCols = 15;                                   // mutation_coati.cc:171

struct Consts {
    float no_gap, gap_stop, gap_open, gap_extend;
    explicit Consts(const float c[4]) : no_gap(c[0]), gap_stop(c[1]), gap_open(c[2]), gap_extend(c[3]) {}
};

// utils.hpp:134-146 (float overload)
inline float log1p_exp(float x) {
    if(x <= -16.0f) return ::expf(x);
    if(x <= 8.0f) return ::log1pf(::expf(x));
    if(x <= 14.5f) return x + ::expf(-x);
    return x;
}
// utils.hpp:152-156
inline float log_sum_exp(float p, float q) {
    const float hi = std::max(p, q);
    const float gap = -std::fabs(p - q);
    return hi + log1p_exp(gap);
}

struct Tropical {  // semiring.hpp:94-121
    static float plus(float p, float q) { return std::max(p, q); }
};
struct Log {  // semiring.hpp:63-92
    static float plus(float p, float q) { return log_sum_exp(p, q); }
};

// Both semirings: times = left-to-right fp32 addition, power(x, n) = x * float(n)
// (semiring.hpp:74-80,104-110).
inline float pw(float x, uint64_t n) { return x * static_cast<float>(n); }

enum Edge { MM = 0, MD, MI, DM, DD, IM, ID, II };  // align_pair.hpp:94-103 order

template <class S>
void fill(const float* table, const Consts& k, uint64_t L, const uint8_t* a, uint64_t la,
          const uint8_t* b, uint64_t lb, float* M, float* D, float* I, float* edges) {
    const uint64_t rows = la + L, cols = lb + L, start = L - 1, n = rows * cols;
    // work.resize(...) -- every matrix fully written with `lowest` (matrix.hpp:94-100)
    std::fill(M, M + n, kLowest);
    std::fill(D, D + n, kLowest);
    std::fill(I, I + n, kLowest);
    if(edges != nullptr) std::fill(edges, edges + 8 * n, kLowest);
    auto at = [cols](uint64_t i, uint64_t j) { return i * cols + j; };

    // margins, align_pair.cc:82-91
    M[at(start, start)] = 0.0f;
    for(uint64_t i = start + L; i < rows; i += L)
        D[at(i, start)] = (k.no_gap + k.gap_open) + pw(k.gap_extend, i - 1);
    for(uint64_t j = start + L; j < cols; j += L)
        I[at(start, j)] = k.gap_open + pw(k.gap_extend, j - 1);
    if(edges != nullptr) {  // init_margins, align_pair.hpp:108-111
        std::memcpy(edges + DD * n, D, n * sizeof(float));
        std::memcpy(edges + II * n, I, n * sizeof(float));
    }

    const float ext_lm1 = pw(k.gap_extend, L - 1), ext_l = pw(k.gap_extend, L);
    // body, align_pair.cc:94-129
    for(uint64_t i = L; i < rows; ++i) {
        const float* trow = table + static_cast<size_t>(a[i - L]) * kTableCols;
        for(uint64_t j = L; j < cols; ++j) {
            const float s = trow[b[j - L]];
            const uint64_t dg = at(i - 1, j - 1), up = at(i - L, j), lf = at(i, j - L), c = at(i, j);
            const float m2m = ((M[dg] + k.no_gap) + k.no_gap) + s;
            const float d2m = (D[dg] + k.gap_stop) + s;
            const float i2m = ((I[dg] + k.gap_stop) + k.no_gap) + s;
            const float m2d = ((M[up] + k.no_gap) + k.gap_open) + ext_lm1;
            const float i2d = ((I[up] + k.gap_stop) + k.gap_open) + ext_lm1;
            const float d2d = D[up] + ext_l;
            const float m2i = (M[lf] + k.gap_open) + ext_lm1;
            const float i2i = I[lf] + ext_l;
            M[c] = S::plus(S::plus(m2m, d2m), i2m);
            D[c] = S::plus(S::plus(m2d, d2d), i2d);
            I[c] = S::plus(m2i, i2i);
            if(edges != nullptr) {
                edges[MM * n + c] = m2m;
                edges[MD * n + c] = m2d;
                edges[MI * n + c] = m2i;
                edges[DM * n + c] = d2m;
                edges[DD * n + c] = d2d;
                edges[IM * n + c] = i2m;
                edges[ID * n + c] = i2d;
                edges[II * n + c] = i2i;
            }
        }
    }
    // terminal state, align_pair.cc:130-138
    const uint64_t e = at(rows - 1, cols - 1);
    M[e] = (M[e] + k.no_gap) + k.no_gap;
    I[e] = (I[e] + k.gap_stop) + k.no_gap;
    D[e] = D[e] + k.gap_stop;
}

// max_mdi / max_mi, align_pair.cc:210-232.  Ties: M over D over I; I over M.
inline int pick3(float m, float d, float i) {
    int w = ORACLE_OP_M;
    float best = m;
    if(d > best) {
        best = d;
        w = ORACLE_OP_D;
    }
    if(i > best) return ORACLE_OP_I;
    return w;
}
inline int pick2(float m, float i) { return m > i ? ORACLE_OP_M : ORACLE_OP_I; }

// The three "which state next" rules of the reference traceback
// (align_pair.cc:275-296), as functions of the arrived-at cell's M/D/I.
inline int after_match(const Consts& k, float m, float d, float i) {
    return pick3((m + k.no_gap) + k.no_gap, d + k.gap_stop, (i + k.gap_stop) + k.no_gap);
}
inline int after_del(const Consts& k, float m, float d, float i) {
    return pick3((m + k.no_gap) + k.gap_open, d + k.gap_extend, (i + k.gap_stop) + k.gap_open);
}
inline int after_ins(const Consts& k, float m, float i) {
    return pick2(m + k.gap_open, i + k.gap_extend);
}


// sample_mdi / sample_mi, align_pair.cc:336-385.  `forced` == kSample draws with
// p; kLikeliest takes the arg-max (used for the one draw made after a forced
// path is exhausted -- on the margins that draw has a single finite option);
// otherwise the given state is taken (oracle_path_logweight).
constexpr int kSample = -1, kLikeliest = -2;
inline std::pair<int, float> draw3(float lm, float ld, float li, float p, int forced) {
    const float m = ::expf(lm), d = ::expf(ld), i = ::expf(li);
    const float scale = m + d + i;
    p *= scale;
    int st;
    if(forced >= 0)
        st = forced;
    else if(forced == kLikeliest)
        st = pick3(lm, ld, li);
    else if(p < m)
        st = ORACLE_OP_M;
    else if(p < d + m)
        st = ORACLE_OP_D;
    else
        st = ORACLE_OP_I;
    const float lx = st == ORACLE_OP_M ? lm : (st == ORACLE_OP_D ? ld : li);
    return {st, lx - ::logf(scale)};
}
inline std::pair<int, float> draw2(float lm, float li, float p, int forced) {
    const float m = ::expf(lm), i = ::expf(li);
    const float scale = m + i;
    p *= scale;
    int st;
    if(forced >= 0)
        st = forced;
    else if(forced == kLikeliest)
        st = lm >= li ? ORACLE_OP_M : ORACLE_OP_I;
    else
        st = p < m ? ORACLE_OP_M : ORACLE_OP_I;
    // a forced DELETION here is not a path of the model (no D->I edge)
    const float lx = st == ORACLE_OP_M ? lm : (st == ORACLE_OP_I ? li : kLowest);
    return {st, lx - ::logf(scale)};
}

// RNG -----------------------------------------------------------------------
using u128 = unsigned __int128;
inline u128 load(const oracle_rng_t* r) { return (static_cast<u128>(r->hi) << 64) | r->lo; }
inline void store(oracle_rng_t* r, u128 s) {
    r->lo = static_cast<uint64_t>(s);
    r->hi = static_cast<uint64_t>(s >> 64);
}

// hash_impl_t, random.hpp:334-361: Weyl-sequence multilinear hash; the Weyl
// counter keeps running across the outputs.
void multilinear(uint64_t init, const uint32_t* in, size_t n_in, uint32_t* out, size_t n_out) {
    constexpr uint64_t kInc = 0x9e3779b97f4a7c15ULL;
    uint64_t w = init;
    for(size_t o = 0; o < n_out; ++o) {
        w += kInc;
        uint64_t acc = w;
        for(size_t q = 0; q < n_in; ++q) {
            w += kInc;
            acc += w * in[q];
        }
        w += kInc;
        acc += w;
        out[o] = static_cast<uint32_t>(acc >> 32);
    }
}

// Shared walker for sampleback in its three flavours.  `Src` supplies cell
// values; `rng` may be null when `forced_ops` drives the walk.
struct MdiSource {
    const float *M, *D, *I, *table;
    const uint8_t *a, *b;
    uint64_t rows, cols, L;
    Consts k;
    float ext_lm1, ext_l;
    float m(uint64_t i, uint64_t j) const { return M[i * cols + j]; }
    float d(uint64_t i, uint64_t j) const { return D[i * cols + j]; }
    float ins(uint64_t i, uint64_t j) const { return I[i * cols + j]; }
    bool body(uint64_t i, uint64_t j) const { return i >= L && j >= L; }
    float s(uint64_t i, uint64_t j) const {
        return table[static_cast<size_t>(a[i - L]) * kTableCols + b[j - L]];
    }
    // The 8 edge matrices of align_pair_work_t, rebuilt from M/D/I: body cells
    // by align_pair.cc:97-119, margin cells by resize + init_margins
    // (align_pair.hpp:76-88,108-111; margins themselves align_pair.cc:82-91).
    float mm(uint64_t i, uint64_t j) const {
        return body(i, j) ? ((m(i - 1, j - 1) + k.no_gap) + k.no_gap) + s(i, j) : kLowest;
    }
    float dm(uint64_t i, uint64_t j) const {
        return body(i, j) ? (d(i - 1, j - 1) + k.gap_stop) + s(i, j) : kLowest;
    }
    float im(uint64_t i, uint64_t j) const {
        return body(i, j) ? ((ins(i - 1, j - 1) + k.gap_stop) + k.no_gap) + s(i, j) : kLowest;
    }
    float md(uint64_t i, uint64_t j) const {
        return body(i, j) ? ((m(i - L, j) + k.no_gap) + k.gap_open) + ext_lm1 : kLowest;
    }
    float id(uint64_t i, uint64_t j) const {
        return body(i, j) ? ((ins(i - L, j) + k.gap_stop) + k.gap_open) + ext_lm1 : kLowest;
    }
    float dd(uint64_t i, uint64_t j) const {
        if(body(i, j)) return d(i - L, j) + ext_l;
        const uint64_t start = L - 1;
        if(j == start && i > start && (i - start) % L == 0)
            return (k.no_gap + k.gap_open) + pw(k.gap_extend, i - 1);
        return kLowest;
    }
    float mi(uint64_t i, uint64_t j) const {
        return body(i, j) ? (m(i, j - L) + k.gap_open) + ext_lm1 : kLowest;
    }
    float ii(uint64_t i, uint64_t j) const {
        if(body(i, j)) return ins(i, j - L) + ext_l;
        const uint64_t start = L - 1;
        if(i == start && j > start && (j - start) % L == 0)
            return k.gap_open + pw(k.gap_extend, j - 1);
        return kLowest;
    }
};

struct MatSource {  // the reference's own 11 matrices
    const float* mats;
    uint64_t rows, cols, L;
    float get(int which, uint64_t i, uint64_t j) const {
        return mats[static_cast<size_t>(which) * rows * cols + i * cols + j];
    }
    float m(uint64_t i, uint64_t j) const { return get(0, i, j); }
    float d(uint64_t i, uint64_t j) const { return get(1, i, j); }
    float ins(uint64_t i, uint64_t j) const { return get(2, i, j); }
    float mm(uint64_t i, uint64_t j) const { return get(3 + MM, i, j); }
    float md(uint64_t i, uint64_t j) const { return get(3 + MD, i, j); }
    float mi(uint64_t i, uint64_t j) const { return get(3 + MI, i, j); }
    float dm(uint64_t i, uint64_t j) const { return get(3 + DM, i, j); }
    float dd(uint64_t i, uint64_t j) const { return get(3 + DD, i, j); }
    float im(uint64_t i, uint64_t j) const { return get(3 + IM, i, j); }
    float id(uint64_t i, uint64_t j) const { return get(3 + ID, i, j); }
    float ii(uint64_t i, uint64_t j) const { return get(3 + II, i, j); }
};

// sampleback, align_pair.cc:401-458.  Ops are produced right-to-left into a
// scratch vector and reversed at the end.  With `forced` (ops left-to-right,
// n_forced of them) no random numbers are consumed.
template <class Src>
int64_t sample_walk(const Src& w, oracle_rng_t* rng, const uint8_t* forced, int64_t n_forced,
                    uint8_t* ops, float* score) {
    const uint64_t L = w.L;
    uint64_t i = w.rows - 1, j = w.cols - 1;
    int64_t fpos = n_forced;  // next forced op to consume is forced[fpos-1]
    auto next_forced = [&]() -> int {
        if(forced == nullptr) return kSample;
        if(fpos <= 0) return kLikeliest;  // path exhausted: the walk ends after this draw
        return forced[fpos - 1];
    };
    auto uniform = [&]() -> float { return rng != nullptr ? oracle_rng_f24(rng) : 0.0f; };

    std::vector<uint8_t> rev;
    rev.reserve(i + j);
    float total = 0.0f;
    float top = std::max(std::max(w.m(i, j), w.d(i, j)), w.ins(i, j));
    auto pick = draw3(w.m(i, j) - top, w.d(i, j) - top, w.ins(i, j) - top, uniform(), next_forced());
    total += pick.second;
    while(j > L - 1 || i > L - 1) {
        switch(pick.first) {
        case ORACLE_OP_M: {
            rev.push_back(ORACLE_OP_M);
            fpos -= 1;
            top = w.m(i, j);
            pick = draw3(w.mm(i, j) - top, w.dm(i, j) - top, w.im(i, j) - top, uniform(),
                         next_forced());
            total += pick.second;
            --i;
            --j;
            break;
        }
        case ORACLE_OP_D: {
            for(uint64_t t = 0; t < L; ++t) rev.push_back(ORACLE_OP_D);
            fpos -= static_cast<int64_t>(L);
            top = w.d(i, j);
            pick = draw3(w.md(i, j) - top, w.dd(i, j) - top, w.id(i, j) - top, uniform(),
                         next_forced());
            total += pick.second;
            i -= L;
            break;
        }
        default: {
            for(uint64_t t = 0; t < L; ++t) rev.push_back(ORACLE_OP_I);
            fpos -= static_cast<int64_t>(L);
            top = w.ins(i, j);
            pick = draw2(w.mi(i, j) - top, w.ii(i, j) - top, uniform(), next_forced());
            total += pick.second;
            j -= L;
            break;
        }
        }
    }
    if(ops != nullptr) std::reverse_copy(rev.begin(), rev.end(), ops);
    *score = total;
    return static_cast<int64_t>(rev.size());
}

}  // namespace

extern "C" {

void oracle_gap_consts(float gap_open, float gap_extend, float consts[4]) {
    consts[0] = std::log1pf(-gap_open);    // no_gap    semiring.hpp:119
    consts[1] = std::log1pf(-gap_extend);  // gap_stop
    consts[2] = ::logf(gap_open);          // gap_open  semiring.hpp:117
    consts[3] = ::logf(gap_extend);        // gap_extend
}

int oracle_fill(int semiring, const float* table, const float consts[4], int gap_len,
                const uint8_t* a, uint64_t len_a, const uint8_t* b, uint64_t len_b, float* M,
                float* D, float* I, float* edges) {
    if(gap_len < 1) return 1;
    const Consts k(consts);
    if(semiring == ORACLE_TROPICAL)
        fill<Tropical>(table, k, gap_len, a, len_a, b, len_b, M, D, I, edges);
    else
        fill<Log>(table, k, gap_len, a, len_a, b, len_b, M, D, I, edges);
    return 0;
}

int64_t oracle_traceback(const float* M, const float* D, const float* I, uint64_t rows,
                         uint64_t cols, const float consts[4], int gap_len, uint8_t* ops,
                         float* score) {
    const Consts k(consts);
    const uint64_t L = gap_len;
    uint64_t i = rows - 1, j = cols - 1;
    auto at = [cols](uint64_t r, uint64_t c) { return r * cols + c; };
    std::vector<uint8_t> rev;
    rev.reserve(i + j);
    // align_pair.cc:265-266 (values at the last cell are already terminal-adjusted)
    *score = std::max(std::max(M[at(i, j)], D[at(i, j)]), I[at(i, j)]);
    int st = pick3(M[at(i, j)], D[at(i, j)], I[at(i, j)]);
    while(j > L - 1 || i > L - 1) {  // align_pair.cc:268-299
        if(st == ORACLE_OP_M) {
            rev.push_back(ORACLE_OP_M);
            --i;
            --j;
            st = after_match(k, M[at(i, j)], D[at(i, j)], I[at(i, j)]);
        } else if(st == ORACLE_OP_D) {
            for(uint64_t t = 0; t < L; ++t) rev.push_back(ORACLE_OP_D);
            i -= L;
            st = after_del(k, M[at(i, j)], D[at(i, j)], I[at(i, j)]);
        } else {
            for(uint64_t t = 0; t < L; ++t) rev.push_back(ORACLE_OP_I);
            j -= L;
            st = after_ins(k, M[at(i, j)], I[at(i, j)]);
        }
    }
    std::reverse_copy(rev.begin(), rev.end(), ops);
    return static_cast<int64_t>(rev.size());
}

void oracle_tb_flags(const float* M, const float* D, const float* I, uint64_t rows,
                     uint64_t cols, const float consts[4], uint8_t* flags) {
    const Consts k(consts);
    for(uint64_t c = 0; c < rows * cols; ++c) {
        const int fm = after_match(k, M[c], D[c], I[c]);
        const int fd = after_del(k, M[c], D[c], I[c]);
        const int fi = after_ins(k, M[c], I[c]) == ORACLE_OP_M ? 0 : 1;
        flags[c] = static_cast<uint8_t>(fm | (fd << 2) | (fi << 4));
    }
}

int64_t oracle_traceback_flags(const uint8_t* flags, uint64_t rows, uint64_t cols, int gap_len,
                               uint8_t start_state, uint8_t* ops) {
    const uint64_t L = gap_len;
    uint64_t i = rows - 1, j = cols - 1;
    std::vector<uint8_t> rev;
    rev.reserve(i + j);
    int st = start_state;
    while(j > L - 1 || i > L - 1) {
        if(st == ORACLE_OP_M) {
            rev.push_back(ORACLE_OP_M);
            --i;
            --j;
            st = flags[i * cols + j] & 3;
        } else if(st == ORACLE_OP_D) {
            for(uint64_t t = 0; t < L; ++t) rev.push_back(ORACLE_OP_D);
            i -= L;
            st = (flags[i * cols + j] >> 2) & 3;
        } else {
            for(uint64_t t = 0; t < L; ++t) rev.push_back(ORACLE_OP_I);
            j -= L;
            st = ((flags[i * cols + j] >> 4) & 1) ? ORACLE_OP_I : ORACLE_OP_M;
        }
    }
    std::reverse_copy(rev.begin(), rev.end(), ops);
    return static_cast<int64_t>(rev.size());
}

int64_t oracle_viterbi(const float* table, const float consts[4], int gap_len, const uint8_t* a,
                       uint64_t len_a, const uint8_t* b, uint64_t len_b, uint8_t* ops,
                       float* score) {
    const uint64_t rows = len_a + gap_len, cols = len_b + gap_len;
    std::vector<float> M(rows * cols), D(rows * cols), I(rows * cols);
    oracle_fill(ORACLE_TROPICAL, table, consts, gap_len, a, len_a, b, len_b, M.data(), D.data(),
                I.data(), nullptr);
    return oracle_traceback(M.data(), D.data(), I.data(), rows, cols, consts, gap_len, ops, score);
}

int64_t oracle_viterbi_lowmem(const float* table, const float consts[4], int gap_len,
                              const uint8_t* a, uint64_t len_a, const uint8_t* b, uint64_t len_b,
                              uint8_t* ops, float* score) {
    const Consts k(consts);
    const uint64_t L = gap_len, rows = len_a + L, cols = len_b + L, start = L - 1;
    const float ext_lm1 = pw(k.gap_extend, L - 1), ext_l = pw(k.gap_extend, L);
    // ring of L+1 rows per matrix; row i lives in slot i % (L+1)
    const uint64_t ring = L + 1;
    std::vector<float> M(ring * cols), D(ring * cols), I(ring * cols);
    std::vector<uint8_t> flags(rows * cols);
    auto flag_of = [&k](float m, float d, float i) {
        const int fm = after_match(k, m, d, i), fd = after_del(k, m, d, i);
        const int fi = after_ins(k, m, i) == ORACLE_OP_M ? 0 : 1;
        return static_cast<uint8_t>(fm | (fd << 2) | (fi << 4));
    };
    uint8_t start_state = 0;
    for(uint64_t i = 0; i < rows; ++i) {
        float* m = &M[(i % ring) * cols];
        float* d = &D[(i % ring) * cols];
        float* in = &I[(i % ring) * cols];
        std::fill(m, m + cols, kLowest);
        std::fill(d, d + cols, kLowest);
        std::fill(in, in + cols, kLowest);
        if(i == start) {
            m[start] = 0.0f;
            for(uint64_t j = start + L; j < cols; j += L) in[j] = k.gap_open + pw(k.gap_extend, j - 1);
        } else if(i > start && (i - start) % L == 0) {
            d[start] = (k.no_gap + k.gap_open) + pw(k.gap_extend, i - 1);
        }
        if(i >= L) {
            const float* trow = table + static_cast<size_t>(a[i - L]) * kTableCols;
            const float* mdg = &M[((i - 1) % ring) * cols];
            const float* ddg = &D[((i - 1) % ring) * cols];
            const float* idg = &I[((i - 1) % ring) * cols];
            const float* mup = &M[((i - L) % ring) * cols];
            const float* dup = &D[((i - L) % ring) * cols];
            const float* iup = &I[((i - L) % ring) * cols];
            for(uint64_t j = L; j < cols; ++j) {
                const float s = trow[b[j - L]];
                const float m2m = ((mdg[j - 1] + k.no_gap) + k.no_gap) + s;
                const float d2m = (ddg[j - 1] + k.gap_stop) + s;
                const float i2m = ((idg[j - 1] + k.gap_stop) + k.no_gap) + s;
                const float m2d = ((mup[j] + k.no_gap) + k.gap_open) + ext_lm1;
                const float i2d = ((iup[j] + k.gap_stop) + k.gap_open) + ext_lm1;
                const float d2d = dup[j] + ext_l;
                const float m2i = (m[j - L] + k.gap_open) + ext_lm1;
                const float i2i = in[j - L] + ext_l;
                m[j] = std::max(std::max(m2m, d2m), i2m);
                d[j] = std::max(std::max(m2d, d2d), i2d);
                in[j] = std::max(m2i, i2i);
            }
        }
        if(i == rows - 1) {
            const uint64_t e = cols - 1;
            const float tm = (m[e] + k.no_gap) + k.no_gap, ti = (in[e] + k.gap_stop) + k.no_gap,
                        td = d[e] + k.gap_stop;
            *score = std::max(std::max(tm, td), ti);
            start_state = static_cast<uint8_t>(pick3(tm, td, ti));
        }
        for(uint64_t j = 0; j < cols; ++j) flags[i * cols + j] = flag_of(m[j], d[j], in[j]);
    }
    return oracle_traceback_flags(flags.data(), rows, cols, gap_len, start_state, ops);
}

void oracle_ops_to_strings(const uint8_t* ops, int64_t n_ops, const char* a_raw,
                           const char* b_raw, char* out_a, char* out_b) {
    size_t pa = 0, pb = 0;
    for(int64_t t = 0; t < n_ops; ++t) {
        out_a[t] = ops[t] == ORACLE_OP_I ? '-' : a_raw[pa++];
        out_b[t] = ops[t] == ORACLE_OP_D ? '-' : b_raw[pb++];
    }
    out_a[n_ops] = out_b[n_ops] = '\0';
}

void oracle_rng_seed(oracle_rng_t* rng, const char* const* seeds, int nseeds) {
    // string_seed_seq, random.hpp:522-540: decimal int32 strings are used as is,
    // anything else goes through the 32-bit FNV-style hash of random.hpp:465-472.
    std::vector<uint32_t> user;
    for(int q = 0; q < nseeds; ++q) {
        std::string_view sv{seeds[q]};
        int32_t v = 0;
        auto [p, ec] = std::from_chars(sv.data(), sv.data() + sv.size(), v, 10);
        if(ec == std::errc() && p == sv.data() + sv.size()) {
            user.push_back(static_cast<uint32_t>(v));
        } else {
            uint32_t h = 2166136261U;
            for(char ch : sv) h = (h * 16777619U) ^ static_cast<uint32_t>(static_cast<int>(ch));
            user.push_back(h);
        }
    }
    uint32_t inner[8], outer[4];
    multilinear(0x3423da0b87484307ULL, user.data(), user.size(), inner, 8);  // SeedSeq<8>::Seed
    multilinear(0xdf8b06c40fa44478ULL, inner, 8, outer, 4);                   // ::Generate
    u128 st = 0;
    std::memcpy(&st, outer, sizeof(outer));  // Lehmer64Fast::Seed(seed_type), random.hpp:105-109
    store(rng, st | 1);                      // SetState forces odd, random.hpp:131-134
}

uint64_t oracle_rng_bits(oracle_rng_t* rng) {
    const u128 st = load(rng) * static_cast<u128>(0xda942042e4dd58b5ULL);  // random.hpp:111
    store(rng, st);
    return static_cast<uint64_t>(st >> 64);
}

float oracle_rng_f24(oracle_rng_t* rng) {  // random.hpp:213-216
    const int64_t n = static_cast<int64_t>(oracle_rng_bits(rng) >> 40);
    return n / 16777216.0f;
}

int64_t oracle_sampleback(const float* mats, uint64_t rows, uint64_t cols, int gap_len,
                          oracle_rng_t* rng, uint8_t* ops, float* score) {
    MatSource src{mats, rows, cols, static_cast<uint64_t>(gap_len)};
    return sample_walk(src, rng, nullptr, 0, ops, score);
}

int64_t oracle_sampleback_mdi(const float* M, const float* D, const float* I, uint64_t rows,
                              uint64_t cols, const float* table, const float consts[4],
                              int gap_len, const uint8_t* a, const uint8_t* b,
                              oracle_rng_t* rng, uint8_t* ops, float* score) {
    const Consts k(consts);
    const uint64_t L = gap_len;
    MdiSource src{M, D, I, table, a, b, rows, cols, L, k, pw(k.gap_extend, L - 1), pw(k.gap_extend, L)};
    return sample_walk(src, rng, nullptr, 0, ops, score);
}

float oracle_path_logweight(const float* M, const float* D, const float* I, uint64_t rows,
                            uint64_t cols, const float* table, const float consts[4],
                            int gap_len, const uint8_t* a, const uint8_t* b,
                            const uint8_t* ops, int64_t n_ops) {
    const Consts k(consts);
    const uint64_t L = gap_len;
    MdiSource src{M, D, I, table, a, b, rows, cols, L, k, pw(k.gap_extend, L - 1), pw(k.gap_extend, L)};
    float score = 0.0f;
    sample_walk(src, nullptr, ops, n_ops, nullptr, &score);
    return score;
}

float oracle_path_score(const float* table, const float consts[4], int gap_len, const uint8_t* a,
                        uint64_t len_a, const uint8_t* b, uint64_t len_b, const uint8_t* ops,
                        int64_t n_ops) {
    const Consts k(consts);
    const uint64_t L = gap_len, start = L - 1, rows = len_a + L, cols = len_b + L;
    const float ext_lm1 = pw(k.gap_extend, L - 1), ext_l = pw(k.gap_extend, L);
    uint64_t i = start, j = start;  // matrix cell the path has reached
    int st = ORACLE_OP_M;           // state at that cell
    float v = 0.0f;                 // M(start, start) = one()
    int64_t t = 0;
    while(t < n_ops) {
        const int op = ops[t];
        if(op == ORACLE_OP_M) {
            if(i + 1 >= rows || j + 1 >= cols) return kLowest;
            ++i;
            ++j;
            const float s = table[static_cast<size_t>(a[i - L]) * kTableCols + b[j - L]];
            v = st == ORACLE_OP_M ? ((v + k.no_gap) + k.no_gap) + s
                                  : (st == ORACLE_OP_D ? (v + k.gap_stop) + s : ((v + k.gap_stop) + k.no_gap) + s);
            t += 1;
        } else if(op == ORACLE_OP_D) {
            if(i + L >= rows) return kLowest;
            for(uint64_t q = 0; q < L; ++q)
                if(t + static_cast<int64_t>(q) >= n_ops || ops[t + q] != ORACLE_OP_D) return kLowest;
            i += L;
            if(j == start)  // margin column (align_pair.cc:82-86)
                v = (k.no_gap + k.gap_open) + pw(k.gap_extend, i - 1);
            else
                v = st == ORACLE_OP_M ? ((v + k.no_gap) + k.gap_open) + ext_lm1
                                      : (st == ORACLE_OP_D ? v + ext_l : ((v + k.gap_stop) + k.gap_open) + ext_lm1);
            t += static_cast<int64_t>(L);
        } else {
            if(j + L >= cols || st == ORACLE_OP_D) return kLowest;  // no D -> I edge in the model
            for(uint64_t q = 0; q < L; ++q)
                if(t + static_cast<int64_t>(q) >= n_ops || ops[t + q] != ORACLE_OP_I) return kLowest;
            j += L;
            if(i == start)  // margin row (align_pair.cc:88-90)
                v = k.gap_open + pw(k.gap_extend, j - 1);
            else
                v = st == ORACLE_OP_M ? (v + k.gap_open) + ext_lm1 : v + ext_l;
            t += static_cast<int64_t>(L);
        }
        st = op;
    }
    if(i != rows - 1 || j != cols - 1) return kLowest;
    // terminal state (align_pair.cc:130-138)
    if(st == ORACLE_OP_M) return (v + k.no_gap) + k.no_gap;
    if(st == ORACLE_OP_D) return v + k.gap_stop;
    return (v + k.gap_stop) + k.no_gap;
}

void oracle_libm(int op, const float* in, uint64_t n, float* out) {
    for(uint64_t i = 0; i < n; ++i) out[i] = op == 0 ? ::expf(in[i]) : (op == 1 ? ::log1pf(in[i]) : ::logf(in[i]));
}

double oracle_viterbi_batch_timed(const float* table, const float consts[4], int gap_len,
                                  uint64_t n_pairs, const uint8_t* a_cat, const uint64_t* a_off,
                                  const uint8_t* b_cat, const uint64_t* b_off, int threads,
                                  float* scores) {
    if(threads < 1) threads = 1;
    std::atomic<uint64_t> next{0};
    auto worker = [&]() {
        std::vector<uint8_t> ops;
        for(;;) {
            const uint64_t p = next.fetch_add(1);
            if(p >= n_pairs) break;
            const uint64_t la = a_off[p + 1] - a_off[p], lb = b_off[p + 1] - b_off[p];
            ops.resize(la + lb + 1);
            float sc = 0.0f;
            oracle_viterbi(table, consts, gap_len, a_cat + a_off[p], la, b_cat + b_off[p], lb,
                           ops.data(), &sc);
            if(scores != nullptr) scores[p] = sc;
        }
    };
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for(int t = 1; t < threads; ++t) pool.emplace_back(worker);
    worker();
    for(auto& th : pool) th.join();
    const auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
