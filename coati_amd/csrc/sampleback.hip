// sampleback: stochastic traceback over the Forward matrices resident in HBM.
//
//   sampleback, sample_mdi, sample_mi          src/lib/align_pair.cc:336-458
//   Lehmer64Fast::operator(), random_f24       contrib/random/random.hpp:80-136,213-216
//
// The reference keeps eight edge matrices (align_pair.hpp:94-103); here the three
// (or two) edge values a step needs are recomputed from the predecessor cell's
// M/D/I with the fill's own expressions (align_pair.cc:97-119), so only M/D/I
// (12 B/cell) are stored.  A walk is a serial chain (one f24() draw per step, the
// next cell depends on the draw), so the parallelism is across walks:
//   * exact stream (independent = 0): one walker per PAIR draws its n samples one
//     after the other from one Lehmer stream, exactly like marg_sample's loop
//     (src/lib/align_marginal.cc:590-593);
//   * independent streams (independent = 1): one walker per (pair, sample); sample n
//     starts from the pair's state advanced by n * 2^32 draws (host: jump-ahead),
//     so sample 0 is the reference's first sample and the rest are statistically
//     equivalent but not the same draws.
// fp32 throughout; expf and logf are the bit-exact restatements of glibc_math.hpp, so with a bit-exact
// Forward fill every draw, path and log-weight equals the CPU's.
#include "common.hpp"

#include <algorithm>

namespace coati_hip_detail {
namespace {

struct Rng128 {
    uint64_t lo, hi;
};
__device__ __forceinline__ float rng_f24(Rng128& s) {
    constexpr uint64_t kMult = 0xda942042e4dd58b5ULL;  // random.hpp:95
    const uint64_t lo = s.lo * kMult;
    const uint64_t hi = s.hi * kMult + __umul64hi(s.lo, kMult);
    s.lo = lo;
    s.hi = hi;
    return static_cast<float>(static_cast<int64_t>(hi >> 40)) / 16777216.0f;
}

struct Walker {
    const GapConsts k;
    const uint32_t L, la, lb;
    const float ext_lm1, ext_l;
    const float* __restrict__ table;
    const uint8_t* __restrict__ a;
    const uint8_t* __restrict__ b;
    const float* __restrict__ mdi;
    const PairDesc& pd;
    const uint64_t* exp_tab;  // LDS copy of expf's table (glibc_math.hpp)

    // M/D/I of MATRIX cell (i, j); the last cell carries the terminal adjustment
    // (the reference stores it adjusted, align_pair.cc:130-138).
    __device__ __forceinline__ void cell(uint32_t i, uint32_t j, float& m, float& d, float& in) const {
        if(i >= L && j >= L) {
            const Mdi v = *reinterpret_cast<const Mdi*>(mdi + mdi_index(pd, i - L, j - L, 0));  // one 12-byte load
            m = v.m;
            d = v.d;
            in = v.in;
        } else {
            margin_mdi(k, L, i, j, m, d, in);
        }
        if(i == la + L - 1 && j == lb + L - 1) {
            m = (m + k.ng) + k.ng;
            in = (in + k.gs) + k.ng;
            d = d + k.gs;
        }
    }
    // like cell() but never adjusted: the fill's inputs (a predecessor is never the last cell)
    __device__ __forceinline__ void pred(uint32_t i, uint32_t j, float& m, float& d, float& in) const {
        if(i >= L && j >= L) {
            const Mdi v = *reinterpret_cast<const Mdi*>(mdi + mdi_index(pd, i - L, j - L, 0));  // one 12-byte load
            m = v.m;
            d = v.d;
            in = v.in;
        } else {
            margin_mdi(k, L, i, j, m, d, in);
        }
    }
    __device__ __forceinline__ float subst(uint32_t i, uint32_t j) const {
        return table[static_cast<uint32_t>(a[i - L]) * kTabCols + b[j - L]];
    }
};

// sample_mdi (align_pair.cc:336-358): returns the state, adds log(x) - log(scale) to score
__device__ __forceinline__ int sample3(float lm, float ld, float li, float p, float& score, const uint64_t* exp_tab) {
    const float m = libm::expf_nonpos(lm, exp_tab), d = libm::expf_nonpos(ld, exp_tab), i = libm::expf_nonpos(li, exp_tab);
    const float scale = m + d + i;
    p *= scale;
    int st;
    float lx;
    if(p < m) {
        st = COATI_HIP_OP_MATCH;
        lx = lm;
    } else if(p < d + m) {
        st = COATI_HIP_OP_DEL;
        lx = ld;
    } else {
        st = COATI_HIP_OP_INS;
        lx = li;
    }
    score += lx - libm::logf_pos(scale);
    return st;
}
// sample_mi (align_pair.cc:370-385) is sample3 with a middle term of weight exactly zero (ld = -inf):
// scale = (m + 0) + i, `p < 0 + m` is the test that already failed, and the picked log-term and
// logf(scale) are the same floats -- so the walk below calls sample3 for all three states.

// One walk (sampleback, align_pair.cc:401-458).  Ops are written right-to-left into
// [slot, slot + la + lb); returns the position of the first op.
//
// Shape of the loop.  A step at cell (i, j) in state st needs M/D/I of ONE predecessor cell --
// (i-1, j-1), (i-L, j) or (i, j-L) for st = M, D, I -- and the draw then picks the state st' the
// walk is in at that cell.  The step is branch-free so that the 64 independent walks of a wavefront
// do not serialise three code paths: the states differ in a handful of selects, then ONE sample_mdi
// runs for all lanes.  The triple loaded for the step is also M/D/I of the cell the walk arrives
// at, so nothing is loaded twice (one 12-byte load per step).  Requesting the three possible
// predecessors of the NEXT cell ahead of the arithmetic was tried twice (matrix-by-matrix layout:
// 9 loads per step, 71 ms against 51 for 16 x 1 000 samples; interleaved layout: 3 loads, 63 ms
// against 40) and loses both times: at 2 wavefronts per SIMD a wavefront issues one instruction
// per ~4.4 cycles, so the ~100 instructions of extra index arithmetic and selects cost more than
// the load latency they hide.  The values and the order of every float operation are those of the
// reference.
struct Triple {
    float m, d, in;
};

__device__ uint64_t sample_walk(const Walker& w, Rng128& rng, uint8_t* __restrict__ ops, uint64_t slot, float& score,
                                uint32_t& draws) {
    const uint32_t L = w.L;
    draws = 1;  // the draw that picks the state of the last cell
    uint32_t i = w.la + L - 1, j = w.lb + L - 1;
    uint64_t pos = slot + w.la + w.lb;
    score = 0.0f;
    Triple cur;  // M/D/I of the cell the walk is at (terminal-adjusted at the last cell)
    w.cell(i, j, cur.m, cur.d, cur.in);
    int st;
    {
        const float top = fmaxf(fmaxf(cur.m, cur.d), cur.in);
        st = sample3(cur.m - top, cur.d - top, cur.in - top, rng_f24(rng), score, w.exp_tab);
    }
    while(j > L - 1 || i > L - 1) {
        ++draws;
        const bool body = i >= L && j >= L;
        const bool is_m = st == COATI_HIP_OP_MATCH, is_d = st == COATI_HIP_OP_DEL;
        // where this step goes, and M/D/I there (a walk that would leave the matrix -- impossible
        // with non-zero probability -- reads nothing)
        const uint32_t pi = is_m ? i - 1 : (is_d ? i - L : i), pj = is_m ? j - 1 : (is_d ? j : j - L);
        // A walk can only leave the matrix (or overrun its slot of la + lb ops) if every weight of a draw
        // was zero or NaN -- a table with -inf / NaN entries; model_create rejects those, this is the belt to
        // those braces: end the walk, mark the sample with a NaN log-weight, never write outside the slot.
        if(pi > i || pj > j || pos < slot + (is_m ? 1u : L)) {
            score = __builtin_nanf("");
            break;
        }
        Triple t{kLowest, kLowest, kLowest};
        w.pred(pi, pj, t.m, t.d, t.in);
        const float top = is_m ? cur.m : (is_d ? cur.d : cur.in);
        // the emitted columns: one match, or L gap columns
        const uint32_t n_emit = is_m ? 1u : L;
        for(uint32_t q = 0; q < n_emit; ++q) ops[--pos] = static_cast<uint8_t>(st);
        // the edge values of the fill (align_pair.cc:97-119) at (i, j) for the current state
        float e0 = kLowest, e1 = kLowest, e2 = kLowest;
        if(body) {
            const float s = w.subst(i, j);
            // st = M: mch_mch, del_mch, ins_mch;  D: mch_del, del_del, ins_del;  I: mch_ins, -, ins_ins
            const float m1 = t.m + (is_m || is_d ? w.k.ng : w.k.go);  // (M+ng) | (M+ng) | (M+go)
            e0 = is_m ? (m1 + w.k.ng) + s : (is_d ? (m1 + w.k.go) + w.ext_lm1 : m1 + w.ext_lm1);
            e1 = is_m ? (t.d + w.k.gs) + s : t.d + w.ext_l;
            const float i1 = t.in + w.k.gs;
            e2 = is_m ? (i1 + w.k.ng) + s : (is_d ? (i1 + w.k.go) + w.ext_lm1 : t.in + w.ext_l);
        } else if(!is_m) {
            // margins: del_del / ins_ins are copies of the margin D / I made BEFORE the terminal
            // adjustment (init_margins, align_pair.hpp:108-111); everything else stays `lowest`
            float mm0, dm0, im0;
            margin_mdi(w.k, L, i, j, mm0, dm0, im0);
            if(is_d)
                e1 = dm0;
            else
                e2 = im0;
        }
        // state I draws among two terms (sample_mi): a middle term of exactly zero weight
        const float l1 = (is_m || is_d) ? e1 - top : -__builtin_inff();
        st = sample3(e0 - top, l1, e2 - top, rng_f24(rng), score, w.exp_tab);
        i = pi;
        j = pj;
        cur = t;
    }
    return pos;
}

__global__ __launch_bounds__(64) void sampleback_kernel(const float* __restrict__ table, GapConsts k, uint32_t L,
                                                        const PairDesc* __restrict__ pairs, uint32_t n_pairs,
                                                        uint32_t n_samples, int independent,
                                                        const uint8_t* __restrict__ a_cat,
                                                        const uint8_t* __restrict__ b_cat,
                                                        const float* __restrict__ mdi, uint64_t* __restrict__ rng_states,
                                                        const uint64_t* __restrict__ sample_base, uint8_t* __restrict__ ops,
                                                        uint64_t* __restrict__ ops_start, uint32_t* __restrict__ ops_len,
                                                        float* __restrict__ log_weights) {
    __shared__ uint64_t exp_tab[32];
    load_exp_table(exp_tab, threadIdx.x);
    __syncthreads();
    const uint64_t walker = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    const uint64_t n_walkers = independent ? static_cast<uint64_t>(n_pairs) * n_samples : n_pairs;
    if(walker >= n_walkers) return;
    const uint32_t pair = independent ? static_cast<uint32_t>(walker / n_samples) : static_cast<uint32_t>(walker);
    const uint32_t first = independent ? static_cast<uint32_t>(walker % n_samples) : 0u;
    const uint32_t count = independent ? 1u : n_samples;
    const PairDesc pd = pairs[pair];
    const Walker w{k, L, pd.la, pd.lb, k.ge * static_cast<float>(L - 1), k.ge * static_cast<float>(L),
                   table + static_cast<size_t>(pd.table) * kTabFloats, a_cat + pd.a_off, b_cat + pd.b_off, mdi, pd, exp_tab};
    Rng128 rng{rng_states[2 * walker], rng_states[2 * walker + 1]};
    const uint64_t width = static_cast<uint64_t>(pd.la) + pd.lb;
    for(uint32_t n = first; n < first + count; ++n) {
        const uint64_t idx = static_cast<uint64_t>(pair) * n_samples + n;
        const uint64_t slot = sample_base[pair] + n * width;
        float score;
        uint32_t draws;
        const uint64_t pos = sample_walk(w, rng, ops, slot, score, draws);
        ops_start[idx] = pos;
        ops_len[idx] = static_cast<uint32_t>(slot + width - pos);
        log_weights[idx] = score;
    }
    rng_states[2 * walker] = rng.lo;  // the stream continues where this launch stopped
    rng_states[2 * walker + 1] = rng.hi;
}

// ---- exact stream, in parallel --------------------------------------------------------------
// marg_sample draws its n samples one after the other from ONE Lehmer stream
// (align_marginal.cc:590-593): sample s starts where sample s-1 stopped, and how many draws a
// sample takes (1 + its number of moves) is only known once it has been walked.  The host
// (abi.hip: sampleback_speculative) therefore lets a window of candidate start offsets be walked
// for every sample of a chunk -- each walker jumps the generator to its assumed offset
// (state * MULT^offset mod 2^128) and reports how many draws it took -- and then follows the chain
// of true offsets through the candidates.  Every committed sample is exactly the walk the serial
// loop would have produced.
__global__ __launch_bounds__(64) void spec_walk_kernel(const float* __restrict__ table, GapConsts k, uint32_t L,
                                                       const PairDesc* __restrict__ pairs,
                                                       const uint8_t* __restrict__ a_cat,
                                                       const uint8_t* __restrict__ b_cat, const float* __restrict__ mdi,
                                                       const uint64_t* __restrict__ origin_state,
                                                       const uint64_t* __restrict__ mult_pow,
                                                       const SpecCandidate* __restrict__ cands, uint32_t n_cands,
                                                       uint8_t* __restrict__ tmp_ops, uint64_t* __restrict__ c_start,
                                                       uint32_t* __restrict__ c_len, float* __restrict__ c_lw,
                                                       uint32_t* __restrict__ c_draws) {
    __shared__ uint64_t exp_tab[32];
    load_exp_table(exp_tab, threadIdx.x);
    __syncthreads();
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if(idx >= n_cands) return;
    const SpecCandidate cd = cands[idx];
    const PairDesc pd = pairs[cd.pair];
    // jump: state at the chunk origin times MULT^offset (square-and-multiply over the host's table
    // of MULT^(2^b)), all modulo 2^128
    unsigned __int128 st = (static_cast<unsigned __int128>(origin_state[2 * cd.pair + 1]) << 64) | origin_state[2 * cd.pair];
    for(uint32_t bit = 0, off = cd.offset; off != 0; ++bit, off >>= 1)
        if(off & 1u) st *= (static_cast<unsigned __int128>(mult_pow[2 * bit + 1]) << 64) | mult_pow[2 * bit];
    Rng128 rng{static_cast<uint64_t>(st), static_cast<uint64_t>(st >> 64)};
    const Walker w{k, L, pd.la, pd.lb, k.ge * static_cast<float>(L - 1), k.ge * static_cast<float>(L),
                   table + static_cast<size_t>(pd.table) * kTabFloats, a_cat + pd.a_off, b_cat + pd.b_off, mdi, pd, exp_tab};
    float score;
    uint32_t draws;
    const uint64_t width = static_cast<uint64_t>(pd.la) + pd.lb;
    const uint64_t pos = sample_walk(w, rng, tmp_ops, cd.slot, score, draws);
    c_start[idx] = pos;
    c_len[idx] = static_cast<uint32_t>(cd.slot + width - pos);
    c_lw[idx] = score;
    c_draws[idx] = draws;
}

// ---- exact stream, round 4: the step table -------------------------------------------------------
// The speculation above walks ~100 candidates per committed sample, and every step of every candidate evaluated
// three expf and one logf (the bit-exact restatements: ~250 instructions) on values that depend only on the CELL and
// the STATE the walk is in -- not on the draw.  So they are computed once: per body cell and state
//     scale = m + d + i,  m,  d + m            the thresholds of sample_mdi / sample_mi (align_pair.cc:336-385)
//     l_M - logf(scale), l_D - ..., l_I - ...  what the picked state adds to the sample's log-weight
// with exactly the operations of sample_walk / sample3 above (same order, same restatements: same bits).  A step is
// then one 24-byte load, one multiply, two compares and an add; 16 pairs of 1 kb are 16 M cells -- the work of ~2 % of
// the steps one sampleback call walks.  gap_len 1; margin cells (the last few steps of a walk) keep the formulas.
// And the candidates no longer write their ops: the rounds only resolve the chain of stream offsets (how many draws
// each sample takes), then ONE launch walks every (pair, sample) from its now known offset and writes the results.
struct StepEntry {
    float scale, m, dm, inc_m, inc_d, inc_i;
};
static_assert(sizeof(StepEntry) == 24, "one 24-byte entry per (cell, state)");

__device__ __forceinline__ StepEntry step_entry(const Walker& w, uint32_t i, uint32_t j, int st) {
    // (i, j): matrix coordinates of a BODY cell, gap_len 1; the lines of sample_walk's loop body, in its order
    Triple cur;
    w.cell(i, j, cur.m, cur.d, cur.in);
    const bool is_m = st == COATI_HIP_OP_MATCH, is_d = st == COATI_HIP_OP_DEL;
    const uint32_t pi = is_m ? i - 1 : (is_d ? i - 1 : i), pj = is_m ? j - 1 : (is_d ? j : j - 1);
    Triple t{kLowest, kLowest, kLowest};
    w.pred(pi, pj, t.m, t.d, t.in);
    const float top = is_m ? cur.m : (is_d ? cur.d : cur.in);
    const float s = w.subst(i, j);
    const float m1 = t.m + (is_m || is_d ? w.k.ng : w.k.go);
    const float e0 = is_m ? (m1 + w.k.ng) + s : (is_d ? (m1 + w.k.go) + w.ext_lm1 : m1 + w.ext_lm1);
    const float e1 = is_m ? (t.d + w.k.gs) + s : t.d + w.ext_l;
    const float i1 = t.in + w.k.gs;
    const float e2 = is_m ? (i1 + w.k.ng) + s : (is_d ? (i1 + w.k.go) + w.ext_lm1 : t.in + w.ext_l);
    const float l0 = e0 - top, l1 = (is_m || is_d) ? e1 - top : -__builtin_inff(), l2 = e2 - top;
    const float m = libm::expf_nonpos(l0, w.exp_tab), d = libm::expf_nonpos(l1, w.exp_tab), in = libm::expf_nonpos(l2, w.exp_tab);
    const float scale = m + d + in;
    const float ls = libm::logf_pos(scale);
    return StepEntry{scale, m, d + m, l0 - ls, l1 - ls, l2 - ls};
}

struct StepThresholds {  // the first 12 bytes of a StepEntry
    float scale, m, dm;
};
// The table a walk reads: the pair's entries (a band of diagonals, common.hpp: step_band), row-major -- column j of row i is
// entry k = j - i + dhi of the row, so a diagonal move is a constant stride of `width` cells -- and the M-state thresholds
// once more, DIAGONAL-major: diagonal k, position t = min(i, j) - 1 at k * diag_len + t.  A run of matches walks down a
// diagonal: the batch of table_walk_count reads 16 consecutive threshold entries (192 bytes), and the candidates of a pair,
// which are all near the same diagonals, share cache lines within a load instruction.
struct StepTable {
    const StepEntry* __restrict__ steps;      // the pair's entries
    const StepThresholds* __restrict__ thr;   // the pair's M thresholds (may be null: the host-rounds path)
    StepBand band;
    __device__ __forceinline__ int64_t diag(uint32_t i, uint32_t j) const { return static_cast<int64_t>(j) - static_cast<int64_t>(i) + band.dhi; }
    __device__ __forceinline__ bool holds(int64_t k) const { return k >= 0 && k < static_cast<int64_t>(band.width); }
    __device__ __forceinline__ const StepEntry* cell(uint32_t i, int64_t k) const {  // the three entries of (i, j), holds(k)
        return steps + (static_cast<uint64_t>(i - 1) * band.width + static_cast<uint64_t>(k)) * 3;
    }
    __device__ __forceinline__ const StepThresholds* thr_at(uint32_t i, uint32_t j, int64_t k) const {
        return thr + static_cast<uint64_t>(k) * band.diag_len + (min(i, j) - 1u);
    }
};

__global__ __launch_bounds__(256) void step_table_kernel(const float* __restrict__ table, GapConsts k, const PairDesc* __restrict__ pairs,
                                                         const uint64_t* __restrict__ tab_off, uint32_t n_pairs, uint32_t band_half,
                                                         const uint8_t* __restrict__ a_cat, const uint8_t* __restrict__ b_cat,
                                                         const float* __restrict__ mdi, StepEntry* __restrict__ steps,
                                                         const uint64_t* __restrict__ thr_off, StepThresholds* __restrict__ thr_m) {
    __shared__ uint64_t exp_tab[32];
    load_exp_table(exp_tab, threadIdx.x);
    __syncthreads();
    const uint32_t pair = blockIdx.y;
    if(pair >= n_pairs) return;
    const PairDesc pd = pairs[pair];
    const StepBand band = step_band(pd.la, pd.lb, band_half);
    const uint64_t cells = static_cast<uint64_t>(pd.la) * band.width;  // (row, k) of the band; columns outside the matrix are skipped
    const Walker w{k, 1u, pd.la, pd.lb, k.ge * 0.0f, k.ge * 1.0f, table + static_cast<size_t>(pd.table) * kTabFloats, a_cat + pd.a_off, b_cat + pd.b_off,
                   mdi, pd, exp_tab};
    StepEntry* out = steps + tab_off[pair];
    for(uint64_t c = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; c < cells; c += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint32_t row = static_cast<uint32_t>(c / band.width), kk = static_cast<uint32_t>(c - static_cast<uint64_t>(row) * band.width);
        const int64_t jj = static_cast<int64_t>(row + 1) - band.dhi + kk;
        if(jj < 1 || jj > static_cast<int64_t>(pd.lb)) continue;
        const uint32_t i = row + 1, j = static_cast<uint32_t>(jj);
#pragma unroll
        for(int st = 0; st < 3; ++st) {
            const StepEntry e = step_entry(w, i, j, st);
            out[c * 3 + st] = e;
            if(st == COATI_HIP_OP_MATCH && thr_m != nullptr)
                thr_m[thr_off[pair] + static_cast<uint64_t>(kk) * band.diag_len + (min(i, j) - 1u)] = StepThresholds{e.scale, e.m, e.dm};
        }
    }
}

// Where a table walk takes its draws from.  The final launch draws from the generator (one 128-bit multiply and a
// conversion per draw: ~27 of a step's ~50 vector instructions); a round's candidates read the pair's stream from a table
// that one launch per round fills (spec_draws_kernel): candidate `offset` takes table[offset], table[offset + 1], ... --
// the same floats, and eight of them arrive with the batch's step-table entries.
struct DrawsFromRng {
    Rng128& rng;
    __device__ __forceinline__ void prefetch(float (&)[8]) const {}
    __device__ __forceinline__ float take(const float (&)[8], int) { return rng_f24(rng); }
    __device__ __forceinline__ float next() { return rng_f24(rng); }
};
struct DrawsFromTable {
    const float* __restrict__ at;
    __device__ __forceinline__ void prefetch(float (&d)[8]) const {
#pragma unroll
        for(int q = 0; q < 8; ++q) d[q] = at[q];  // (the table has slack behind every slice)
    }
    __device__ __forceinline__ float take(const float (&d)[8], int q) {
        ++at;
        return d[q];
    }
    __device__ __forceinline__ float next() { return *at++; }
};

// sample_walk with the body steps read from the table.  kOps: write the ops (the final launch) or only count the
// draws (the candidates of a round).  Identical decisions, log-weight and draw count as sample_walk.
template <bool kOps, class Draws>
__device__ uint64_t table_walk(const Walker& w, const StepTable& tb, Draws src, uint8_t* __restrict__ ops, uint64_t slot,
                               float& score, uint32_t& draws) {
    draws = 1;
    uint32_t i = w.la, j = w.lb;  // (gap_len 1: the last cell is (la, lb))
    uint64_t pos = slot + w.la + w.lb;
    score = 0.0f;
    Triple cur;
    w.cell(i, j, cur.m, cur.d, cur.in);
    bool cur_valid = true;  // `cur` is M/D/I of the cell the walk is at (only kept up to date on the margins)
    int st;
    {
        const float top = fmaxf(fmaxf(cur.m, cur.d), cur.in);
        st = sample3(cur.m - top, cur.d - top, cur.in - top, src.next(), score, w.exp_tab);
    }
    // Body steps go in BATCHES.  A walk is a chain of ~la dependent loads (~0.35 us each: a round of candidates took the
    // same 0.37-0.47 ms whether it held 20 000 or 130 000 of them); but 97 % of a sample's steps are matches, which move
    // down the diagonal, so the entries of the next kAhead - 1 diagonal cells for state M are loaded TOGETHER with the
    // entry the walk needs now, and consumed for as long as the draws keep choosing M (a lane whose draw chose a gap
    // state starts its next batch from there).  Same entries, same draws, same order: same decisions.
    constexpr int kAhead = 8;
    bool failed = false;
    while((j > 0 || i > 0) && !failed) {
        if(i >= 1 && j >= 1) {
            StepEntry e[kAhead];
            const int64_t k0 = tb.diag(i, j);
            // (a cell outside the band of the table: its entry the way the table's were made -- rare: a sampled path that
            // drifted more than the band's half width off the pair's straight line)
            e[0] = tb.holds(k0) ? tb.cell(i, k0)[static_cast<uint32_t>(st)] : step_entry(w, i, j, st);
            uint32_t room;  // steps 1 .. room of the batch are at body cells (i1 - q + 1, j1 - q + 1) the table holds
            {
                const bool m0 = st == COATI_HIP_OP_MATCH, d0 = st == COATI_HIP_OP_DEL;
                const uint32_t i1 = (m0 || d0) ? i - 1 : i, j1 = d0 ? j : j - 1;  // where this step leads
                const int64_t k1 = tb.diag(i1, j1);
                room = tb.holds(k1) ? min(i1, j1) : 0u;
                // (entry of cell (i1, j1) for M, then one diagonal step = `width` cells back per entry: no multiply per address)
                const StepEntry* at = room > 0 ? tb.cell(i1, k1) + COATI_HIP_OP_MATCH : tb.steps;
                const uint64_t diag = static_cast<uint64_t>(tb.band.width) * 3;
#pragma unroll
                for(int q = 1; q < kAhead; ++q) {
                    e[q] = *(static_cast<uint32_t>(q) <= room ? at : tb.steps);
                    at -= diag;
                }
            }
            static_assert(kAhead == 8, "the draw sources hand out eight draws per batch");
            float dr[8];
            src.prefetch(dr);
            bool go = true;
#pragma unroll
            for(int q = 0; q < kAhead; ++q) {
                if(go) {
                    ++draws;
                    if(pos < slot + 1u) {  // (as in sample_walk: only with a table of -inf / NaN weights)
                        score = __builtin_nanf("");
                        failed = true;
                        go = false;
                    } else {
                        if(kOps) ops[--pos] = static_cast<uint8_t>(st);
                        else --pos;
                        float p = src.take(dr, q);
                        p *= e[q].scale;
                        int nst;
                        float inc;
                        if(p < e[q].m) {
                            nst = COATI_HIP_OP_MATCH;
                            inc = e[q].inc_m;
                        } else if(p < e[q].dm) {
                            nst = COATI_HIP_OP_DEL;
                            inc = e[q].inc_d;
                        } else {
                            nst = COATI_HIP_OP_INS;
                            inc = e[q].inc_i;
                        }
                        score += inc;
                        const bool is_m = st == COATI_HIP_OP_MATCH, is_d = st == COATI_HIP_OP_DEL;
                        i = (is_m || is_d) ? i - 1 : i;
                        j = is_d ? j : j - 1;
                        st = nst;
                        go = nst == COATI_HIP_OP_MATCH && static_cast<uint32_t>(q) < room;  // (the next prefetched entry is this cell's, for M)
                    }
                }
            }
            cur_valid = false;
            continue;
        }
        ++draws;
        const bool is_m = st == COATI_HIP_OP_MATCH, is_d = st == COATI_HIP_OP_DEL;
        const uint32_t pi = is_m ? i - 1 : (is_d ? i - 1 : i), pj = is_m ? j - 1 : (is_d ? j : j - 1);
        if(pi > i || pj > j || pos < slot + 1u) {  // (as in sample_walk: only with a table of -inf / NaN weights)
            score = __builtin_nanf("");
            break;
        }
        if(kOps) ops[--pos] = static_cast<uint8_t>(st);
        else --pos;
        {
            // a margin cell: del_del / ins_ins are copies of the margin D / I (init_margins, align_pair.hpp:108-111)
            if(!cur_valid) w.pred(i, j, cur.m, cur.d, cur.in);
            Triple t{kLowest, kLowest, kLowest};
            w.pred(pi, pj, t.m, t.d, t.in);
            const float top = is_m ? cur.m : (is_d ? cur.d : cur.in);
            float e0 = kLowest, e1 = kLowest, e2 = kLowest;
            if(!is_m) {
                float mm0, dm0, im0;
                margin_mdi(w.k, 1u, i, j, mm0, dm0, im0);
                if(is_d)
                    e1 = dm0;
                else
                    e2 = im0;
            }
            const float l1 = (is_m || is_d) ? e1 - top : -__builtin_inff();
            st = sample3(e0 - top, l1, e2 - top, src.next(), score, w.exp_tab);
            cur = t;
            cur_valid = true;
        }
        i = pi;
        j = pj;
    }
    return pos;
}

__device__ __forceinline__ Rng128 rng_jump(const uint64_t* __restrict__ origin_state, const uint64_t* __restrict__ mult_pow, uint32_t pair, uint64_t offset) {
    unsigned __int128 st = (static_cast<unsigned __int128>(origin_state[2 * pair + 1]) << 64) | origin_state[2 * pair];
    for(uint32_t bit = 0; offset != 0; ++bit, offset >>= 1)
        if(offset & 1u) st *= (static_cast<unsigned __int128>(mult_pow[2 * bit + 1]) << 64) | mult_pow[2 * bit];
    return Rng128{static_cast<uint64_t>(st), static_cast<uint64_t>(st >> 64)};
}

// What a round's candidate needs from its walk is only HOW MANY draws it takes.  Counting does not have to go step by step:
// a batch loads the thresholds of the cell the walk is at (for its state) and of the next K - 1 cells down the diagonal
// (for M), with the K draws; the K products and comparisons are independent of each other (a lone wavefront overlaps
// them: the step-by-step loop was ~55 dependent instructions per step); the number of leading "match" decisions says how
// many of the steps happened, and one more comparison of the first other decision says which gap state the walk is in
// then.  Same thresholds, same draws, same comparisons as table_walk: the same count.  ~60 instructions per batch of up
// to 16 steps instead of ~55 per step.
template <int K>
__device__ uint32_t table_walk_count(const Walker& w, const StepTable& tb, const float* __restrict__ at) {
    static_assert(K >= 2 && K <= 31, "decisions of a batch are bits of a word");
    uint32_t draws = 1;
    uint32_t i = w.la, j = w.lb;
    float score = 0.0f;  // (sample3 accumulates it; not a result here)
    Triple cur;
    w.cell(i, j, cur.m, cur.d, cur.in);
    bool cur_valid = true;
    int st;
    {
        const float top = fmaxf(fmaxf(cur.m, cur.d), cur.in);
        st = sample3(cur.m - top, cur.d - top, cur.in - top, *at++, score, w.exp_tab);
    }
    while(j > 0 || i > 0) {
        if(i >= 1 && j >= 1) {
            const bool m0 = st == COATI_HIP_OP_MATCH, d0 = st == COATI_HIP_OP_DEL;
            const uint32_t i1 = (m0 || d0) ? i - 1 : i, j1 = d0 ? j : j - 1;  // where this step leads
            const int64_t k0 = tb.diag(i, j), k1 = tb.diag(i1, j1);
            const uint32_t room = tb.holds(k1) ? min(i1, j1) : 0u;  // steps 1 .. room of the batch are at body cells (i1 - q + 1, j1 - q + 1) the table holds
            StepThresholds e[K];
            float dr[K];
            // (two batches in three end because all K steps were matches: the next one starts in state M, on the diagonal the
            // neighbouring candidates are reading too; a cell outside the table's band: its entry the way the table's were made)
            if(tb.holds(k0)) {
                e[0] = tb.thr != nullptr && m0 ? *tb.thr_at(i, j, k0) : *reinterpret_cast<const StepThresholds*>(tb.cell(i, k0) + static_cast<uint32_t>(st));
            } else {
                const StepEntry full = step_entry(w, i, j, st);
                e[0] = StepThresholds{full.scale, full.m, full.dm};
            }
            if(tb.thr != nullptr) {  // the next K - 1 diagonal cells' M thresholds: consecutive, descending
                const StepThresholds* dg = room > 0 ? tb.thr_at(i1, j1, k1) : tb.thr;
#pragma unroll
                for(int q = 1; q < K; ++q) e[q] = *(static_cast<uint32_t>(q) <= room ? dg - (q - 1) : tb.thr);
            } else {
                const StepEntry* dg = room > 0 ? tb.cell(i1, k1) + COATI_HIP_OP_MATCH : tb.steps;
                const uint64_t diag = static_cast<uint64_t>(tb.band.width) * 3;
#pragma unroll
                for(int q = 1; q < K; ++q) {
                    e[q] = *reinterpret_cast<const StepThresholds*>(static_cast<uint32_t>(q) <= room ? dg : tb.steps);
                    dg -= diag;
                }
            }
#pragma unroll
            for(int q = 0; q < K; ++q) dr[q] = at[q];  // (the table has slack behind every slice)
            uint32_t is_m = 0, is_d = 0;  // bit q: the draw of step q falls under the M threshold / under the D threshold
#pragma unroll
            for(int q = 0; q < K; ++q) {
                const float p = dr[q] * e[q].scale;
                is_m |= (p < e[q].m ? 1u : 0u) << q;
                is_d |= (p < e[q].dm ? 1u : 0u) << q;
            }
            // steps 0 .. lead - 1 chose M; step q >= 1 happens if every step before it chose M and its cell is a body cell
            const uint32_t lead = static_cast<uint32_t>(__builtin_ctz(~is_m | (1u << K)));
            const uint32_t c = min(min(lead + 1u, static_cast<uint32_t>(K)), room + 1u);  // steps of this batch that happen
            const uint32_t last = c - 1u;
            st = last < lead ? COATI_HIP_OP_MATCH : (((is_d >> last) & 1u) != 0 ? COATI_HIP_OP_DEL : COATI_HIP_OP_INS);
            i = i1 - last;
            j = j1 - last;
            draws += c;
            at += c;
            cur_valid = false;
            continue;
        }
        ++draws;
        const bool is_m = st == COATI_HIP_OP_MATCH, is_d = st == COATI_HIP_OP_DEL;
        const uint32_t pi = is_m ? i - 1 : (is_d ? i - 1 : i), pj = is_m ? j - 1 : (is_d ? j : j - 1);
        if(pi > i || pj > j) break;  // (as in sample_walk: only with a table of -inf / NaN weights)
        {
            // a margin cell: del_del / ins_ins are copies of the margin D / I (init_margins, align_pair.hpp:108-111)
            if(!cur_valid) w.pred(i, j, cur.m, cur.d, cur.in);
            Triple t{kLowest, kLowest, kLowest};
            w.pred(pi, pj, t.m, t.d, t.in);
            const float top = is_m ? cur.m : (is_d ? cur.d : cur.in);
            float e0 = kLowest, e1 = kLowest, e2 = kLowest;
            if(!is_m) {
                float mm0, dm0, im0;
                margin_mdi(w.k, 1u, i, j, mm0, dm0, im0);
                if(is_d)
                    e1 = dm0;
                else
                    e2 = im0;
            }
            const float l1 = (is_m || is_d) ? e1 - top : -__builtin_inff();
            st = sample3(e0 - top, l1, e2 - top, *at++, score, w.exp_tab);
            cur = t;
            cur_valid = true;
        }
        i = pi;
        j = pj;
    }
    return draws;
}

// a round's candidates: how many draws does a sample that starts `offset` draws after the chunk origin take?
__global__ __launch_bounds__(64) void spec_len_kernel(const float* __restrict__ table, GapConsts k, const PairDesc* __restrict__ pairs,
                                                      const uint64_t* __restrict__ tab_off, const uint8_t* __restrict__ a_cat,
                                                      const uint8_t* __restrict__ b_cat, const float* __restrict__ mdi,
                                                      const StepEntry* __restrict__ steps, uint32_t band_half, const uint64_t* __restrict__ origin_state,
                                                      const uint64_t* __restrict__ mult_pow, const SpecCandidate* __restrict__ cands,
                                                      uint32_t n_cands, uint32_t* __restrict__ c_draws) {
    __shared__ uint64_t exp_tab[32];
    load_exp_table(exp_tab, threadIdx.x);
    __syncthreads();
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if(idx >= n_cands) return;
    const SpecCandidate cd = cands[idx];
    const PairDesc pd = pairs[cd.pair];
    Rng128 rng = rng_jump(origin_state, mult_pow, cd.pair, cd.offset);
    const Walker w{k, 1u, pd.la, pd.lb, k.ge * 0.0f, k.ge * 1.0f, table + static_cast<size_t>(pd.table) * kTabFloats, a_cat + pd.a_off, b_cat + pd.b_off,
                   mdi, pd, exp_tab};
    float score;
    uint32_t draws;
    (void)table_walk<false>(w, StepTable{steps + tab_off[cd.pair], nullptr, step_band(pd.la, pd.lb, band_half)}, DrawsFromRng{rng}, nullptr, 0, score, draws);
    c_draws[idx] = draws;
}

// ---- device rounds (round 4).  The host loop of sample_host.hip, on the device: per round ONE plan launch (windows of
// every unfinished pair, under the same rules), ONE walk launch over the whole candidate array (a pair's candidates sit in
// its share: index = rank * share + local), ONE chain launch (a workgroup per pair follows the true offsets through the
// draw counts, staged through LDS, writes every resolved sample's stream offset and updates the pair's estimate).  The
// host only enqueues rounds and looks at the number of unfinished pairs every few rounds: a round cost ~0.27 ms of
// host work and copies beside ~0.45 ms of walks.
__global__ __launch_bounds__(64) void spec_plan_kernel(const PairDesc* __restrict__ pairs, uint32_t n_pairs, uint32_t n_samples, uint32_t max_cands,
                                                       uint32_t max_width, double z, SpecPairState* __restrict__ states, SpecWindow* __restrict__ windows,
                                                       uint32_t* __restrict__ rank_pair, SpecRound* __restrict__ round) {
    // one wavefront per pair: how many pairs are unfinished, how many of them come before this one (its rank = its share of
    // the candidate array), then the pair's windows, eight per lane, with a prefix sum over their sizes
    __shared__ uint32_t s_count[kSpecChunkMax];
    const uint32_t p = blockIdx.x, lane = threadIdx.x;
    uint32_t active = 0, before = 0;
    for(uint32_t q = lane; q < n_pairs; q += kWave) {
        const uint32_t unfinished = states[q].done < n_samples ? 1u : 0u;
        active += unfinished;
        before += q < p ? unfinished : 0u;
    }
    for(int sh = 1; sh < kWave; sh <<= 1) {
        active += __shfl_xor(active, sh);
        before += __shfl_xor(before, sh);
    }
    // (pairs of a round: as many as have a share of the candidates AND a slice of the draw table that holds one whole walk)
    const uint32_t share = max(max_cands / max(active, 1u), 1u);
    const uint32_t ranked = min(min(active, max_cands / share), max(kSpecDrawFloats / ((max_width + 2u + 16u + 63u) / 64u * 64u), 1u));
    const uint32_t slice = kSpecDrawFloats / max(ranked, 1u) / 64u * 64u;  // floats of the draw table per pair of the round
    if(p == 0 && lane == 0) round->active = active, round->share = share, round->ranked = ranked, round->slice = slice;
    SpecPairState& s = states[p];
    const bool mine = s.done < n_samples && before < ranked;  // (more unfinished pairs than a round holds: the others wait)
    if(!mine) {
        if(lane == 0) s.n_cands = 0, s.n_windows = 0, s.n_draws = 0;
        return;
    }
    const double width = static_cast<double>(pairs[p].la) + static_cast<double>(pairs[p].lb);
    const uint32_t remaining = n_samples - s.done;
    const uint32_t want = s.cnt == 0 ? 8u : (s.cnt < 8 ? 16u : kSpecChunkMax);
    const uint32_t chunk = min(remaining, want);
    double mean = s.mean;
    if(s.cnt == 0) mean = 1.0 + static_cast<double>(max(pairs[p].la, pairs[p].lb)) + 0.005 * width;  // (the prior: sample_host.hip)
    const double sigma = s.cnt >= 2 ? sqrt(s.m2 / (s.cnt - 1)) * (1.0 + 4.0 / s.cnt) + 1.0 : 0.02 * width + 2.0;
    constexpr uint32_t kPer = kSpecChunkMax / kWave;  // windows lane * kPer .. lane * kPer + kPer - 1
    uint32_t lo_w[kPer], hi_w[kPer];
    unsigned long long mine_sum = 0;
    constexpr uint32_t kUnusable = 0x80000000u;  // (an unusable window is "larger than any share": it ends the chunk)
#pragma unroll
    for(uint32_t t = 0; t < kPer; ++t) {
        const uint32_t j = lane * kPer + t;
        const long long center = llrint(j * mean);
        const long long half = j == 0 ? 0 : static_cast<long long>(ceil(z * sigma * sqrt(static_cast<double>(j)))) + 2;
        const long long lo = max(center - half, static_cast<long long>(j)), hi = max(center + half, lo);
        // (a walk from offset hi takes at most la + lb + 1 draws: they must lie in the pair's slice of the draw table)
        const bool usable = j < chunk && hi <= 0x7ffffff0ll && hi + static_cast<long long>(width) + 2 <= static_cast<long long>(slice);
        lo_w[t] = static_cast<uint32_t>(lo), hi_w[t] = static_cast<uint32_t>(hi);
        const uint32_t count = usable ? static_cast<uint32_t>(hi - lo + 1) : kUnusable;
        s_count[j] = count;
        mine_sum += count;
    }
    unsigned long long incl = mine_sum;  // inclusive prefix over the lanes
    for(int sh = 1; sh < kWave; sh <<= 1) {
        const unsigned long long up = __shfl_up(incl, sh);
        if(lane >= static_cast<uint32_t>(sh)) incl += up;
    }
    unsigned long long first = incl - mine_sum;
    SpecWindow* w = windows + static_cast<uint64_t>(p) * kSpecChunkMax;
    uint32_t last_in = 0, used = 0, last_hi = 0;  // the windows that fit the share are a prefix (sizes are positive; window 0 is one candidate: always in)
#pragma unroll
    for(uint32_t t = 0; t < kPer; ++t) {
        const uint32_t j = lane * kPer + t, count = s_count[j];
        const bool fits = count != kUnusable && (j == 0 || first + count <= share);
        if(fits) {
            w[j] = SpecWindow{static_cast<uint32_t>(first), lo_w[t], hi_w[t]};
            last_in = j + 1;
            used = static_cast<uint32_t>(first) + count;
            last_hi = hi_w[t];
        }
        first += count;
    }
    for(int sh = 1; sh < kWave; sh <<= 1) {
        last_in = max(last_in, __shfl_xor(last_in, sh));
        used = max(used, __shfl_xor(used, sh));
        last_hi = max(last_hi, __shfl_xor(last_hi, sh));
    }
    if(lane == 0) {
        s.rank = before;
        rank_pair[before] = p;
        s.n_cands = min(used, share);
        s.n_windows = last_in;
        s.n_draws = min(slice, last_hi + static_cast<uint32_t>(width) + 2u);
    }
}

// the round's draws: slice r of the table = the stream of pair rank_pair[r] from its origin on, as far as the round's walks
// can read (n_draws); a thread jumps to its 64 draws and steps through them
__global__ __launch_bounds__(64) void spec_draws_kernel(const uint64_t* __restrict__ state0, const uint64_t* __restrict__ mult_pow,
                                                        const SpecPairState* __restrict__ states, const uint32_t* __restrict__ rank_pair,
                                                        const SpecRound* __restrict__ round, float* __restrict__ draw_table) {
    const uint32_t slice = round->slice, ranked = round->ranked;
    if(slice == 0) return;
    const uint64_t pos = (blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x) * 64u;
    const uint32_t r = static_cast<uint32_t>(pos / slice), at = static_cast<uint32_t>(pos % slice);
    if(r >= ranked) return;
    const uint32_t pair = rank_pair[r];
    const SpecPairState& s = states[pair];
    if(at >= s.n_draws) return;
    Rng128 rng = rng_jump(state0, mult_pow, pair, s.origin + at);
    float* out = draw_table + static_cast<uint64_t>(r) * slice + at;
    const uint32_t n = min(64u, s.n_draws - at);
    for(uint32_t q = 0; q < n; ++q) out[q] = rng_f24(rng);
}

__global__ __launch_bounds__(64) void spec_len_round_kernel(const float* __restrict__ table, GapConsts k, const PairDesc* __restrict__ pairs,
                                                            const uint64_t* __restrict__ tab_off, const uint8_t* __restrict__ a_cat,
                                                            const uint8_t* __restrict__ b_cat, const float* __restrict__ mdi,
                                                            const StepEntry* __restrict__ steps, uint32_t band_half, const uint64_t* __restrict__ state0,
                                                            const uint64_t* __restrict__ mult_pow, const SpecPairState* __restrict__ states,
                                                            const SpecWindow* __restrict__ windows, const uint32_t* __restrict__ rank_pair,
                                                            const SpecRound* __restrict__ round, const float* __restrict__ draw_table,
                                                            const uint64_t* __restrict__ thr_off, const StepThresholds* __restrict__ thr_m,
                                                            uint32_t* __restrict__ c_draws) {
    __shared__ uint64_t exp_tab[32];
    const uint32_t share = round->share, ranked = round->ranked;
    const uint64_t idx = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if(static_cast<uint64_t>(blockIdx.x) * blockDim.x >= static_cast<uint64_t>(ranked) * share) return;  // (the whole workgroup: before the barrier)
    load_exp_table(exp_tab, threadIdx.x);
    __syncthreads();
    const uint32_t r = static_cast<uint32_t>(idx / share), local = static_cast<uint32_t>(idx % share);
    if(r >= ranked) return;
    const uint32_t pair = rank_pair[r];
    const SpecPairState& s = states[pair];
    if(local >= s.n_cands) return;
    const SpecWindow* w = windows + static_cast<uint64_t>(pair) * kSpecChunkMax;
    uint32_t lo_w = 0, hi_w = s.n_windows;  // the last window whose first candidate is <= local
    while(hi_w - lo_w > 1) {
        const uint32_t mid = (lo_w + hi_w) / 2;
        if(w[mid].first <= local)
            lo_w = mid;
        else
            hi_w = mid;
    }
    const uint32_t offset = w[lo_w].lo + (local - w[lo_w].first);  // draws after the pair's origin: where this walk starts in its slice
    const PairDesc pd = pairs[pair];
    const Walker wk{k, 1u, pd.la, pd.lb, k.ge * 0.0f, k.ge * 1.0f, table + static_cast<size_t>(pd.table) * kTabFloats, a_cat + pd.a_off, b_cat + pd.b_off,
                    mdi, pd, exp_tab};
    const StepTable tb{steps + tab_off[pair], thr_m != nullptr ? thr_m + thr_off[pair] : nullptr, step_band(pd.la, pd.lb, band_half)};
    const uint32_t draws = table_walk_count<16>(wk, tb, draw_table + static_cast<uint64_t>(r) * round->slice + offset);
    c_draws[idx] = draws;
}

constexpr uint32_t kChainLds = 12288;  // draw counts staged per batch of windows (48 KB)
__global__ __launch_bounds__(256) void spec_chain_kernel(uint32_t n_samples, SpecPairState* __restrict__ states, const SpecWindow* __restrict__ windows,
                                                         const SpecRound* __restrict__ round, const uint32_t* __restrict__ c_draws,
                                                         uint64_t* __restrict__ sample_off) {
    __shared__ uint32_t lds[kChainLds];
    __shared__ SpecWindow w[kSpecChunkMax];  // (the chain reads a window per sample: from LDS, not from memory)
    __shared__ uint32_t s_next, s_alive, s_off;
    const uint32_t pair = blockIdx.x;
    SpecPairState& s = states[pair];
    const uint32_t nw = s.n_windows, n_cands = s.n_cands;
    if(nw == 0 || n_cands == 0) return;
    for(uint32_t q = threadIdx.x; q < nw; q += blockDim.x) w[q] = windows[static_cast<uint64_t>(pair) * kSpecChunkMax + q];
    const uint32_t* draws = c_draws + static_cast<uint64_t>(s.rank) * round->share;
    // the pair's state, advanced by thread 0 only; the running estimate of draws per sample takes this round's
    // observations as integer sums (one double division per round, not per sample: a lone lane pays ~5 cycles per instruction)
    const uint64_t origin = s.origin;
    uint32_t done = s.done;
    unsigned long long sum_x = 0, sum_xx = 0;
    uint32_t seen = 0;
    if(threadIdx.x == 0) s_next = 0, s_alive = 1, s_off = 0;
    __syncthreads();
    for(;;) {
        const uint32_t j0 = s_next;
        if(j0 >= nw || s_alive == 0) break;
        // windows j0 .. j1-1 fit the staging area together (binary search on `first`: monotone); a window wider than the
        // area goes alone and is read from memory
        const uint32_t first0 = w[j0].first;
        uint32_t lo_w = j0, hi_w = nw;  // the last window that still ends inside first0 + kChainLds
        while(hi_w - lo_w > 1) {
            const uint32_t mid = (lo_w + hi_w) / 2;
            if(w[mid].first + (w[mid].hi - w[mid].lo + 1) - first0 <= kChainLds)
                lo_w = mid;
            else
                hi_w = mid;
        }
        const uint32_t j1 = lo_w + 1;
        const uint32_t end = min(w[j1 - 1].first + (w[j1 - 1].hi - w[j1 - 1].lo + 1), n_cands);
        const bool staged = end - first0 <= kChainLds;
        if(staged) {  // (eight loads in flight per thread: one load per trip of a plain loop was 32 memory round trips, most of the kernel's 48 us)
            const uint32_t n_stage = end - first0;
            for(uint32_t q0 = 0; q0 < n_stage; q0 += 8 * blockDim.x) {
                uint32_t v[8];
#pragma unroll
                for(uint32_t u = 0; u < 8; ++u) {
                    const uint32_t q = q0 + u * blockDim.x + threadIdx.x;
                    v[u] = q < n_stage ? draws[first0 + q] : 0u;
                }
#pragma unroll
                for(uint32_t u = 0; u < 8; ++u) {
                    const uint32_t q = q0 + u * blockDim.x + threadIdx.x;
                    if(q < n_stage) lds[q] = v[u];
                }
            }
        }
        __syncthreads();
        if(threadIdx.x == 0) {
            uint32_t off = s_off, j = j0;
            bool alive = true;
            for(; j < j1; ++j) {
                const SpecWindow wj = w[j];
                if(off < wj.lo || off > wj.hi || wj.first + (off - wj.lo) >= n_cands) {  // not speculated: first sample of the next round
                    alive = false;
                    break;
                }
                const uint32_t c = wj.first + (off - wj.lo);
                uint32_t x;
                if(staged) {
                    x = lds[c - first0];
                } else {  // (a window wider than the staging area: alone in its batch.  A branch, not a select: as a select the
                    // compiler issued the memory load for every sample, ~0.6 us each, most of this kernel's time)
                    x = __hip_atomic_load(draws + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                sample_off[static_cast<uint64_t>(pair) * n_samples + done] = origin + off;
                sum_x += x;
                sum_xx += static_cast<unsigned long long>(x) * x;
                seen += 1;
                off += x;
                done += 1;
            }
            s_off = off, s_next = j, s_alive = alive ? 1u : 0u;
        }
        __syncthreads();
    }
    if(threadIdx.x == 0) {
        s.origin = origin + s_off;
        s.done = done;
        if(seen > 0) {  // Welford's update for a batch (Chan et al.): the round's mean and sum of squared deviations joined to the estimate
            const double nb = static_cast<double>(seen), mean_b = static_cast<double>(sum_x) / nb;
            const double m2_b = fmax(static_cast<double>(sum_xx) - nb * mean_b * mean_b, 0.0);
            const double na = static_cast<double>(s.cnt);  // (0: the prior has served, the estimate starts from the observations)
            const double delta = mean_b - (s.cnt == 0 ? 0.0 : s.mean), nab = na + nb;
            s.mean = s.cnt == 0 ? mean_b : s.mean + delta * nb / nab;
            s.m2 = s.cnt == 0 ? m2_b : s.m2 + m2_b + delta * delta * na * nb / nab;
            s.cnt = s.cnt + seen;
        }
    }
}

// the chain is known: every (pair, sample) from its exact offset (draws after the pair's ORIGINAL state), results written
__global__ __launch_bounds__(64) void final_walk_kernel(const float* __restrict__ table, GapConsts k, const PairDesc* __restrict__ pairs,
                                                        const uint64_t* __restrict__ tab_off, const uint8_t* __restrict__ a_cat,
                                                        const uint8_t* __restrict__ b_cat, const float* __restrict__ mdi,
                                                        const StepEntry* __restrict__ steps, uint32_t band_half, const uint64_t* __restrict__ start_state,
                                                        const uint64_t* __restrict__ mult_pow, const uint64_t* __restrict__ sample_offset,
                                                        const uint64_t* __restrict__ sample_base, uint32_t n_pairs, uint32_t n_samples,
                                                        uint8_t* __restrict__ ops, uint64_t* __restrict__ ops_start, uint32_t* __restrict__ ops_len,
                                                        float* __restrict__ log_weights) {
    __shared__ uint64_t exp_tab[32];
    load_exp_table(exp_tab, threadIdx.x);
    __syncthreads();
    const uint64_t idx = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if(idx >= static_cast<uint64_t>(n_pairs) * n_samples) return;
    const uint32_t pair = static_cast<uint32_t>(idx / n_samples), n = static_cast<uint32_t>(idx % n_samples);
    const PairDesc pd = pairs[pair];
    Rng128 rng = rng_jump(start_state, mult_pow, pair, sample_offset[idx]);
    const Walker w{k, 1u, pd.la, pd.lb, k.ge * 0.0f, k.ge * 1.0f, table + static_cast<size_t>(pd.table) * kTabFloats, a_cat + pd.a_off, b_cat + pd.b_off,
                   mdi, pd, exp_tab};
    const uint64_t width = static_cast<uint64_t>(pd.la) + pd.lb, slot = sample_base[pair] + n * width;
    float score;
    uint32_t draws;
    const uint64_t pos = table_walk<true>(w, StepTable{steps + tab_off[pair], nullptr, step_band(pd.la, pd.lb, band_half)}, DrawsFromRng{rng}, ops, slot, score, draws);
    ops_start[idx] = pos;
    ops_len[idx] = static_cast<uint32_t>(slot + width - pos);
    log_weights[idx] = score;
}

// Copy the candidates that turned out to be the true samples into the result arrays
// (one workgroup per sample).
__global__ __launch_bounds__(64) void spec_commit_kernel(const SpecCommit* __restrict__ commits, uint32_t n_commits,
                                                         const uint8_t* __restrict__ tmp_ops,
                                                         const uint64_t* __restrict__ c_start,
                                                         const uint32_t* __restrict__ c_len,
                                                         const float* __restrict__ c_lw, uint8_t* __restrict__ ops,
                                                         uint64_t* __restrict__ ops_start, uint32_t* __restrict__ ops_len,
                                                         float* __restrict__ log_weights) {
    const uint32_t q = blockIdx.x;
    if(q >= n_commits) return;
    const SpecCommit cm = commits[q];
    const uint32_t len = c_len[cm.cand];
    const uint64_t src = c_start[cm.cand], dst = cm.slot_end - len;
    for(uint32_t t = threadIdx.x; t < len; t += blockDim.x) ops[dst + t] = tmp_ops[src + t];
    if(threadIdx.x == 0) {
        ops_start[cm.out_index] = dst;
        ops_len[cm.out_index] = len;
        log_weights[cm.out_index] = c_lw[cm.cand];
    }
}

// Debug/parity: the first n f24() draws of a stream (one thread).
__global__ void rng_f24_kernel(uint64_t lo, uint64_t hi, uint32_t n, float* __restrict__ out) {
    Rng128 r{lo, hi};
    for(uint32_t q = 0; q < n; ++q) out[q] = rng_f24(r);
}

}  // namespace

hipError_t launch_spec_walk(const BatchDeviceView& v, const uint64_t* origin_state, const uint64_t* mult_pow,
                            const SpecCandidate* cands, uint32_t n_cands, uint8_t* tmp_ops, uint64_t* c_start,
                            uint32_t* c_len, float* c_lw, uint32_t* c_draws, hipStream_t stream) {
    if(n_cands == 0) return hipSuccess;
    hipLaunchKernelGGL(spec_walk_kernel, dim3((n_cands + 63) / 64), dim3(64), 0, stream, v.table, v.k, v.gap_len, v.pairs,
                       v.a_cat, v.b_cat, v.mdi, origin_state, mult_pow, cands, n_cands, tmp_ops, c_start, c_len, c_lw,
                       c_draws);
    return hipGetLastError();
}

hipError_t launch_spec_commit(const SpecCommit* commits, uint32_t n_commits, const uint8_t* tmp_ops,
                              const uint64_t* c_start, const uint32_t* c_len, const float* c_lw, uint8_t* ops,
                              uint64_t* ops_start, uint32_t* ops_len, float* log_weights, hipStream_t stream) {
    if(n_commits == 0) return hipSuccess;
    hipLaunchKernelGGL(spec_commit_kernel, dim3(n_commits), dim3(64), 0, stream, commits, n_commits, tmp_ops, c_start,
                       c_len, c_lw, ops, ops_start, ops_len, log_weights);
    return hipGetLastError();
}

uint64_t step_entry_bytes() { return sizeof(StepEntry); }
hipError_t launch_step_table(const BatchDeviceView& v, const uint64_t* tab_off, uint64_t max_cells, uint32_t band, void* steps, const uint64_t* thr_off,
                             void* thr_m, hipStream_t stream) {
    if(v.n_pairs == 0 || max_cells == 0) return hipSuccess;
    const uint32_t gx = static_cast<uint32_t>(std::min<uint64_t>((max_cells + 255) / 256, 4096));
    hipLaunchKernelGGL(step_table_kernel, dim3(gx, v.n_pairs), dim3(256), 0, stream, v.table, v.k, v.pairs, tab_off, v.n_pairs, band, v.a_cat, v.b_cat, v.mdi,
                       static_cast<StepEntry*>(steps), thr_off, static_cast<StepThresholds*>(thr_m));
    return hipGetLastError();
}
hipError_t launch_spec_len(const BatchDeviceView& v, const uint64_t* tab_off, uint32_t band, const void* steps, const uint64_t* origin_state,
                           const uint64_t* mult_pow, const SpecCandidate* cands, uint32_t n_cands, uint32_t* c_draws, hipStream_t stream) {
    if(n_cands == 0) return hipSuccess;
    hipLaunchKernelGGL(spec_len_kernel, dim3((n_cands + 63) / 64), dim3(64), 0, stream, v.table, v.k, v.pairs, tab_off, v.a_cat, v.b_cat, v.mdi,
                       static_cast<const StepEntry*>(steps), band, origin_state, mult_pow, cands, n_cands, c_draws);
    return hipGetLastError();
}
hipError_t launch_spec_round(const BatchDeviceView& v, const uint64_t* tab_off, uint32_t band, const void* steps, const uint64_t* state0, const uint64_t* mult_pow,
                             uint32_t n_samples, uint32_t max_cands, uint32_t max_width, double z, SpecPairState* states, SpecWindow* windows,
                             uint32_t* rank_pair, SpecRound* round, float* draw_table, const uint64_t* thr_off, const void* thr_m, uint32_t* c_draws,
                             uint64_t* sample_off, hipStream_t stream) {
    if(v.n_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(spec_plan_kernel, dim3(v.n_pairs), dim3(kWave), 0, stream, v.pairs, v.n_pairs, n_samples, max_cands, max_width, z, states, windows, rank_pair, round);
    hipLaunchKernelGGL(spec_draws_kernel, dim3(kSpecDrawFloats / 64 / 64), dim3(64), 0, stream, state0, mult_pow, states, rank_pair, round, draw_table);
    hipLaunchKernelGGL(spec_len_round_kernel, dim3((max_cands + 63) / 64), dim3(64), 0, stream, v.table, v.k, v.pairs, tab_off, v.a_cat, v.b_cat, v.mdi,
                       static_cast<const StepEntry*>(steps), band, state0, mult_pow, states, windows, rank_pair, round, draw_table, thr_off,
                       static_cast<const StepThresholds*>(thr_m), c_draws);
    hipLaunchKernelGGL(spec_chain_kernel, dim3(v.n_pairs), dim3(256), 0, stream, n_samples, states, windows, round, c_draws, sample_off);
    return hipGetLastError();
}
hipError_t launch_final_walk(const BatchDeviceView& v, const uint64_t* tab_off, uint32_t band, const void* steps, const uint64_t* start_state,
                             const uint64_t* mult_pow, const uint64_t* sample_offset, const uint64_t* sample_base, uint32_t n_samples, uint8_t* ops, uint64_t* ops_start,
                             uint32_t* ops_len, float* log_weights, hipStream_t stream) {
    const uint64_t walkers = static_cast<uint64_t>(v.n_pairs) * n_samples;
    if(walkers == 0) return hipSuccess;
    hipLaunchKernelGGL(final_walk_kernel, dim3(static_cast<uint32_t>((walkers + 63) / 64)), dim3(64), 0, stream, v.table, v.k, v.pairs, tab_off, v.a_cat,
                       v.b_cat, v.mdi, static_cast<const StepEntry*>(steps), band, start_state, mult_pow, sample_offset, sample_base, v.n_pairs, n_samples, ops,
                       ops_start, ops_len, log_weights);
    return hipGetLastError();
}

hipError_t launch_rng_f24(const uint64_t state[2], uint32_t n, float* d_out, hipStream_t stream) {
    hipLaunchKernelGGL(rng_f24_kernel, dim3(1), dim3(1), 0, stream, state[0], state[1], n, d_out);
    return hipGetLastError();
}

hipError_t launch_sampleback(const BatchDeviceView& v, uint32_t n_samples, bool independent, uint64_t* rng_states,
                             const uint64_t* sample_base, uint8_t* ops, uint64_t* ops_start, uint32_t* ops_len,
                             float* log_weights, hipStream_t stream) {
    const uint64_t walkers = independent ? static_cast<uint64_t>(v.n_pairs) * n_samples : v.n_pairs;
    if(walkers == 0) return hipSuccess;
    const uint32_t grid = static_cast<uint32_t>((walkers + 63) / 64);
    hipLaunchKernelGGL(sampleback_kernel, dim3(grid), dim3(64), 0, stream, v.table, v.k, v.gap_len, v.pairs, v.n_pairs,
                       n_samples, independent ? 1 : 0, v.a_cat, v.b_cat, v.mdi, rng_states, sample_base, ops, ops_start,
                       ops_len, log_weights);
    return hipGetLastError();
}

// ---- the results leave the device PACKED.  The walkers write a sample's ops right-aligned into its slot of la + lb bytes
// (they walk from the last column to the first and do not know the length beforehand); a sample of a 1 kb pair fills
// ~1 012 of its 2 000 bytes, so half of what a download of the slots moves is padding.  Two launches pack them: an
// exclusive scan over the lengths (one workgroup: tens of thousands of samples) and a copy, one workgroup per sample.
namespace {
__global__ __launch_bounds__(1024) void ops_scan_kernel(const uint32_t* __restrict__ len, uint64_t n, uint64_t* __restrict__ packed_off,
                                                        uint64_t* __restrict__ total) {
    // a thread sums its run of consecutive lengths, one scan over the 1 024 sums, then the run again with its prefix
    __shared__ uint64_t part[1024];
    const uint64_t per = (n + 1023) / 1024, q0 = threadIdx.x * per, q1 = q0 + per < n ? q0 + per : n;
    uint64_t mine = 0;
    for(uint64_t q = q0; q < q1; ++q) mine += len[q];
    part[threadIdx.x] = mine;
    __syncthreads();
    for(uint32_t sh = 1; sh < 1024; sh <<= 1) {  // inclusive scan of the sums
        const uint64_t add = threadIdx.x >= sh ? part[threadIdx.x - sh] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint64_t at = part[threadIdx.x] - mine;
    for(uint64_t q = q0; q < q1; ++q) {
        packed_off[q] = at;
        at += len[q];
    }
    if(threadIdx.x == 1023) *total = part[1023];
}
__global__ __launch_bounds__(64) void ops_pack_kernel(const uint8_t* __restrict__ ops, const uint64_t* __restrict__ start, const uint32_t* __restrict__ len,
                                                      const uint64_t* __restrict__ packed_off, uint8_t* __restrict__ packed) {
    const uint64_t q = blockIdx.x;
    const uint8_t* src = ops + start[q];
    uint8_t* dst = packed + packed_off[q];
    for(uint32_t t = threadIdx.x; t < len[q]; t += blockDim.x) dst[t] = src[t];
}
}  // namespace
hipError_t launch_ops_pack(const uint8_t* ops, const uint64_t* start, const uint32_t* len, uint64_t n, uint64_t* packed_off, uint64_t* total, uint8_t* packed,
                           hipStream_t stream) {
    if(n == 0) return hipSuccess;
    hipLaunchKernelGGL(ops_scan_kernel, dim3(1), dim3(1024), 0, stream, len, n, packed_off, total);
    hipLaunchKernelGGL(ops_pack_kernel, dim3(static_cast<uint32_t>(n)), dim3(64), 0, stream, ops, start, len, packed_off, packed);
    return hipGetLastError();
}

}  // namespace coati_hip_detail
