// forward_l1: the Forward fill (log semiring) for gap_len == 1, in the wavefront shape of
// viterbi_l1: persistent wavefronts, one strip (64 * W descendant columns, W = 16, 8 or 4) of one
// pair at a time per wavefront, a lane owns W columns, one row of skew per lane, DPP hand-off,
// strips of a pair pipelined across wavefronts through boundary arrays in HBM.
//
// What it replaces in the reference:
//   forward -> forward_impl<log, align_pair_work_t>   src/lib/align_pair.cc:62-139,149
//   semiring::log::plus, log_sum_exp, log1p_exp       src/include/coati/semiring.hpp:86-121,
//                                                     src/include/coati/utils.hpp:134-156
//
// Output: fp32 M/D/I of every body cell in HBM (12 B/cell, M, D, I of a cell adjacent; layout in
// common.hpp) for sampleback, and the terminal-adjusted last cell per pair.  The reference's eight
// edge matrices (align_pair.hpp:94-103) are not stored; the sampler recomputes what it needs.
//
// Every cell evaluates the reference's expressions in the reference's order
// (align_pair.cc:97-124).  `plus(a, b) = max(a, b) + log1p(exp(-|a - b|))` comes in two builds of
// the kernel (template parameter kFast):
//   kFast = false (default): glibc's expf / log1pf restated for the device (glibc_math.hpp,
//     log_plus_exact in common.hpp) -- every M/D/I value has the CPU's bits;
//   kFast = true (COATI_HIP_FORWARD_FAST=1): the hardware exp2/log2 (v_exp_f32, v_log_f32, ~1 ulp).
//     log1p(e) for e in [0, 1] is taken as log(u) + (e - (u - 1)) with u = fl(1 + e): the second
//     term is the exact rounding residue of 1 + e, so the result degrades gracefully to `e` when
//     1 + e rounds to 1 -- the same value utils.hpp:142-144 returns for y <= -16.  Absolute error
//     of one plus() below 1e-7; results within 1e-5 relative of the CPU's, not bit-identical.
#include "common.hpp"

#include <algorithm>
#include <cstdlib>

namespace coati_hip_detail {
namespace {

// hand-over granule of strips of 1 or 2 columns per lane (round 5, same box: 16 pairs 3.18 -> 3.13 ms, one pair 3.12 -> 3.04 with 8
// instead of 16 rows; 4 rows: 3.27 / 3.17 -- a publish drains the strip's M/D/I stores)
constexpr uint32_t kFwdSubRowsNarrow = 8;

template <int W>
struct FwdLane {
    float M[W], D[W], I[W];        // the lane's W columns of the row it processed last
    float oM, oD, oI;              // column W-1 of the row before that: the right neighbour's diagonal
};

struct FwdCtx {
    GapConsts k;
    GapVec kv;  // the same constants in VGPRs for the per-cell adds (an SGPR operand costs an add the slow issue cadence: common.hpp)
    uint32_t la, col0, nsteps, pair;
    int lane, last_lane, last_c;
    bool last_strip;
    float* mout;
    float* bnd_out;   // [la + 1][3]: M, D, I of this strip's last column; entry 0 = the margin row
    float* final_mdi;
    const uint64_t* exp_tab;  // LDS copy of expf's table (glibc_math.hpp)
};

template <bool kFast>
__device__ __forceinline__ float plus2(const FwdCtx& cx, float a, float b) {
    if constexpr(kFast) return log_plus(a, b);
    return log_plus_exact(a, b, cx.exp_tab);
}

__device__ __forceinline__ void store_through(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Up to 64 wavefront steps (see viterbi_l1.hip: run_chunk).  ch*: what lane 0 needs at step
// kbase + l, held by lane l: diagonal cell (M, D, I) and left cell (M, I) of column col0 - 1.
template <int W, bool kFirst, bool kFast>
__device__ __forceinline__ void fwd_step(const FwdCtx& cx, FwdLane<W>& st, uint32_t& arow, float (&s)[W],
                                         const uint32_t (&boff)[W], uint32_t kbase, uint32_t kk, uint32_t a_chunk,
                                         float chDM, float chDD, float chDI, float chLM, float chLI,
                                         const char* tab_bytes) {
    const GapConsts& k = cx.k;
    const GapVec& kv = cx.kv;
    const int lane = cx.lane;
    {
        const uint32_t kstep = kbase + kk;
        if constexpr(kFirst) {
            if(kk == static_cast<uint32_t>(lane)) {
                // the lane starts: the row above is matrix row 0 (align_pair.cc:88-90):
                // M = D = lowest, I = go + ge*float(j-1)
                uint32_t bj0 = cx.col0 + lane * W;
                asm volatile("" : "+v"(bj0));
#pragma unroll
                for(int c = 0; c < W; ++c) {
                    st.M[c] = kLowest;
                    st.D[c] = kLowest;
                    st.I[c] = k.go + k.ge * static_cast<float>(bj0 + c);
                }
                if(!cx.last_strip && lane == kWave - 1) {
                    store_through(&cx.bnd_out[0], kLowest);
                    store_through(&cx.bnd_out[1], kLowest);
                    store_through(&cx.bnd_out[2], st.I[W - 1]);
                }
            }
        }
        // ---- hand-off from the left neighbour (full exec)
        float dgM = shift_in(st.oM, read_lane(chDM, kk));
        float dgD = shift_in(st.oD, read_lane(chDD, kk));
        float dgI = shift_in(st.oI, read_lane(chDI, kk));
        float lfM = shift_in(st.M[W - 1], read_lane(chLM, kk));
        float lfI = shift_in(st.I[W - 1], read_lane(chLI, kk));
        const uint32_t arow_next = shift_in(arow, read_lane(a_chunk, kk));
        st.oM = st.M[W - 1];
        st.oD = st.D[W - 1];
        st.oI = st.I[W - 1];
        float* dst = cx.mout + static_cast<uint64_t>(kstep) * (3 * W * kWave);
#pragma unroll
        for(int c = 0; c < W; ++c) {
            const float sc = s[c];
            // gather the next step's score now (consumed a step later)
            s[c] = *reinterpret_cast<const float*>(tab_bytes + arow_next + boff[c]);
            const float upM = st.M[c], upD = st.D[c], upI = st.I[c];
            // align_pair.cc:97-119 with look_back 1 (power(ge, 0) = -0.0f is the additive identity)
            const float m2m = ((dgM + kv.ng) + kv.ng) + sc;
            const float d2m = (dgD + kv.gs) + sc;
            const float i2m = ((dgI + kv.gs) + kv.ng) + sc;
            const float m2d = (upM + kv.ng) + kv.go;
            const float i2d = (upI + kv.gs) + kv.go;
            const float d2d = upD + kv.ge;
            const float m2i = lfM + kv.go;
            const float i2i = lfI + kv.ge;
            float M, D;
            if constexpr(kFast) {
                M = log_plus3(m2m, d2m, i2m);
                D = log_plus3(m2d, d2d, i2d);
            } else if constexpr(W <= 4) {
                // the M and the D sums do not depend on each other: their two dependent chains side by side (common.hpp).
                // Narrow strips only -- few pairs, every wavefront alone on its SIMD: 16 pairs 3.41 -> 3.27 ms (round 5, same
                // box, alternating); with four wavefronts per SIMD the others already fill the chain's latencies and the joint
                // choice of log1pf's short route is taken less often: 6 144 pairs 60.6 -> 62.1 ms, so the bulk shape keeps
                // one chain at a time
                float m12, d12;
                log_plus_exact_x2(m2m, d2m, m2d, d2d, cx.exp_tab, m12, d12);
                log_plus_exact_x2(m12, i2m, d12, i2d, cx.exp_tab, M, D);
            } else {
                M = plus2<kFast>(cx, plus2<kFast>(cx, m2m, d2m), i2m);
                D = plus2<kFast>(cx, plus2<kFast>(cx, m2d, d2d), i2d);
            }
            const float I = plus2<kFast>(cx, m2i, i2i);
            dgM = upM;
            dgD = upD;
            dgI = upI;
            lfM = M;
            lfI = I;
            st.M[c] = M;
            st.D[c] = D;
            st.I[c] = I;
            if constexpr(kFast) {
                // (the tolerance build runs at the rate HBM takes its 12 B per cell: the M/D/I stream is written once and read by
                // the sampler much later -- a non-temporal store keeps it from being allocated in the L2 on its way)
                typedef float f3_t __attribute__((ext_vector_type(3)));
                const f3_t v{M, D, I};
                asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(dst + c * (3 * kWave)), "v"(v) : "memory");
            } else {
                *reinterpret_cast<Mdi*>(dst + c * (3 * kWave)) = Mdi{M, D, I};
            }
        }
        arow = arow_next;
        const int r = static_cast<int>(kstep) - lane;  // body row this lane just did
        if(!cx.last_strip && lane == kWave - 1 && r >= 0 && r < static_cast<int>(cx.la)) {
            float* b = cx.bnd_out + 3 * static_cast<uint64_t>(r + 1);
            store_through(&b[0], st.M[W - 1]);
            store_through(&b[1], st.D[W - 1]);
            store_through(&b[2], st.I[W - 1]);
        }
        if(cx.last_strip && r == static_cast<int>(cx.la) - 1 && lane == cx.last_lane) {
            float m = st.M[0], d = st.D[0], in = st.I[0];
#pragma unroll
            for(int c = 1; c < W; ++c) {
                m = (c == cx.last_c) ? st.M[c] : m;
                d = (c == cx.last_c) ? st.D[c] : d;
                in = (c == cx.last_c) ? st.I[c] : in;
            }
            // terminal adjustment (align_pair.cc:130-138)
            float* f = cx.final_mdi + 3 * static_cast<uint64_t>(cx.pair);
            f[0] = (m + k.ng) + k.ng;
            f[1] = d + k.gs;
            f[2] = (in + k.gs) + k.ng;
        }
    }
}

template <int W, bool kFirst, bool kFast>
__device__ __forceinline__ void fwd_steps(const FwdCtx& cx, FwdLane<W>& st, uint32_t& arow, float (&s)[W], const uint32_t (&boff)[W], uint32_t kbase,
                                          uint32_t k0, uint32_t k1, uint32_t a_chunk, float chDM, float chDD, float chDI, float chLM, float chLI,
                                          const char* tab_bytes) {
    // steps kbase + k0 .. kbase + k1 - 1 of a chunk
    if constexpr(!kFirst && kFast) {
        // two steps per iteration: M/D/I of the row above ping-pong between two register sets
        // instead of being copied (see viterbi_l1.hip: run_chunk)
        uint32_t kk = k0;
        for(; kk + 1 < k1; kk += 2) {
            fwd_step<W, kFirst, kFast>(cx, st, arow, s, boff, kbase, kk, a_chunk, chDM, chDD, chDI, chLM, chLI, tab_bytes);
            fwd_step<W, kFirst, kFast>(cx, st, arow, s, boff, kbase, kk + 1, a_chunk, chDM, chDD, chDI, chLM, chLI, tab_bytes);
        }
        if(kk < k1) fwd_step<W, kFirst, kFast>(cx, st, arow, s, boff, kbase, kk, a_chunk, chDM, chDD, chDI, chLM, chLI, tab_bytes);
    } else {
        for(uint32_t kk = k0; kk < k1; ++kk)
            fwd_step<W, kFirst, kFast>(cx, st, arow, s, boff, kbase, kk, a_chunk, chDM, chDD, chDI, chLM, chLI, tab_bytes);
    }
}

__device__ __forceinline__ void publish(uint32_t* word, uint32_t rows, bool leader) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if(leader) __hip_atomic_store(word, rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (no acquire fence -- on this chip an invalidate of the XCD's L2, once per hand-over: round 5 -- the boundary values the
// wait is for were stored THROUGH the producer's L2 (store_through) and are read past this one's: load_through)
__device__ __forceinline__ bool wait_rows(const uint32_t* word, uint32_t need) {
    for(uint32_t spins = 0; __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need; ++spins) {
        if(spins > (1u << 26)) return false;
        __builtin_amdgcn_s_sleep(4);
    }
    return true;
}
__device__ __forceinline__ float load_through(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One strip (64 * W descendant columns) of one pair by one wavefront.
template <int W, bool kFast>
__device__ __forceinline__ void forward_strip(const GapConsts& k, const PairDesc& pd, uint32_t pair, uint32_t strip,
                                              uint32_t ticket, int lane, const uint8_t* __restrict__ a_cat,
                                              const uint8_t* __restrict__ b_cat, float* __restrict__ bnd,
                                              float* __restrict__ mdi, float* __restrict__ final_mdi,
                                              uint32_t* __restrict__ progress, const uint64_t* exp_tab,
                                              const char* tab_bytes) {
    const uint32_t la = pd.la, lb = pd.lb;
    const uint8_t* __restrict__ a = a_cat + pd.a_off;
    const uint8_t* __restrict__ b = b_cat + pd.b_off;
    const uint32_t strips = fwd_strips_w(lb, W);
    const uint32_t col0 = strip * (kWave * W);
    const uint32_t ncol = min(static_cast<uint32_t>((kWave * W)), lb - col0);
    const uint32_t nlanes = (ncol + W - 1) / W;
    const uint32_t nsteps = la + nlanes - 1;
    const bool last_strip = strip + 1 == strips;
    const uint64_t bstride = 3 * (static_cast<uint64_t>(la) + 1);
    float* __restrict__ bnd_out = bnd + pd.bnd_off + strip * bstride;
    const float* __restrict__ bnd_in = bnd + pd.bnd_off + (strip - 1) * bstride;  // strip > 0 only

    uint32_t boff[W];
#pragma unroll
    for(int c = 0; c < W; ++c) {
        const uint32_t bj = col0 + lane * W + c;
        boff[c] = bj < lb ? static_cast<uint32_t>(b[bj]) * 4u : 0u;
    }
    const FwdCtx cx{k, gap_vec(k), la, col0, nsteps, pair, lane,
                    static_cast<int>(((lb - 1) & ((kWave * W) - 1)) / W), static_cast<int>((lb - 1) & (W - 1)),
                    last_strip, mdi + pd.mdi_off + strip * strip_mdi_floats_w(la, W) + 3 * lane, bnd_out, final_mdi, exp_tab};
    FwdLane<W> st;
#pragma unroll
    for(int c = 0; c < W; ++c) st.M[c] = st.D[c] = st.I[c] = kLowest;
    st.oM = st.oD = st.oI = kLowest;
    uint32_t arow = lane == 0 ? static_cast<uint32_t>(a[0]) * (kTabStride * 4u) : 0u;
    float s[W];
#pragma unroll
    for(int c = 0; c < W; ++c) s[c] = *reinterpret_cast<const float*>(tab_bytes + arow + boff[c]);

    // Hand-over between the strips of a pair.  Wide strips: once per 64-step chunk (the consumer waits for the 64 rows its
    // next chunk needs, the producer publishes its progress at the end of a chunk: a strip follows its neighbour at ~130-190
    // steps).  Strips of <= 4 columns per lane -- few pairs: `coati sample` works on one, and a 1 kb pair is 16 such strips in
    // a row -- hand over every kSubRows = 16 (one or two columns per lane: 8) steps instead (the cost is a drain of the M/D/I stores per publish, nothing
    // when a step is ~1 us of arithmetic): the pipeline of 16 strips fills in 16 x ~85 steps instead of 16 x ~160.
    constexpr uint32_t kSubRows = W <= 2 ? kFwdSubRowsNarrow : W <= 4 ? 16u : static_cast<uint32_t>(kWave);
    bool ok = true;
    for(uint32_t kbase = 0; kbase < nsteps; kbase += kWave) {
        const uint32_t crow = kbase + lane;  // the body row lane 0 processes at step kbase + lane
        uint32_t a_chunk = 0;
        float chDM = kLowest, chDD = kLowest, chDI = kLowest, chLM = kLowest, chLI = kLowest;
        if(crow + 1 < la) a_chunk = static_cast<uint32_t>(a[crow + 1]) * (kTabStride * 4u);
        if(strip == 0) {
            // matrix column 0 (align_pair.cc:82-86): diagonal of body row r is matrix cell (r, 0)
            if(crow == 0) chDM = 0.0f;
            else if(crow < la) chDD = (k.ng + k.go) + k.ge * static_cast<float>(crow - 1);
        }
        const uint32_t kend = min(static_cast<uint32_t>(kWave), nsteps - kbase);
        for(uint32_t k0 = 0; k0 < kend; k0 += kSubRows) {
            const uint32_t k1 = min(k0 + kSubRows, kend);
            if(strip > 0) {
                // rows kbase + k0 .. kbase + k1 - 1 of the left neighbour's last column: lanes k0 .. k1 - 1 take theirs
                ok = ok && wait_rows(progress + ticket - 1, min(la, kbase + k1));
                if(crow < la && static_cast<uint32_t>(lane) >= k0 && static_cast<uint32_t>(lane) < k1) {
                    const float* dgp = bnd_in + 3 * static_cast<uint64_t>(crow);      // body row crow - 1
                    const float* lfp = bnd_in + 3 * static_cast<uint64_t>(crow + 1);  // body row crow
                    chDM = load_through(dgp + 0);
                    chDD = load_through(dgp + 1);
                    chDI = load_through(dgp + 2);
                    chLM = load_through(lfp + 0);
                    chLI = load_through(lfp + 2);
                }
            }
            asm volatile("" : "+v"(a_chunk), "+v"(chDM), "+v"(chDD), "+v"(chDI), "+v"(chLM), "+v"(chLI));
            if(kbase == 0)
                fwd_steps<W, true, kFast>(cx, st, arow, s, boff, kbase, k0, k1, a_chunk, chDM, chDD, chDI, chLM, chLI, tab_bytes);
            else
                fwd_steps<W, false, kFast>(cx, st, arow, s, boff, kbase, k0, k1, a_chunk, chDM, chDD, chDI, chLM, chLI, tab_bytes);
            if(!last_strip) {
                const uint32_t done = kbase + k1;  // steps completed: lane 63 has finished body row done - 64
                if(done > kWave - 1) publish(progress + ticket, min(la, done - (kWave - 1)), lane == kWave - 1);
            }
        }
    }
    if(!last_strip) publish(progress + ticket, la, lane == kWave - 1);
    if(!ok && last_strip && lane == 0) final_mdi[3 * static_cast<uint64_t>(pair)] = __builtin_nanf("");
}

// ---------------------------------------------------------------------------------------------
// QUAD strips (round 5): a handful of pairs -- `coati sample` works on ONE, BASELINE configs[3] on 16.
// Every wavefront is then alone on its SIMD and issues one instruction per ~6 cycles whatever its lanes hold: a step of
// the 1-column strip above costs the 417 instructions of a cell, 1.4 us, and a 1 kb pair is la + lb ~ 2 000 of them in a
// row (+ the hand-overs of its 16 strips): 3.1 ms.  Here the three sums of a cell go to three lanes of a QUAD --
//     lane 4c + 0:  M = plus(plus(m2m, d2m), i2m)      from the DIAGONAL cell  (left column, the row before)
//     lane 4c + 1:  D = plus(plus(m2d, d2d), i2d)      from the cell ABOVE     (own column)
//     lane 4c + 2:  I = plus(m2i, i2i)                 from the cell to the LEFT (left column, same row)
// (lane 4c + 3 idles) -- so a step is two `plus` instead of five, and a wavefront holds 16 columns instead of 64: a 1 kb
// pair is 63 wavefronts.  Column c of a strip does body row r = step - c.  Every lane keeps the value it computed last
// (`cur`: row r - 1 of its column before the step) and the one before (`old`: row r - 2); four wave_shr:1 bring both of
// the LEFT column to the lane of the same role, quad-wide DPP broadcasts hand each lane the three values its sum needs.
// The expressions, their order and the adds are align_pair.cc:97-119's: an operand a role does not have is -0.0f, the
// exact additive identity (x + -0.0f == x for every x, both zeros included).
// M/D/I go to the SAME places as from 1-column strips (common.hpp: fwd index with f_wlog2 = 0), the boundary arrays
// and progress words are per quad strip (plan.hip: BatchDeviceView::fwd_quad).
template <bool kFast>
__device__ __forceinline__ float quad_plus(float a, float b, const uint64_t* exp_tab) {
    if constexpr(kFast) return log_plus(a, b);
    return log_plus_exact(a, b, exp_tab);
}
template <int kCtrl>
__device__ __forceinline__ float quad_bcast(float v) {  // quad_perm: every lane of a quad reads lane kCtrl & 3 of its quad
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), kCtrl, 0xf, 0xf, false));
}
// lane l receives lane l - 4's `v`; lanes 0 .. 3 receive f0 .. f3
__device__ __forceinline__ float shift_in4(float v, float f0, float f1, float f2, float f3) {
    return shift_in(shift_in(shift_in(shift_in(v, f3), f2), f1), f0);
}

template <bool kFast>
__device__ __forceinline__ void forward_quad_strip(const GapConsts& k, const PairDesc& pd, uint32_t pair, uint32_t sub,
                                                   uint32_t ticket, int lane, const uint8_t* __restrict__ a_cat,
                                                   const uint8_t* __restrict__ b_cat, float* __restrict__ bnd,
                                                   float* __restrict__ mdi, float* __restrict__ final_mdi,
                                                   uint32_t* __restrict__ progress, const uint64_t* exp_tab,
                                                   const char* tab_bytes) {
    const uint32_t la = pd.la, lb = pd.lb;
    const uint8_t* __restrict__ a = a_cat + pd.a_off;
    const uint8_t* __restrict__ b = b_cat + pd.b_off;
    const uint32_t subs = fwd_quad_strips(lb);
    const uint32_t col0 = sub * kFwdQuadCols;
    const uint32_t ncol = min(kFwdQuadCols, lb - col0);
    const uint32_t nsteps = la + ncol - 1;
    const bool last_sub = sub + 1 == subs;
    const uint64_t bstride = 3 * (static_cast<uint64_t>(la) + 1);
    float* __restrict__ bnd_out = bnd + pd.bnd_off + sub * bstride;
    const float* __restrict__ bnd_in = bnd + pd.bnd_off + (sub - 1) * bstride;  // sub > 0 only

    const int role = lane & 3;
    const uint32_t c = static_cast<uint32_t>(lane) >> 2;  // column of the strip
    const uint32_t bj = col0 + c;
    const bool live = c < ncol && role < 3;  // this lane computes a value that is stored
    const bool is_m = role == 0, is_d = role == 1, is_i = role == 2;
    const uint32_t boff = c < ncol ? static_cast<uint32_t>(b[bj]) * 4u : 0u;
    // the constants of the role's three inputs (align_pair.cc:97-119; -0.0f where the expression has no such add)
    const float kz = -0.0f;
    const float ka1 = is_m ? k.ng : is_d ? k.ng : k.go, ka2 = is_m ? k.ng : is_d ? k.go : kz;
    const float kb1 = is_m ? k.gs : k.ge, kb2 = kz;
    const float kc1 = is_m ? k.gs : is_d ? k.gs : kz, kc2 = is_m ? k.ng : is_d ? k.go : kz;
    // where the lane's value of body row r goes: 1-column strip layout, strip bj / 64, lane t = bj % 64, step r + t
    const uint32_t t64 = bj & (kWave - 1);
    float* const mout = mdi + pd.mdi_off + static_cast<uint64_t>(bj >> 6) * strip_mdi_floats_w(la, 1) + (static_cast<uint64_t>(t64) * kWave + t64) * 3 + role;
    // the margin row (matrix row 0, align_pair.cc:88-90) as this role sees it
    const float margin = is_i ? k.go + k.ge * static_cast<float>(bj) : kLowest;

    // `cur`: the value this lane computed last (row r - 1 of its column before a step); `l1p`: what the step before received
    // from the left column -- that column's row r - 1, i.e. this step's DIAGONAL cell
    float cur = kLowest;
    // table-row byte offset of the row this lane's column processes at the CURRENT step (column 0 starts with body row 0; the
    // others receive theirs from the left as the steps go, like forward_strip's lanes), and that row's score
    uint32_t arow = c == 0 ? static_cast<uint32_t>(a[0]) * (kTabStride * 4u) : 0u;
    float sc = *reinterpret_cast<const float*>(tab_bytes + arow + boff);
    // the margin row of this strip's last column is the first thing its right neighbour needs
    if(!last_sub && c == ncol - 1 && role < 3) store_through(&bnd_out[role], margin);

    constexpr uint32_t kSubRows = 4;  // (rows per hand-over)
    // Never hang: a strip whose input does not arrive gives up, says so in its progress word (zero at every launch, otherwise
    // unused by the quad strips) and stores nothing more; its successors, which then wait themselves, look at that word now
    // and then and do the same; the last one marks the pair.
    constexpr uint32_t kQuadGaveUp = 0xffffffffu;
    auto give_up = [&]() {
        if(lane == 0) {
            __hip_atomic_store(progress + ticket, kQuadGaveUp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if(last_sub) final_mdi[3 * static_cast<uint64_t>(pair)] = __builtin_nanf("");
        }
    };
    // three lanes poll entry `e` of the left boundary (M, D, I of body row e - 1; entry 0 = the margin row) until it is there
    auto wait_entry = [&](uint64_t e) {
        for(uint32_t spins = 0;; ++spins) {
            const bool missing = lane < 3 && __builtin_bit_cast(uint32_t, load_through(bnd_in + 3 * e + lane)) == 0xffffffffu;
            if(__builtin_amdgcn_ballot_w64(missing) == 0ull) return true;
            if(spins > (1u << 24) || ((spins & 255u) == 255u && __hip_atomic_load(progress + ticket - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kQuadGaveUp)) return false;
            __builtin_amdgcn_s_sleep(2);
        }
    };
    // what column 0 finds to its left in the row ABOVE its first: matrix cell (0, col0) -- the left strip's margin entry, or
    // M(0, 0) = 0 (align_pair.cc:82-90); lanes 0 .. 2 hold it by role
    float l1p = kLowest;
    if(sub == 0) {
        if(lane == 0) l1p = 0.0f;
    } else {
        if(!wait_entry(0)) return give_up();
        if(lane < 3) l1p = load_through(bnd_in + lane);
    }
    for(uint32_t kbase = 0; kbase < nsteps; kbase += kWave) {
        const uint32_t crow = kbase + lane;  // the body row column 0 processes at step kbase + lane
        // the cell to the LEFT of column 0 in body row crow: matrix column 0 (align_pair.cc:82-86) or the left strip's entry crow + 1
        float chM = kLowest, chD = kLowest, chI = kLowest;
        uint32_t a_chunk = 0;
        if(crow + 1 < la) a_chunk = static_cast<uint32_t>(a[crow + 1]) * (kTabStride * 4u);
        if(sub == 0 && crow < la) chD = (k.ng + k.go) + k.ge * static_cast<float>(crow);
        const uint32_t kend = min(static_cast<uint32_t>(kWave), nsteps - kbase);
        for(uint32_t k0 = 0; k0 < kend; k0 += kSubRows) {
            const uint32_t k1 = min(k0 + kSubRows, kend);
            if(sub > 0) {
                // The left strip's last column, SELF-VALIDATING (as viterbi_ck's strips hand over): the boundary arrays start the
                // launch as NaN patterns, the producer stores its values through its L2 as it goes -- no drain of its M/D/I
                // stores, no progress word: a publish every 8 steps cost a wavefront of 0.8-us steps a third of its time -- and the
                // consumer loads the rows of its next steps past its own L2 until none is the pattern.  The wait itself is three
                // lanes polling the LAST row the block needs (the producer stores its rows in order), not a dozen loads per round:
                // a consumer on the producer's CU otherwise keeps the CU's memory pipeline busy.
                const bool mine = crow < la && static_cast<uint32_t>(lane) >= k0 && static_cast<uint32_t>(lane) < k1;
                const float* lfp = bnd_in + 3 * static_cast<uint64_t>(crow + 1);  // body row crow
                for(;;) {
                    if(!wait_entry(min(la, kbase + k1))) return give_up();
                    bool valid = true;
                    if(mine) {
                        chM = load_through(lfp + 0);
                        chD = load_through(lfp + 1);
                        chI = load_through(lfp + 2);
                        valid = __builtin_bit_cast(uint32_t, chM) != 0xffffffffu && __builtin_bit_cast(uint32_t, chD) != 0xffffffffu &&
                                __builtin_bit_cast(uint32_t, chI) != 0xffffffffu;
                    }
                    // (the rows before the last one were stored earlier, but nothing orders their arrival: checked, tried again)
                    if(__builtin_amdgcn_ballot_w64(valid) == __builtin_amdgcn_ballot_w64(true)) break;
                }
            }
            asm volatile("" : "+v"(a_chunk), "+v"(chM), "+v"(chD), "+v"(chI));
            for(uint32_t kk = k0; kk < k1; ++kk) {
                const uint32_t kstep = kbase + kk;
                const int32_t r = static_cast<int32_t>(kstep) - static_cast<int32_t>(c);  // body row of this lane's column
                cur = kstep == c ? margin : cur;  // the column starts: the row above is the margin row
                // the next step's row and its score (gathered now, consumed a step later)
                const uint32_t a_next = read_lane(a_chunk, static_cast<int>(kk));
                const uint32_t arow_next = shift_in(shift_in(shift_in(shift_in(arow, a_next), a_next), a_next), a_next);
                const float sc_next = *reinterpret_cast<const float*>(tab_bytes + arow_next + boff);
                // ---- the left column's row r, to the lane of the same role (column 0: the strip's left boundary)
                const float l1 = shift_in4(cur, read_lane(chM, static_cast<int>(kk)), read_lane(chD, static_cast<int>(kk)), read_lane(chI, static_cast<int>(kk)), kLowest);
                // ---- the three inputs of this lane's sum
                // (every broadcast by ALL lanes, then the choice: a DPP read under a role's EXEC mask finds its source lane off)
                const float dgM = quad_bcast<0x00>(l1p), dgD = quad_bcast<0x55>(l1p), dgI = quad_bcast<0xaa>(l1p);
                const float upM = quad_bcast<0x00>(cur), upD = quad_bcast<0x55>(cur), upI = quad_bcast<0xaa>(cur);
                const float lfM = quad_bcast<0x00>(l1), lfI = quad_bcast<0xaa>(l1);
                const float a1 = is_m ? dgM : is_d ? upM : lfM;
                const float a2 = is_m ? dgD : is_d ? upD : lfI;
                const float a3 = is_m ? dgI : is_d ? upI : kLowest;
                const float scm = is_m ? sc : kz;
                const float t1 = ((a1 + ka1) + ka2) + scm;
                const float t2 = ((a2 + kb1) + kb2) + scm;
                const float t3 = ((a3 + kc1) + kc2) + scm;
                const float r12 = quad_plus<kFast>(t1, t2, exp_tab);
                const float r123 = quad_plus<kFast>(r12, t3, exp_tab);
                const float val = is_i ? r12 : r123;
                l1p = l1;
                cur = val;
                sc = sc_next;
                arow = arow_next;
                if(live && r >= 0 && r < static_cast<int32_t>(la)) {
                    mout[static_cast<uint64_t>(r) * (3 * kWave)] = val;
                    if(!last_sub && c == ncol - 1) store_through(&bnd_out[3 * static_cast<uint64_t>(r + 1) + role], val);
                }
            }
        }
    }
    // the last column's last step was the strip's last: `cur` is its value of body row la - 1.  Terminal adjustment
    // (align_pair.cc:130-138): (m + ng) + ng, d + gs, (in + gs) + ng
    if(last_sub && live && c == ncol - 1) {
        const float f1 = is_m ? k.ng : k.gs, f2 = is_d ? kz : k.ng;
        final_mdi[3 * static_cast<uint64_t>(pair) + role] = (cur + f1) + f2;
    }
}

// kNarrow: strips of at most 8 columns per lane only -- half the lane state, so the build fits 128 VGPRs and four
// wavefronts share a SIMD (the latency of the fp64 / reciprocal chains of the exact `plus` is what the wide build
// cannot hide with three).
template <bool kFast, bool kNarrow, bool kQuad = false>
__device__ __forceinline__ void forward_l1_body(
    const float* __restrict__ table, GapConsts k, const PairDesc* __restrict__ pairs,
    const WorkItem* __restrict__ items, uint32_t n_items, uint32_t* __restrict__ queue,
    uint32_t* __restrict__ progress, const uint8_t* __restrict__ a_cat, const uint8_t* __restrict__ b_cat,
    float* __restrict__ bnd, float* __restrict__ mdi, float* __restrict__ final_mdi, float (*tab_all)[kTabRows * kTabStride],
    uint64_t* exp_tab) {
    load_exp_table(exp_tab, threadIdx.x);
    const int lane_id = threadIdx.x & (kWave - 1);
    float* tab = tab_all[kNarrow ? 0 : threadIdx.x / kWave];
    uint32_t tab_held = 0xffffffffu;
    if constexpr(kNarrow) {  // (the narrow build serves models with ONE table: a copy per workgroup, not per wavefront)
        for(int idx = threadIdx.x; idx < kTabFloats; idx += kFillWaves * kWave) {
            const int r = idx / kTabCols, c = idx - r * kTabCols;
            tab[r * kTabStride + c] = table[idx];
        }
        tab_held = 0u;
    }
    __syncthreads();
    const char* tab_bytes = reinterpret_cast<const char*>(tab);
    for(;;) {
        int lane = lane_id;  // opaque per iteration (see viterbi_l1.hip)
        asm volatile("" : "+v"(lane));
        uint32_t ticket = atomicAdd(queue, lane == 0 ? 1u : 0u);
        ticket = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ticket)));
        if(ticket >= n_items) break;
        const WorkItem item = items[ticket];
        const uint32_t pair = item.pair, strip = item.strip;
        const PairDesc pd = pairs[pair];
        const uint32_t la = pd.la, lb = pd.lb;
        if(!kNarrow && pd.table != tab_held) {
            const float* __restrict__ src = table + static_cast<size_t>(pd.table) * kTabFloats;
            for(int idx = lane; idx < kTabFloats; idx += kWave) {
                const int r = idx / kTabCols, c = idx - r * kTabCols;
                tab[r * kTabStride + c] = src[idx];
            }
            tab_held = pd.table;
        }
        if(la == 0 || lb == 0) {  // no body cells: the last cell is a margin cell
            float m, d, in;
            margin_mdi(k, 1u, la, lb, m, d, in);
            if(lane == 0) {
                final_mdi[3 * static_cast<uint64_t>(pair) + 0] = (m + k.ng) + k.ng;
                final_mdi[3 * static_cast<uint64_t>(pair) + 1] = d + k.gs;
                final_mdi[3 * static_cast<uint64_t>(pair) + 2] = (in + k.gs) + k.ng;
            }
            continue;
        }
        if constexpr(kQuad) {
            forward_quad_strip<kFast>(k, pd, pair, strip, ticket, lane, a_cat, b_cat, bnd, mdi, final_mdi, progress, exp_tab, tab_bytes);
            continue;
        }
        // strips are 64 * W columns, W = 16 unless the batch has too few pairs to fill the GPU (abi.hip)
        switch(pd.f_wlog2) {
            case 0: forward_strip<1, kFast>(k, pd, pair, strip, ticket, lane, a_cat, b_cat, bnd, mdi, final_mdi, progress, exp_tab, tab_bytes); break;
            case 1: forward_strip<2, kFast>(k, pd, pair, strip, ticket, lane, a_cat, b_cat, bnd, mdi, final_mdi, progress, exp_tab, tab_bytes); break;
            case 2: forward_strip<4, kFast>(k, pd, pair, strip, ticket, lane, a_cat, b_cat, bnd, mdi, final_mdi, progress, exp_tab, tab_bytes); break;
            case 3: forward_strip<8, kFast>(k, pd, pair, strip, ticket, lane, a_cat, b_cat, bnd, mdi, final_mdi, progress, exp_tab, tab_bytes); break;
            default:
                if constexpr(!kNarrow) forward_strip<16, kFast>(k, pd, pair, strip, ticket, lane, a_cat, b_cat, bnd, mdi, final_mdi, progress, exp_tab, tab_bytes);
                break;
        }
    }
}

// The four builds.  The narrow ones are held to 128 VGPRs (amdgpu_num_vgpr: __launch_bounds__' second argument is only
// a request -- left alone the compiler took 135-138 registers and with them the fourth wavefront per SIMD).
#define COATI_FWD_KERNEL(NAME, FAST, NARROW, QUAD, ATTR)                                                                  \
    __global__ ATTR void NAME(const float* __restrict__ table, GapConsts k, const PairDesc* __restrict__ pairs,          \
                              const WorkItem* __restrict__ items, uint32_t n_items, uint32_t* __restrict__ queue,        \
                              uint32_t* __restrict__ progress, const uint8_t* __restrict__ a_cat,                        \
                              const uint8_t* __restrict__ b_cat, float* __restrict__ bnd, float* __restrict__ mdi,       \
                              float* __restrict__ final_mdi) {                                                           \
        __shared__ float tab_all[NARROW ? 1 : kFillWaves][kTabRows * kTabStride]; /* wide: one table per wavefront */     \
        __shared__ uint64_t exp_tab[32];                                                                                 \
        forward_l1_body<FAST, NARROW, QUAD>(table, k, pairs, items, n_items, queue, progress, a_cat, b_cat, bnd, mdi, final_mdi, \
                                      tab_all, exp_tab);                                                                 \
    }
COATI_FWD_KERNEL(forward_l1_exact_wide, false, false, false, __launch_bounds__(kFillWaves* kWave, 3))
COATI_FWD_KERNEL(forward_l1_fast_wide, true, false, false, __launch_bounds__(kFillWaves* kWave, 2))
COATI_FWD_KERNEL(forward_l1_exact_narrow, false, true, false, __launch_bounds__(kFillWaves* kWave, 4) __attribute__((amdgpu_num_vgpr(128))))
COATI_FWD_KERNEL(forward_l1_fast_narrow, true, true, false, __launch_bounds__(kFillWaves* kWave, 4) __attribute__((amdgpu_num_vgpr(128))))
// (quad strips: one table per workgroup like the narrow builds; a wavefront per SIMD is all there is to place)
COATI_FWD_KERNEL(forward_l1_exact_quad, false, true, true, __launch_bounds__(kFillWaves* kWave))
COATI_FWD_KERNEL(forward_l1_fast_quad, true, true, true, __launch_bounds__(kFillWaves* kWave))
#undef COATI_FWD_KERNEL

}  // namespace

hipError_t launch_forward_l1(const BatchDeviceView& v, bool one_table, hipStream_t stream) {
    hipError_t e = zero_queue_and_progress(v, v.n_fwd_items, stream);  // ticket counter + polled words: zero every launch
    if(e != hipSuccess) return e;
    // workgroups of 4 wavefronts (one per SIMD); per CU: 2 (fast, 16 columns per lane), 3 (exact, 16 columns), 4 (at most
    // 8 columns per lane: the narrow build); fewer when there are fewer items than wavefronts
    const bool fast = v.fwd_fast != 0, narrow = one_table && v.fwd_wlog2_max <= 3 && !env_options().fwd_wide_build;
    const uint32_t per_cu = narrow ? 4u : (fast ? 2u : 3u);
    const uint32_t cus = device_cu_count();
    const uint32_t blocks = std::min<uint32_t>(cus * per_cu, std::max<uint32_t>(cus, (v.n_fwd_items + kFillWaves - 1) / kFillWaves));
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kFillWaves * kWave), 0, stream, v.table, v.k, v.pairs, v.fwd_items, v.n_fwd_items,
                           v.queue, v.progress, v.a_cat, v.b_cat, v.bnd, v.mdi, v.final_mdi);
    };
    if(v.fwd_quad != 0 && one_table) {
        // (the quad strips' boundary values validate themselves: every launch starts from NaN patterns)
        if(v.bnd_bytes != 0) {
            e = hipMemsetAsync(v.bnd, 0xff, v.bnd_bytes, stream);
            if(e != hipSuccess) return e;
        }
        if(fast) go(forward_l1_fast_quad);
        else go(forward_l1_exact_quad);
        return hipGetLastError();
    }
    if(fast && narrow) go(forward_l1_fast_narrow);
    else if(fast) go(forward_l1_fast_wide);
    else if(narrow) go(forward_l1_exact_narrow);
    else go(forward_l1_exact_wide);
    return hipGetLastError();
}

}  // namespace coati_hip_detail
