// viterbi_lp: the gap_len-1 Viterbi fill + traceback for a FEW LONG pairs (BASELINE configs[2]: one 160 kb pair) -- the batches
// whose strip plan has fewer strips than the GPU has SIMDs, so that every strip is 4, 3 or 2 columns per lane wide and every
// wavefront is alone on its SIMD.  Same recurrence and the same five decision bits per cell as viterbi_l1.hip, the same strip
// pipeline through self-validating boundary values; what differs is how a step is issued and how the path is walked.
//
// What it replaces in the reference: forward_impl<tropical, align_pair_work_mem_t> (src/lib/align_pair.cc:62-139)
// and traceback<tropical> (align_pair.cc:249-303) for one pair that the CPU tool cannot hold in memory.
//
// The fill.  A lone wavefront issues one instruction per ~4.5 cycles WHATEVER the instruction (an LDS or vector memory
// instruction: ~18), so its time is its instruction count.  viterbi_l1's 4-column step is ~175 instructions; here 81
// (4 columns), 63.5 (3) and 46 (2):
//   * the cell's eleven additions are six (v_pk_add_f32 does two fp32 additions -- IEEE, the same bits -- in one
//     instruction) and the five sign differences of the decisions three: 17 instructions per cell (gen_viterbi_lp.py);
//   * 16 steps are ONE block of hand-allocated instruction text: lane 0 takes the strip's left boundary straight
//     from lane j of the chunk registers by a row_shl:j DPP, stores use immediate offsets, the last column ping-pongs its
//     state between two register pairs so that the diagonal hand-off needs no copy;
//   * the left boundary arrives in 16-row chunks, (X of row r - 1, Z of row r) as one 8-byte load per row, requested late
//     in the block before and checked after it; the right boundary leaves as one 8-byte store per step (round 6);
//   * 3 columns per lane (round 6: the 160 kb pair is 834 strips on 1 024 SIMDs where 4 columns leave 398 SIMDs without
//     one) keep their decision words per column, lane-major, three stores per block (common.hpp).
// The traceback (round 6): lp_walk_iter -- the walk away from the margins, one asm statement per iteration --, and the splice:
// every strip's wavefront walks its strip speculatively as soon as its fill is done (lp_spec_walk: a record of run starts, and
// a bridge from its right neighbour's exit to that record), the true walk copies what it meets (lp_list_enter,
// lp_splice_apply).  160 002 x 160 002: 33.4 ms, of which the walk 1 ms (round 5: 43.0 / 5.9).
// fp32 only, adds/max/compares in the reference's evaluation order; built with -ffp-contract=off.
#include "viterbi_cell.hpp"

#include "viterbi_lp_block.inc"

#include <algorithm>
#include <cstdlib>
#include <utility>

namespace coati_hip_detail {
#ifdef COATI_FILL_TRACE
// Debug build only (make trace): per wavefront {start, fill of its strip done, traceback done} in 100 MHz ticks and the
// strip it took, read back by tools/trace_long.py through coati_hip_debug_trace_lp.
__device__ unsigned long long g_lp_trace[4096 * 4];
#endif
namespace {

constexpr uint32_t kLpRows = 16;
// Pair table: the scores of two adjacent descendant columns (A/C/G/T both) as one 8-byte entry, 16 entries per table row;
// row stride 136 bytes = twice the single table's 68, so that the row offset the lanes hand along is simply doubled
constexpr uint32_t kLpPairStride = 2u * kTabStride * 4u;  // steps per block = rows per boundary chunk
typedef u32x4_t u32x4;
constexpr uint32_t kLpDrop = kDropOffset;  // offset register of a lane that does not store (out of every range)
__device__ __forceinline__ u32x4 lp_rsrc(const void* p, uint64_t bytes) { return raw_rsrc(p, bytes); }

// what one strip's wavefront keeps besides the LaneState: named operands of the blocks (W = columns per lane: 4 or 2)
template <int W>
struct LpStrip {
    GapVec kv;
    uint32_t bl[W];   // LDS byte address of the lane's column c in table row 0
    uint32_t blp[W / 2];  // ... of the lane's column pair (2h, 2h+1) in row 0 of the pair table
    bool pairtab;     // (wave-uniform) the strip's descendant columns are all A/C/G/T and the launch has pair tables
    uint32_t offx, offb;  // per-lane offsets: the boundary store (lane 63: 0, else kLpDrop), decision rows (3 columns: lane * 12)
    uint32_t offb2;       // (3 columns) the lane's offset in the group's C block: lane * 8
    u32x4 rs_in, rs_out, rs_bits, rs_a;
    float mx[W], my[W];  // the lane's margin-row state (taken at step == lane)
};

// 3 columns per lane: the decision bits of each column in accumulators of its own (common.hpp: the per-column layout)
struct LpColAcc {
    uint32_t a[3], b[3], c[3];
};

// 16 wavefront steps from `kbase`.  In: the chunk (bx, bz, ach: lane j < 16 holds the strip's left boundary of row
// kbase + j and the table row offset of ancestor row kbase + j + 1).  Out: the raw next chunk (rows kbase + 16 ...).
template <int W, bool kFirst>
__device__ __forceinline__ void lp_block(const LpStrip<W>& sp, LaneState<W>& st, uint32_t& arow, float (&s)[W], uint32_t kbase,
                                         int lane, uint32_t la, float bx, float bz, uint32_t ach, uint64_t& nxz, uint32_t& na,
                                         LpColAcc& ca) {
    static_assert(W == 4 || W == 3 || W == 2, "gen_viterbi_lp.py writes these three shapes");
    uint32_t arb;
    const uint32_t next = kbase + kLpRows + static_cast<uint32_t>(lane);
    // (the interleaved boundary array, below: row r's pair (X of row r - 1, Z of row r) is floats 2r, 2r + 1)
    const uint32_t vin_x = next * 8u, vin_a = next + 1u;
    const uint32_t so_bits = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(
        static_cast<int>(W == 3 ? (kbase / kLpRows) * (kLp3GroupDwords * 4u) : (kbase / (32u / (W == 3 ? 4 : W))) * (kPairDwords * 4u))));
    const uint32_t lrel = static_cast<uint32_t>(lane) - kbase;  // (first blocks) the lane starts at step kbase + lrel
    // (main blocks) lane 63 did body row kbase + j - 63 at step j: [Z : X] of its last column go to floats 1 + 2 row, 2 + 2 row
    const uint32_t so_out = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>((kbase - (kWave - 1u)) * 8u)));
    const uint32_t sel_lo16 = 0x05040100u;  // v_perm_b32: the low halves of two registers side by side
    // [X:Y] of column c in v[8+2c : 9+2c]; the last column's pair alternates with the pair after it, whose low half is
    // xlast_old between blocks; the scores in v44.. (the other set, v48.., is scratch) -- gen_viterbi_lp.py
#define COATI_LP_STATE_OUT                                                                                                    \
    "+{v28}"(st.zlast), "+{v44}"(s[0]), "+{v45}"(s[1]), [ara] "+v"(arow), [arb] "=&v"(arb), [nxz] "=&v"(nxz), [na] "=&v"(na)
#define COATI_LP_COMMON_OUT COATI_LP_STATE_OUT, [aa] "+v"(st.acc[ACC_A]), [ab] "+v"(st.acc[ACC_B]), [ac] "+v"(st.acc[ACC_C])
#define COATI_LP_COMMON_IN                                                                                                    \
    "{v2}"(sp.kv.go), "{v3}"(sp.kv.ng), "{v4}"(sp.kv.ge), "{v5}"(sp.kv.gs), [bx] "v"(bx), [bz] "v"(bz), [ach] "v"(ach),       \
        [bl0] "v"(sp.bl[0]), [bl1] "v"(sp.bl[1]), [blp0] "v"(sp.blp[0]), [offb] "v"(sp.offb), [vin_x] "v"(vin_x),             \
        [vin_a] "v"(vin_a), [rs_in] "s"(sp.rs_in), [rs_bits] "s"(sp.rs_bits), [rs_a] "s"(sp.rs_a), [so_bits] "s"(so_bits)
#define COATI_LP_FIRST_IN [lrel] "v"(lrel), [mx0] "v"(sp.mx[0]), [mx1] "v"(sp.mx[1]), [my0] "v"(sp.my[0]), [my1] "v"(sp.my[1])
#define COATI_LP_MAIN_IN [offx] "v"(sp.offx), [rs_out] "s"(sp.rs_out), [so_out] "s"(so_out)
    if constexpr(W == 4) {
#define COATI_LP4_OUT                                                                                                         \
    "+{v8}"(st.X[0]), "+{v9}"(st.Y[0]), "+{v10}"(st.X[1]), "+{v11}"(st.Y[1]), "+{v12}"(st.X[2]), "+{v13}"(st.Y[2]),          \
        "+{v14}"(st.X[3]), "+{v15}"(st.Y[3]), "+{v16}"(st.xlast_old), "+{v46}"(s[2]), "+{v47}"(s[3]), COATI_LP_COMMON_OUT
#define COATI_LP4_IN COATI_LP_COMMON_IN, [bl2] "v"(sp.bl[2]), [bl3] "v"(sp.bl[3]), [blp1] "v"(sp.blp[1])
#define COATI_LP4_CLOBBERS COATI_LP_SCRATCH_CLOBBERS, "v17", "v48", "v49", "v50", "v51", "memory"
#define COATI_LP4_FIRST_IN COATI_LP4_IN, COATI_LP_FIRST_IN, [mx2] "v"(sp.mx[2]), [mx3] "v"(sp.mx[3]), [my2] "v"(sp.my[2]), [my3] "v"(sp.my[3])
        if(sp.pairtab) {  // (wave-uniform)
            if constexpr(kFirst)
                asm volatile(COATI_LP4P_BLOCK_FIRST_ASM : COATI_LP4_OUT : COATI_LP4_FIRST_IN : COATI_LP4_CLOBBERS, "vcc");
            else
                asm volatile(COATI_LP4P_BLOCK_MAIN_ASM : COATI_LP4_OUT : COATI_LP4_IN, COATI_LP_MAIN_IN : COATI_LP4_CLOBBERS);
        } else {
            if constexpr(kFirst)
                asm volatile(COATI_LP4_BLOCK_FIRST_ASM : COATI_LP4_OUT : COATI_LP4_FIRST_IN : COATI_LP4_CLOBBERS, "vcc");
            else
                asm volatile(COATI_LP4_BLOCK_MAIN_ASM : COATI_LP4_OUT : COATI_LP4_IN, COATI_LP_MAIN_IN : COATI_LP4_CLOBBERS);
        }
#undef COATI_LP4_OUT
#undef COATI_LP4_IN
#undef COATI_LP4_CLOBBERS
#undef COATI_LP4_FIRST_IN
    } else if constexpr(W == 3) {
        // [X:Y] of columns 0, 1 in v[8:9], v[10:11]; column 2 alternates between v[12:13] and v[14:15] (v14 = xlast_old between blocks);
        // the accumulators where the block's three decision stores take them from (gen_viterbi_lp.py: ACC3_*)
#define COATI_LP3_OUT                                                                                                         \
    "+{v8}"(st.X[0]), "+{v9}"(st.Y[0]), "+{v10}"(st.X[1]), "+{v11}"(st.Y[1]), "+{v12}"(st.X[2]), "+{v13}"(st.Y[2]),          \
        "+{v14}"(st.xlast_old), "+{v46}"(s[2]), COATI_LP_STATE_OUT, "+{v52}"(ca.a[0]), "+{v53}"(ca.a[1]), "+{v54}"(ca.a[2]),     \
        "+{v56}"(ca.b[0]), "+{v57}"(ca.b[1]), "+{v58}"(ca.b[2]), "+{v62}"(ca.c[0]), "+{v63}"(ca.c[1]), "+{v61}"(ca.c[2])
#define COATI_LP3_IN COATI_LP_COMMON_IN, [bl2] "v"(sp.bl[2]), [offb2] "v"(sp.offb2), [sel_lo16] "s"(sel_lo16)
#define COATI_LP3_CLOBBERS COATI_LP_SCRATCH_CLOBBERS, "v15", "v48", "v49", "v50", "v60", "memory"
#define COATI_LP3_FIRST_IN COATI_LP3_IN, COATI_LP_FIRST_IN, [mx2] "v"(sp.mx[2]), [my2] "v"(sp.my[2])
        if(sp.pairtab) {  // (wave-uniform)
            if constexpr(kFirst)
                asm volatile(COATI_LP3P_BLOCK_FIRST_ASM : COATI_LP3_OUT : COATI_LP3_FIRST_IN : COATI_LP3_CLOBBERS, "vcc");
            else
                asm volatile(COATI_LP3P_BLOCK_MAIN_ASM : COATI_LP3_OUT : COATI_LP3_IN, COATI_LP_MAIN_IN : COATI_LP3_CLOBBERS);
        } else {
            if constexpr(kFirst)
                asm volatile(COATI_LP3_BLOCK_FIRST_ASM : COATI_LP3_OUT : COATI_LP3_FIRST_IN : COATI_LP3_CLOBBERS, "vcc");
            else
                asm volatile(COATI_LP3_BLOCK_MAIN_ASM : COATI_LP3_OUT : COATI_LP3_IN, COATI_LP_MAIN_IN : COATI_LP3_CLOBBERS);
        }
#undef COATI_LP3_OUT
#undef COATI_LP3_IN
#undef COATI_LP3_CLOBBERS
#undef COATI_LP3_FIRST_IN
    } else {
#define COATI_LP2_OUT "+{v8}"(st.X[0]), "+{v9}"(st.Y[0]), "+{v10}"(st.X[1]), "+{v11}"(st.Y[1]), "+{v12}"(st.xlast_old), COATI_LP_COMMON_OUT
#define COATI_LP2_CLOBBERS COATI_LP_SCRATCH_CLOBBERS, "v13", "v48", "v49", "memory"
        if(sp.pairtab) {
            if constexpr(kFirst)
                asm volatile(COATI_LP2P_BLOCK_FIRST_ASM : COATI_LP2_OUT : COATI_LP_COMMON_IN, COATI_LP_FIRST_IN : COATI_LP2_CLOBBERS, "vcc");
            else
                asm volatile(COATI_LP2P_BLOCK_MAIN_ASM : COATI_LP2_OUT : COATI_LP_COMMON_IN, COATI_LP_MAIN_IN : COATI_LP2_CLOBBERS);
        } else {
            if constexpr(kFirst)
                asm volatile(COATI_LP2_BLOCK_FIRST_ASM : COATI_LP2_OUT : COATI_LP_COMMON_IN, COATI_LP_FIRST_IN : COATI_LP2_CLOBBERS, "vcc");
            else
                asm volatile(COATI_LP2_BLOCK_MAIN_ASM : COATI_LP2_OUT : COATI_LP_COMMON_IN, COATI_LP_MAIN_IN : COATI_LP2_CLOBBERS);
        }
#undef COATI_LP2_OUT
#undef COATI_LP2_CLOBBERS
    }
#undef COATI_LP_STATE_OUT
#undef COATI_LP_COMMON_OUT
#undef COATI_LP_COMMON_IN
#undef COATI_LP_FIRST_IN
#undef COATI_LP_MAIN_IN
}

// One step in plain C++ (viterbi_cell.hpp's cell): the strip's last nsteps % 16 steps, and every step of a strip
// shorter than a block.  kFirst: the lane takes its margin-row state when the step is its first.
template <int W, bool kFirst>
__device__ __forceinline__ void lp_tail_step(const LpStrip<W>& sp, LaneState<W>& st, uint32_t& arow, float (&s)[W],
                                             const uint32_t (&boff)[W], uint32_t lds_tab, uint32_t kstep, uint32_t kk, int lane,
                                             uint32_t la, bool last_strip, uint32_t* fout, float* bnd_x, uint32_t ach,
                                             float bx, float bz) {
    constexpr uint32_t kMA = 16 / W, kMC = 32 / W;
    if constexpr(kFirst) {
        if(kstep == static_cast<uint32_t>(lane)) {
#pragma unroll
            for(int c = 0; c < W; ++c) {
                st.X[c] = sp.mx[c];
                st.Y[c] = sp.my[c];
            }
        }
    }
    const float diag = shift_in(st.xlast_old, read_lane(bx, static_cast<int>(kk)));
    const float zl = shift_in(st.zlast, read_lane(bz, static_cast<int>(kk)));
    const uint32_t arow_next = shift_in(arow, read_lane(ach, static_cast<int>(kk)));
    row_l1<W>(sp.kv, st, diag, zl, s, lds_tab + arow_next, boff, std::make_integer_sequence<int, W>{});
    arow = arow_next;
    if((kstep & (kMA - 1u)) == kMA - 1u) {
        const uint32_t q = kstep & (kMC - 1u);
        uint32_t* dst = fout + static_cast<uint64_t>(kstep / kMC) * kPairDwords + (q / kMA) * (2 * kWave);
        dst[0] = st.acc[ACC_A];
        dst[kWave] = st.acc[ACC_B];
        if(q == kMC - 1u) fout[static_cast<uint64_t>(kstep / kMC) * kPairDwords + 4 * kWave] = st.acc[ACC_C];
    }
    const int r = static_cast<int>(kstep) - lane;
    if(!last_strip && lane == kWave - 1 && r >= 0 && r < static_cast<int>(la)) {
        store_through(&bnd_x[2 + 2 * r], st.X[W - 1]);
        store_through(&bnd_x[1 + 2 * r], st.zlast);
    }
}

// The same for 3 columns per lane, in plain C++ with the per-column accumulators: the block's operations in the block's order
// (gen_viterbi_lp.py: step) -- every sum a single fp32 addition, the maxima left to right, the five tests as the sign bits of
// z2 - z1, x1 - X, x2 - X, y1 - Y, y2 - Y.
__device__ __forceinline__ uint32_t lp_sign(float v) { return __builtin_bit_cast(uint32_t, v) >> 31; }
// one group of the per-column layout (common.hpp), the accumulators shifted left by `missing` steps (an incomplete last group)
__device__ __forceinline__ void lp3_store_group(uint32_t* grp, int lane, const LpColAcc& ca, uint32_t missing) {
#pragma unroll
    for(int c = 0; c < 3; ++c) {
        grp[3 * lane + c] = ca.a[c] << (2u * missing);
        grp[3 * kWave + 3 * lane + c] = ca.b[c] << (2u * missing);
        reinterpret_cast<uint16_t*>(grp + 6 * kWave)[4 * lane + c] = static_cast<uint16_t>(ca.c[c] << missing);
    }
}
template <bool kFirst>
__device__ __forceinline__ void lp3_tail_step(const LpStrip<3>& sp, LaneState<3>& st, LpColAcc& ca, uint32_t& arow, float (&s)[3],
                                              const uint32_t (&boff)[3], uint32_t lds_tab, uint32_t kstep, uint32_t kk, int lane,
                                              uint32_t la, bool last_strip, uint32_t* fout_strip, float* bnd_x, uint32_t ach, float bx,
                                              float bz) {
    if constexpr(kFirst) {
        if(kstep == static_cast<uint32_t>(lane)) {
#pragma unroll
            for(int c = 0; c < 3; ++c) {
                st.X[c] = sp.mx[c];
                st.Y[c] = sp.my[c];
            }
        }
    }
    float diag = shift_in(st.xlast_old, read_lane(bx, static_cast<int>(kk)));
    float zl = shift_in(st.zlast, read_lane(bz, static_cast<int>(kk)));
    const uint32_t arow_next = shift_in(arow, read_lane(ach, static_cast<int>(kk)));
    st.xlast_old = st.X[2];
#pragma unroll
    for(int c = 0; c < 3; ++c) {
        const float m = diag + s[c];
        const float z1 = m + sp.kv.go, m1 = m + sp.kv.ng;
        const float z2 = zl + sp.kv.ge, i1 = zl + sp.kv.gs;
        const float x1 = m1 + sp.kv.ng, y1 = m1 + sp.kv.go;
        const float x2 = st.Y[c] + sp.kv.gs, y2 = st.Y[c] + sp.kv.ge;
        const float x3 = i1 + sp.kv.ng, y3 = i1 + sp.kv.go;
        const float x = __builtin_fmaxf(__builtin_fmaxf(x1, x2), x3), y = __builtin_fmaxf(__builtin_fmaxf(y1, y2), y3);
        ca.c[c] = (ca.c[c] << 1) | lp_sign(z2 - z1);
        ca.a[c] = (ca.a[c] << 2) | (lp_sign(x1 - x) << 1) | lp_sign(x2 - x);
        ca.b[c] = (ca.b[c] << 2) | (lp_sign(y1 - y) << 1) | lp_sign(y2 - y);
        zl = __builtin_fmaxf(z1, z2);
        diag = st.X[c];
        st.X[c] = x;
        st.Y[c] = y;
        s[c] = *reinterpret_cast<const __attribute__((address_space(3))) float*>(lds_tab + arow_next + boff[c]);
    }
    st.zlast = zl;
    arow = arow_next;
    if((kstep & 15u) == 15u) lp3_store_group(fout_strip + static_cast<uint64_t>(kstep >> 4) * kLp3GroupDwords, lane, ca, 0u);
    const int r = static_cast<int>(kstep) - lane;
    if(!last_strip && lane == kWave - 1 && r >= 0 && r < static_cast<int>(la)) {
        store_through(&bnd_x[2 + 2 * r], st.X[2]);
        store_through(&bnd_x[1 + 2 * r], st.zlast);
    }
}

// One work item: one strip (64 W descendant columns) of one pair, all its rows.  Returns false if the left neighbour's
// boundary column did not arrive within the spin bound.
template <int W>
__device__ __forceinline__ bool fill_strip_lp(const GapConsts& k, const PairDesc& pd, uint32_t pair, uint32_t strip, uint32_t ticket,
                                              int lane, uint32_t lds_tab, uint32_t lds_pair, const char* tab_bytes, const uint8_t* __restrict__ a,
                                              const uint8_t* __restrict__ b, uint32_t* __restrict__ flags, float* __restrict__ bnd,
                                              float* __restrict__ scores, uint32_t* __restrict__ progress) {
    const uint32_t la = pd.la, lb = pd.lb;
    const uint32_t col0 = strip * (kWave * W);
    const uint32_t ncol = min(static_cast<uint32_t>(kWave * W), lb - col0);
    const uint32_t nlanes = (ncol + W - 1) / W;
    const uint32_t nsteps = la + nlanes - 1;
    const bool last_strip = strip + 1 == pd.v_strips;
    uint32_t* __restrict__ fout_strip = flags + pd.flags_off + strip * strip_dwords(la, W);
    uint32_t* __restrict__ fout = fout_strip + lane;
    // strip-boundary columns, one array of 2 (la + 1) floats per boundary (the plan's size, viterbi_l1's too), INTERLEAVED here
    // (round 6): float 0 = X of the margin row, floats 1 + 2r, 2 + 2r = Z, X of the strip's last column in row r -- so that a
    // step's values leave with one 8-byte store and the pair the next strip needs for ITS row r (X of row r - 1: the diagonal
    // input; Z of row r) are the aligned floats 2r, 2r + 1: one 8-byte load per row of a chunk
    const uint64_t bstride = 2 * (static_cast<uint64_t>(la) + 1);
    float* __restrict__ bnd_x = bnd + pd.bnd_off + strip * bstride;  // written by this strip
    const float* __restrict__ in_x = bnd + pd.bnd_off + (strip - 1) * bstride;  // read by it (strip > 0)
    bool handoff_ok = true;

    uint32_t boff[W];
    LpStrip<W> sp;
    sp.kv = gap_vec(k);
    uint32_t worst = 0;
#pragma unroll
    for(int c = 0; c < W; ++c) {
        const uint32_t bj = col0 + lane * W + c;
        boff[c] = bj < lb ? static_cast<uint32_t>(b[bj]) * 4u : 0u;
        worst = max(worst, boff[c]);
        sp.bl[c] = boff[c] + lds_tab;
        // margin row (matrix row 0, align_pair.cc:88-90): M = D = lowest, I = go + ge*float(j-1)
        const float im = k.go + k.ge * static_cast<float>(bj);
        const float i1 = im + k.gs;
        sp.mx[c] = i1 + k.ng;
        sp.my[c] = i1 + k.go;
    }
    // pair table (kLpPairStride bytes per row, entry (b0, b1) at (b0 * 4 + b1) * 8): only if every column of the strip is A/C/G/T
    sp.pairtab = lds_pair != 0 && __builtin_amdgcn_ballot_w64(worst < 16u) == ~0ull;
#pragma unroll
    for(int h = 0; h < W / 2; ++h) sp.blp[h] = lds_pair + (boff[2 * h] * 4u + boff[2 * h + 1]) * 2u;  // ((b0 * 4 + b1) * 8; boff = code * 4)
    const bool publisher = !last_strip && lane == kWave - 1;
    sp.offx = publisher ? 0u : kLpDrop;
    sp.offb = static_cast<uint32_t>(lane) * (W == 3 ? 12u : 4u);
    sp.offb2 = static_cast<uint32_t>(lane) * 8u;
    sp.rs_in = lp_rsrc(in_x, strip > 0 ? bstride * 4u : 0u);
    sp.rs_out = lp_rsrc(bnd_x, bstride * 4u);
    sp.rs_bits = lp_rsrc(fout_strip, strip_dwords(la, W) * 4u);
    sp.rs_a = lp_rsrc(a, la);
    if(publisher) store_through(&bnd_x[0], sp.mx[W - 1]);  // (boundary index 0 is the margin row)

    LaneState<W> st;
#pragma unroll
    for(int c = 0; c < W; ++c) st.X[c] = st.Y[c] = 0.0f;
#pragma unroll
    for(int p = 0; p < kAccs; ++p) st.acc[p] = 0u;
    st.xlast_old = 0.0f;
    st.zlast = 0.0f;
    LpColAcc ca;
#pragma unroll
    for(int c = 0; c < 3; ++c) ca.a[c] = ca.b[c] = ca.c[c] = 0u;
    uint32_t arow = lane == 0 ? static_cast<uint32_t>(a[0]) * (kTabStride * 4u) : 0u;
    float s[W];
#pragma unroll
    for(int c = 0; c < W; ++c) s[c] = *reinterpret_cast<const float*>(tab_bytes + arow + boff[c]);

    // The chunk of rows r0 .. r0+15 from its raw loads (lanes 0..15): the left neighbour's values once they are all
    // there (they were 0xffffffff before the launch; rows that are not come again, bypassing the L2, until they are),
    // the matrix' column 0 for the first strip (align_pair.cc:82-86).
    float bx = kLowest, bz = kLowest;
    uint32_t ach = 0;
    auto take_chunk = [&](uint32_t r0, uint32_t xb, uint32_t zb, uint32_t code) {
        const uint32_t crow = r0 + static_cast<uint32_t>(lane);
        const bool mine = lane < static_cast<int>(kLpRows) && crow < la;
        bx = bz = kLowest;
        if(strip == 0) {
            if(mine) bx = crow == 0 ? (0.0f + k.ng) + k.ng : ((k.ng + k.go) + k.ge * static_cast<float>(crow - 1)) + k.gs;
        } else {
            // (the first test is on the values the block loaded; the loop's own loads -- and the waits for them, which
            // also wait for the block's stores -- stay on the slow path)
            auto all_there = [&] { return __builtin_amdgcn_ballot_w64(!mine || (xb != 0xffffffffu && zb != 0xffffffffu)) == ~0ull; };
            if(__builtin_expect(!all_there(), 0)) {
                for(uint32_t spins = 0;; ++spins) {
                    if(spins > (1u << 24)) {
                        handoff_ok = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    if(mine) {
                        xb = __hip_atomic_load(reinterpret_cast<const uint32_t*>(in_x) + 2u * crow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        zb = __hip_atomic_load(reinterpret_cast<const uint32_t*>(in_x) + 2u * crow + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if(all_there()) break;
                }
            }
            if(mine) {
                bx = __builtin_bit_cast(float, xb);
                bz = __builtin_bit_cast(float, zb);
            }
        }
        ach = code * (kTabStride * 4u);  // (0 beyond the last row: the load is range-checked)
        asm volatile("" : "+v"(ach), "+v"(bx), "+v"(bz));
    };
    {
        const uint32_t crow = static_cast<uint32_t>(lane);
        take_chunk(0, 0xffffffffu, 0xffffffffu, crow + 1 < la ? static_cast<uint32_t>(a[crow + 1]) : 0u);
    }
    uint32_t kbase = 0;
    // (two loops, not one with a branch: the state stays in the registers the blocks name)
    for(; kbase < static_cast<uint32_t>(kWave) && kbase + kLpRows <= nsteps; kbase += kLpRows) {
        uint64_t nxz;
        uint32_t na;
        lp_block<W, true>(sp, st, arow, s, kbase, lane, la, bx, bz, ach, nxz, na, ca);
        // step 63 is lane 63's first row
        if(kbase + kLpRows == static_cast<uint32_t>(kWave) && publisher) {
            store_through(&bnd_x[2], st.X[W - 1]);
            store_through(&bnd_x[1], st.zlast);
        }
        take_chunk(kbase + kLpRows, static_cast<uint32_t>(nxz), static_cast<uint32_t>(nxz >> 32), na);
    }
    for(; kbase + kLpRows <= nsteps; kbase += kLpRows) {
        uint64_t nxz;
        uint32_t na;
        lp_block<W, false>(sp, st, arow, s, kbase, lane, la, bx, bz, ach, nxz, na, ca);
        take_chunk(kbase + kLpRows, static_cast<uint32_t>(nxz), static_cast<uint32_t>(nxz >> 32), na);
    }
    for(uint32_t kk = 0; kbase + kk < nsteps; ++kk) {
        if constexpr(W == 3) {
            if(kbase < static_cast<uint32_t>(kWave))
                lp3_tail_step<true>(sp, st, ca, arow, s, boff, lds_tab, kbase + kk, kk, lane, la, last_strip, fout_strip, bnd_x, ach, bx, bz);
            else
                lp3_tail_step<false>(sp, st, ca, arow, s, boff, lds_tab, kbase + kk, kk, lane, la, last_strip, fout_strip, bnd_x, ach, bx, bz);
        } else {
            if(kbase < static_cast<uint32_t>(kWave))
                lp_tail_step<W, true>(sp, st, arow, s, boff, lds_tab, kbase + kk, kk, lane, la, last_strip, fout, bnd_x, ach, bx, bz);
            else
                lp_tail_step<W, false>(sp, st, arow, s, boff, lds_tab, kbase + kk, kk, lane, la, last_strip, fout, bnd_x, ach, bx, bz);
        }
    }
    // score = X of the last body cell (align_pair.cc:130-138,265): held by the lane of the last column after the last step
    const int last_lane = static_cast<int>((lb - 1 - col0) / W), last_c = static_cast<int>((lb - 1 - col0) % W);
    if(last_strip && lane == last_lane) {
        float sc = st.X[0];
#pragma unroll
        for(int c = 1; c < W; ++c) sc = (c == last_c) ? st.X[c] : sc;
        scores[pair] = sc;
    }
    // flush the accumulators of an incomplete last dword, left-aligned (layout in common.hpp)
    if constexpr(W == 3) {
        const uint32_t ra = nsteps & 15u;
        if(ra != 0) lp3_store_group(fout_strip + static_cast<uint64_t>(nsteps >> 4) * kLp3GroupDwords, lane, ca, 16u - ra);
    } else {
        constexpr uint32_t kMA = 16 / W, kMC = 32 / W;
        const uint32_t g = nsteps / kMC, q = nsteps & (kMC - 1u);
        const uint32_t ra = nsteps & (kMA - 1u);
        if(ra != 0) {
            uint32_t* dst = fout + static_cast<uint64_t>(g) * kPairDwords + (q / kMA) * (2 * kWave);
            dst[0] = st.acc[ACC_A] << (32u - 2u * W * ra);
            dst[kWave] = st.acc[ACC_B] << (32u - 2u * W * ra);
        }
        if(q != 0) fout[static_cast<uint64_t>(g) * kPairDwords + 4 * kWave] = st.acc[ACC_C] << (32u - W * q);
    }
    if(strip > 0) {
        // (the chain "every earlier strip has released its decision bits" runs through the progress words)
        handoff_ok = handoff_ok && wait_progress(progress + ticket - 1, la);
        if(__hip_atomic_load(progress + ticket - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kHandoffPoison) handoff_ok = false;
    }
    if(!last_strip) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        publish_progress(progress + ticket, handoff_ok ? la : kHandoffPoison, lane == kWave - 1);
    }
    return handoff_ok;
}

// where the decision word of body cell (bi, bj) for a walk in state `st` lies (dword index from the pair's first strip) and
// the shift that brings its bits to the bottom: common.hpp's layouts for this one shape -- shifts and constant divisions only
template <int W>
__device__ __forceinline__ void lp_lookup_addr(uint64_t sd, uint32_t bi, uint32_t bj, int st, uint64_t& idx, uint32_t& shift) {
    if constexpr(W == 3) {
        const uint32_t strip = bj / (3u * kWave), colin = bj - strip * (3u * kWave), t = colin / 3u, c = colin - 3u * t;
        const uint32_t kstep = bi + t, g = kstep >> 4, tt = kstep & 15u;
        const uint64_t base = strip * sd + static_cast<uint64_t>(g) * kLp3GroupDwords;
        if(st == COATI_HIP_OP_INS) {  // (wave-uniform)
            idx = base + 6u * kWave + 2u * t + (c >> 1);
            shift = ((c & 1u) << 4) + 15u - tt;
        } else {
            idx = base + (st == COATI_HIP_OP_DEL ? 3u * kWave : 0u) + 3u * t + c;
            shift = 30u - 2u * tt;
        }
    } else {
        constexpr uint32_t lgW = W == 4 ? 2u : 1u, lg_mc = 5u - lgW, lg_ma = 4u - lgW, lg_cols = 6u + lgW;
        const uint32_t strip = bj >> lg_cols, colin = bj & ((1u << lg_cols) - 1u), t = colin >> lgW, c = colin & (W - 1u);
        const uint32_t kstep = bi + t, g = kstep >> lg_mc, q = kstep & ((1u << lg_mc) - 1u);
        const uint64_t grp = strip * sd + static_cast<uint64_t>(g) * kPairDwords + t;
        if(st == COATI_HIP_OP_INS) {  // (wave-uniform)
            idx = grp + 4 * kWave;
            shift = 31u - ((q << lgW) + c);
        } else {
            const uint32_t half = q >> lg_ma, tt = q & ((1u << lg_ma) - 1u);
            idx = grp + half * (2u * kWave) + (st == COATI_HIP_OP_DEL ? kWave : 0);
            shift = 30u - 2u * ((tt << lgW) + c);
        }
    }
}
// the dword a lane asks for AHEAD of the walk (below): slot 0 .. 127 of the request -> one row of one group of steps on the
// diagonal `64 + ...` moves ahead; false: nothing to ask for
template <int W>
__device__ __forceinline__ bool lp_ahead_addr(uint64_t sd, uint32_t i, uint32_t j, uint32_t slot, uint64_t& idx) {
    if constexpr(W == 3) {
        // A group is 16 steps = 12 diagonal moves (a move is 1 + 1/3 steps); the diagonal crosses the lanes t - 4 ... t of it.  What the
        // walk may look up there: A words along the diagonal; B words of a deletion run that comes UP a column from the groups
        // after it (the diagonal was up to 16 lanes further right there); C words of an insertion run along a row (up to 21 lanes
        // further left).  A lane's words are 12, 12 and 8 bytes, a cache line 10 or 16 lanes: eight requests per group -- A at lanes
        // t - 4, t; B at t - 4, t + 6, t + 16; C at t - 21, t - 10, t -- 16 groups: 64 ... 256 moves ahead.
        const uint32_t m = slot >> 3, piece = slot & 7u, ahead = 64u + m * 12u;
        if(i <= ahead || j <= ahead) return false;
        const uint32_t bi = i - ahead - 1u, bj = j - ahead - 1u;
        const uint32_t strip = bj / (3u * kWave), t = (bj - strip * (3u * kWave)) / 3u, g = (bi + t) >> 4;
        constexpr int kLaneOff[8] = {-4, 0, -4, 6, 16, -21, -10, 0};
        const int tl = min(max(static_cast<int>(t) + kLaneOff[piece], 0), kWave - 1);
        const uint32_t blk = piece < 2u ? 0u : (piece < 5u ? 1u : 2u);  // A, B, C
        idx = strip * sd + static_cast<uint64_t>(g) * kLp3GroupDwords + (blk < 2u ? blk * 3u * kWave + 3u * tl : 6u * kWave + 2u * tl);
        return true;
    } else {
        constexpr uint32_t lgW = W == 4 ? 2u : 1u, lg_mc = 5u - lgW, lg_cols = 6u + lgW;
        const uint32_t m = slot / 5u, row = slot % 5u;  // (slot = lane + 64 * half: m = lane / 5 + 13 * half as before for half 0; 12.8 -> 13 groups per half)
        const uint32_t ahead = 64u + (m * 32u) / 5u;  // moves along the diagonal: a group of 8 steps is 6.4 of them (W = 4)
        if(i <= ahead || j <= ahead) return false;
        const uint32_t bi = i - ahead - 1u, bj = j - ahead - 1u;
        const uint32_t strip = bj >> lg_cols, t = (bj & ((1u << lg_cols) - 1u)) >> lgW, g = (bi + t) >> lg_mc;
        idx = strip * sd + static_cast<uint64_t>(g) * kPairDwords + row * kWave + t;
        return true;
    }
}

// ---- the walk away from the margins (round 6) --------------------------------------------------------------------------
// The round-5 loop (kept below for the last 64 rows / columns) took 0.6 us per iteration on the 160 kb pair, and its lookup
// was the smaller part: ~250 instructions with a dozen taken branches (a lone wavefront pays ~4.5 cycles for each), 64-bit
// index arithmetic, margin code, and an `s_waitcnt vmcnt(0)` in front of the lookup's load that drained the ops store of the
// iteration before (the counter retires in order).  Here an iteration is compiled for its kind of move (ST), works with 32-bit
// offsets into a window of two strips behind a buffer descriptor, and ONE asm statement issues the lookup's load, the ops
// store of the iteration BEFORE (dropped by its offset where there is nothing to store) and -- when the walk enters a new
// window of 64 diagonal moves -- the requests for the words further ahead, and waits for the lookup alone: `vmcnt(n)` with the
// n operations behind it.  The requests are LDS-DMA loads: no register that would have to stay free while they are in flight
// (an asm load's VGPR destination is the compiler's to reuse at once); what they write -- 256 bytes of this wavefront's table,
// which the walk does not read and the kernel reloads for the next item -- is ignored; M0 is the compiler's, saved and
// restored inside the statement.  s_nop 4 -- here and in front of EVERY asm memory instruction of this file that takes a
// descriptor: its SGPRs may have been written by scalar instructions just before (five wait states; the hazard recognizer
// does not look into inline asm: DESIGN.md 5.2, 5.30 -- a flush without it lost a bridge's ops in round 6).
struct LpWalk {
    uint32_t i, j, pos32;        // matrix cell the walk is at; ops written so far end here (within the pair's slot)
    int st;
    uint32_t pend_off, pend_st;  // the ops store of the iteration before (per lane; kLpDrop: nothing to store)
    uint32_t pf_window, strip_held;
    uint64_t base;               // address of the window's first strip
    // the spliced traceback (MODE 1 and 2 of lp_walk_iter)
    bool fresh;                  // the iteration before ended a run: the walk is at a run start
    bool exited;                 // (MODE 2) the speculative walk has left its strip; exit_mv = the move that took it out
    bool hit;                    // the walk stands on a run start of the strip's recorded walk: e_* = that entry
    uint32_t exit_mv, e_cnt, e_exit_i, e_exit_mv, e_total;
    uint32_t list_strip;         // the true walk holds the run starts of this strip's record, one entry per lane: ...
    u32x4 ea, eb;                // ... {i, j, state, ops before}, {exit row, arriving move, ops in all, check word}
    u32x4 ha, hb;                // ... and the strip's bridge header (common.hpp), the same in every lane
#ifdef COATI_FILL_TRACE
    uint64_t tr_iter, tr_wait, tr_ask, tr_hits;
#endif
};
__device__ __forceinline__ uint8_t* lp_splice_area(float* bnd, const PairDesc& pd, uint32_t strip) {
    return reinterpret_cast<uint8_t*>(bnd + pd.bnd_off + lp_splice_first_float(pd.la, pd.v_strips)) + static_cast<uint64_t>(strip) * kSpStrideBytes;
}
constexpr uint32_t kSpCheck = 0x5a5a5a5au, kSpCheckB = 0x3c3c3c3cu;
// a strip whose speculative walk left no record says so in its list's first word, so that its left neighbour does not wait for one
constexpr uint32_t kSpNoRecord = 0xfffffffeu;
__device__ __forceinline__ void lp_mark_no_record(uint8_t* area, int lane) {
    if(lane == 0) __hip_atomic_store(reinterpret_cast<uint32_t*>(area + kSpOps), kSpNoRecord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// MODE 0: the walk.  MODE 2: a strip's speculative walk -- confined to its strip (lanes whose cell lies left of body column
// `col0` end the run; landing there is the exit), its ops stored through the L2 into the strip's record area, no requests
// ahead (the strip's words are this wavefront's own, fresh in its L2).
template <int W, int ST, int MODE>
__device__ __forceinline__ void lp_walk_iter(LpWalk& w, int lane, const uint32_t* __restrict__ fl, uint64_t sd, uint32_t win_bytes,
                                             const u32x4& rs_ops, uint32_t lds_sink, uint32_t col0) {
    constexpr uint32_t di = ST != COATI_HIP_OP_INS ? 1u : 0u, dj = ST != COATI_HIP_OP_DEL ? 1u : 0u;
    constexpr uint32_t kCols = kWave * W;
    // the window: the strip of column j - 1 and the one before it (the 64 cells ahead span at most two)
    const uint32_t strip_cur = (w.j - 1u) / kCols, strip0 = strip_cur > 0u ? strip_cur - 1u : 0u;
    if(strip0 != w.strip_held) {  // (wave-uniform; once per strip)
        w.strip_held = strip0;
        const uint64_t a = reinterpret_cast<uint64_t>(fl + strip0 * sd);
        w.base = (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(a >> 32)))) << 32) |
                 static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(a)));
    }
    u32x4 rs_fl;  // (readfirstlane: free where the value is in SGPRs already, and keeps the descriptor out of VGPRs where the compiler is not sure)
    rs_fl.x = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(w.base))));
    rs_fl.y = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(w.base >> 32) & 0xffffu)));
    rs_fl.z = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(win_bytes)));
    rs_fl.w = 0x00020000u;
    const uint32_t step = static_cast<uint32_t>(lane) + 1u;
    const uint32_t bi = w.i - 1u - di * step, bj = w.j - 1u - dj * step;
    const uint32_t sd4 = static_cast<uint32_t>(sd) * 4u;
    uint32_t off, lsh;  // byte offset in the window; the cell's bits are the top bits of (word << lsh)
    if constexpr(W == 3) {
        const uint32_t q3 = bj / 3u, strip = q3 >> 6, t = q3 & 63u, c = bj - 3u * q3;
        const uint32_t kstep = bi + t, g = kstep >> 4;
        const uint32_t row = (strip != strip0 ? sd4 : 0u) + g * (kLp3GroupDwords * 4u);
        if constexpr(ST == COATI_HIP_OP_INS) {
            off = row + 6u * kWave * 4u + t * 8u + c * 2u;  // (the short itself: an aligned 16-bit load)
            lsh = 16u + (kstep & 15u);
        } else {
            off = row + (ST == COATI_HIP_OP_DEL ? 3u * kWave * 4u : 0u) + t * 12u + c * 4u;
            lsh = (kstep & 15u) << 1;
        }
    } else {
        constexpr uint32_t lgW = W == 4 ? 2u : 1u, lg_mc = 5u - lgW, lg_ma = 4u - lgW, lg_cols = 6u + lgW;
        const uint32_t strip = bj >> lg_cols, colin = bj & ((1u << lg_cols) - 1u), t = colin >> lgW, c = colin & (W - 1u);
        const uint32_t kstep = bi + t, g = kstep >> lg_mc, q = kstep & ((1u << lg_mc) - 1u);
        const uint32_t row = (strip != strip0 ? sd4 : 0u) + g * (kPairDwords * 4u) + t * 4u;
        if constexpr(ST == COATI_HIP_OP_INS) {
            off = row + 4u * kWave * 4u;
            lsh = (q << lgW) + c;
        } else {
            const uint32_t half = q >> lg_ma, tt = q & ((1u << lg_ma) - 1u);
            off = row + half * (2u * kWave * 4u) + (ST == COATI_HIP_OP_DEL ? kWave * 4u : 0u);
            lsh = 2u * ((tt << lgW) + c);
        }
    }
#ifdef COATI_WALK_NO_ASK  // (experiment: the walk without its requests ahead)
    const bool ask = false;
#else
    const bool ask = MODE != 2 && ((w.i + w.j) >> 7) != w.pf_window;  // (wave-uniform)
#endif
    uint64_t p0 = reinterpret_cast<uint64_t>(fl), p1 = p0;
    if(ask) {
        w.pf_window = (w.i + w.j) >> 7;
        uint64_t idx;
        if(lp_ahead_addr<W>(sd, w.i, w.j, static_cast<uint32_t>(lane), idx)) p0 = reinterpret_cast<uint64_t>(fl + idx);
        if(lp_ahead_addr<W>(sd, w.i, w.j, static_cast<uint32_t>(lane) + kWave + (W == 3 ? 0u : 1u), idx)) p1 = reinterpret_cast<uint64_t>(fl + idx);
    }
#if defined(COATI_FILL_TRACE) && defined(COATI_WALK_TRACE_WAIT)
    const uint64_t tr_t0 = __builtin_amdgcn_s_memtime();
#endif
    uint32_t word;
#define COATI_LP_WALK_ASK_TAIL                                                                                                 \
    "s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\t"                                                           \
    "global_load_lds_dword %[p0], off\n\tglobal_load_lds_dword %[p1], off\n\t"                                                \
    "s_mov_b32 m0, %[keep]\n\ts_waitcnt vmcnt(3)"
#define COATI_LP_WALK_STEP(LOAD, STORE_BITS)                                                                                   \
    if(ask) {                                                                                                                  \
        uint32_t keep;                                                                                                         \
        asm volatile("s_nop 4\n\t" LOAD " %[w], %[off], %[rs], 0 offen\n\t"                                                   \
                     "buffer_store_byte %[pst], %[poff], %[rso], 0 offen" STORE_BITS "\n\t" COATI_LP_WALK_ASK_TAIL             \
                     : [w] "=&v"(word), [keep] "=&s"(keep)                                                                     \
                     : [off] "v"(off), [rs] "s"(rs_fl), [pst] "v"(w.pend_st), [poff] "v"(w.pend_off), [rso] "s"(rs_ops),       \
                       [lds] "s"(lds_sink), [p0] "v"(p0), [p1] "v"(p1)                                                         \
                     : "memory");                                                                                              \
    } else {                                                                                                                   \
        asm volatile("s_nop 4\n\t" LOAD " %[w], %[off], %[rs], 0 offen\n\t"                                                   \
                     "buffer_store_byte %[pst], %[poff], %[rso], 0 offen" STORE_BITS "\n\t"                                    \
                     "s_waitcnt vmcnt(1)"                                                                                      \
                     : [w] "=&v"(word)                                                                                         \
                     : [off] "v"(off), [rs] "s"(rs_fl), [pst] "v"(w.pend_st), [poff] "v"(w.pend_off), [rso] "s"(rs_ops)        \
                     : "memory");                                                                                              \
    }
    if constexpr(MODE == 2) {
        if constexpr(W == 3 && ST == COATI_HIP_OP_INS) {
            COATI_LP_WALK_STEP("buffer_load_ushort", " sc1")
        } else {
            COATI_LP_WALK_STEP("buffer_load_dword", " sc1")
        }
    } else {
        if constexpr(W == 3 && ST == COATI_HIP_OP_INS) {
            COATI_LP_WALK_STEP("buffer_load_ushort", "")
        } else {
            COATI_LP_WALK_STEP("buffer_load_dword", "")
        }
    }
#undef COATI_LP_WALK_STEP
#undef COATI_LP_WALK_ASK_TAIL
#ifdef COATI_FILL_TRACE
    // (trace build: iterations and windows asked ahead; -DCOATI_WALK_TRACE_WAIT also times the statement above -- two SMEM round
    // trips per iteration, which then weigh more than what they measure)
    if constexpr(MODE != 2) {
        w.tr_iter += 1;
        w.tr_ask += ask ? 1u : 0u;
    }
#ifdef COATI_WALK_TRACE_WAIT
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    w.tr_wait += __builtin_amdgcn_s_memtime() - tr_t0;
#endif
#endif
    // the decision of every lane's cell: does the run go on there?  (max_mdi's two bits: bit 1 = the M argument is not the
    // maximum, bit 0 = the D argument is not; max_mi's one: M is the maximum)
    const uint32_t top = word << lsh;
    bool go_on;
    uint32_t code;
    if constexpr(ST == COATI_HIP_OP_INS) {
        code = top >> 31;
        go_on = code == 0u;
    } else {
        code = top >> 30;
        go_on = ST == COATI_HIP_OP_MATCH ? code < 2u : code == 2u;
    }
    if constexpr(MODE == 2 && dj != 0u) go_on = go_on && bj >= col0;
    const unsigned long long cont = __builtin_amdgcn_ballot_w64(go_on);
    const uint32_t run = cont == ~0ull ? kWave : static_cast<uint32_t>(__builtin_ctzll(~cont));
    const uint32_t moves = run == kWave ? kWave : run + 1u;
    w.pend_off = static_cast<uint32_t>(lane) < moves ? w.pos32 - 1u - static_cast<uint32_t>(lane) : kLpDrop;
    w.pend_st = static_cast<uint32_t>(ST);
    w.pos32 -= moves;
    w.i -= di * moves;
    w.j -= dj * moves;
    w.fresh = run < kWave;
    if(run < kWave) {
        if constexpr(MODE == 2 && dj != 0u) {
            if(w.j <= col0) {  // (the move landed left of the strip: body column j - 1 = col0 - 1)
                w.exited = true;
                w.exit_mv = static_cast<uint32_t>(ST);
                return;
            }
        }
        const uint32_t cr = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(code), static_cast<int>(run)));
        if constexpr(ST == COATI_HIP_OP_INS) w.st = COATI_HIP_OP_MATCH;  // (the run ended: M is the maximum)
        else w.st = !(cr & 2u) ? COATI_HIP_OP_MATCH : ((cr & 1u) ? COATI_HIP_OP_INS : COATI_HIP_OP_DEL);
    }
}
template <int W, int MODE>
__device__ __forceinline__ void lp_walk_step(LpWalk& w, int lane, const uint32_t* __restrict__ fl, uint64_t sd, uint32_t win_bytes,
                                             const u32x4& rs_ops, uint32_t lds_sink, uint32_t col0) {
    if(w.st == COATI_HIP_OP_MATCH) lp_walk_iter<W, COATI_HIP_OP_MATCH, MODE>(w, lane, fl, sd, win_bytes, rs_ops, lds_sink, col0);
    else if(w.st == COATI_HIP_OP_DEL) lp_walk_iter<W, COATI_HIP_OP_DEL, MODE>(w, lane, fl, sd, win_bytes, rs_ops, lds_sink, col0);
    else lp_walk_iter<W, COATI_HIP_OP_INS, MODE>(w, lane, fl, sd, win_bytes, rs_ops, lds_sink, col0);
}
__device__ __forceinline__ void lp_walk_init(LpWalk& w, uint32_t i, uint32_t j, uint32_t pos32, int st) {
    w.i = i;
    w.j = j;
    w.pos32 = pos32;
    w.st = st;
    w.pend_off = kLpDrop;
    w.pend_st = 0u;
    w.pf_window = 0xffffffffu;
    w.strip_held = 0xffffffffu;
    w.base = 0;
    w.fresh = true;
    w.exited = w.hit = false;
    w.exit_mv = w.e_cnt = w.e_exit_i = w.e_exit_mv = w.e_total = 0u;
    w.list_strip = 0xffffffffu;
    w.ea = w.eb = w.ha = w.hb = u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
#ifdef COATI_FILL_TRACE
    w.tr_iter = w.tr_wait = w.tr_ask = w.tr_hits = 0;
#endif
}

// The spliced traceback (round 6).  A long pair's walk was ONE wavefront following the path through hundreds of strips
// while the strips' own wavefronts, their fill done, idled: 10 333 dependent iterations for the 160 kb pair.  Now every
// strip's wavefront, as soon as its fill is done, walks ITS strip speculatively: from the strip's right edge, at a row BELOW
// where the pair's straight line crosses it (a walk that starts below the true path climbs to it by a deletion run and
// follows it from there: a walk from ABOVE would have to leave the strip sideways), with the same iterations (MODE 2),
// confined to the strip's own decision words, until it leaves the strip on the left.  It records its ops (right to left) and
// a list of its run starts (up to 64): {cell, state, ops recorded before it; exit row, arriving move, ops in all}.  The true
// walk, entering a strip, loads that list -- one entry per lane, the lines asked for a strip earlier -- and before every lookup
// compares its position with all 64 at once; a hit means both walks stand in the same cell in the same state, so the rest of
// the strip is the recorded one: it copies the recorded ops behind the hit and continues from the recorded exit with the
// lookup of the arrival state -- traceback<tropical>'s own walk (align_pair.cc:249-303) from there on.  No hit in a strip
// (records are absent where the walk came near a margin, overflowed its 1 024 bytes, or -- COATI_HIP_LP_SPLICE=miss -- by
// request): the walk goes through the strip as before.  Nothing here decides anything: the ops are the walk's own, earlier.
template <int W>
__device__ __forceinline__ void lp_spec_walk(int lane, const PairDesc& pd, uint32_t strip, const uint32_t* __restrict__ flags,
                                             float* __restrict__ bnd, uint32_t lds_scratch, uint32_t mode) {
    constexpr uint32_t kCols = kWave * W;
    const uint32_t la = pd.la, lb = pd.lb, col0 = strip * kCols, j_e = col0 + kCols;  // (only strips of full width are followed by another)
    const uint32_t* __restrict__ fl = flags + pd.flags_off;
    const uint64_t sd = strip_dwords(la, W);
    uint32_t i_e = kWave + 1u;  // (mode 2: a walk that ends at once and leaves no record)
    if(mode != 2u) i_e = static_cast<uint32_t>(std::min<uint64_t>(la, static_cast<uint64_t>(j_e) * la / lb + 256u + ((la + lb) >> 10)));
    uint8_t* area = lp_splice_area(bnd, pd, strip);
    if(i_e <= kWave || j_e <= kWave || col0 == 0u) {
        lp_mark_no_record(area, lane);
        return;
    }
    LpWalk w;
    lp_walk_init(w, i_e, j_e, kSpOps, COATI_HIP_OP_MATCH);
    const u32x4 rs_buf = lp_rsrc(area, kSpOps);
    const uint32_t win_bytes = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(std::min<uint64_t>(2u * sd * 4u, 0x7ffffff0ull))));
    auto* ent = reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(static_cast<uintptr_t>(lds_scratch));
    uint32_t n_ent = 0;
    bool over = false;
    __builtin_amdgcn_s_waitcnt(0x0f70);  // (see walk_pair_lp)
    while(w.st != kWalkEnd && w.i > kWave && w.j > kWave && !w.exited) {
        if(w.pos32 < kWave) {
            over = true;
            break;
        }
        if(w.fresh && n_ent < static_cast<uint32_t>(kWave)) {
            if(lane == 0) {
                ent[4u * n_ent + 0u] = w.i;
                ent[4u * n_ent + 1u] = w.j;
                ent[4u * n_ent + 2u] = static_cast<uint32_t>(w.st);
                ent[4u * n_ent + 3u] = kSpOps - w.pos32;
            }
            ++n_ent;
        }
        lp_walk_step<W, 2>(w, lane, fl, sd, win_bytes, rs_buf, 0u, col0);
    }
    asm volatile("s_nop 4\n\tbuffer_store_byte %0, %1, %2, 0 offen sc1\n\ts_waitcnt vmcnt(0)" ::"v"(w.pend_st), "v"(w.pend_off), "s"(rs_buf) : "memory");
    if(!w.exited || over) {
        lp_mark_no_record(area, lane);
        return;
    }
    // every run start's entry, one lane each, through the L2 (both halves carry their own check: a reader may see one without the other)
    u32x4 mya{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    if(static_cast<uint32_t>(lane) < n_ent) {
        const uint32_t e_i = ent[4u * lane + 0u], e_j = ent[4u * lane + 1u], e_st = ent[4u * lane + 2u], e_cnt = ent[4u * lane + 3u];
        const u32x4 h0{e_i, e_j, e_st, e_cnt}, h1{w.i, w.exit_mv, kSpOps - w.pos32, e_i ^ e_j ^ kSpCheck};
        mya = h0;
        const uint64_t dst = reinterpret_cast<uint64_t>(area + kSpOps + static_cast<uint32_t>(lane) * 32u);
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1\n\ts_nop 1\n\ts_waitcnt vmcnt(0)" ::"v"(dst), "v"(h0), "v"(h1) : "memory");
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);
    // ---- the bridge.  The true walk will arrive in this strip where the RIGHT neighbour's recorded walk left that strip (once the
    // walks have met there, its exit is the true one) -- not where this strip's record began -- and would need two or three
    // iterations of its own to meet this record.  So this wavefront does that piece too, now, while the later strips still fill:
    // it waits for the neighbour's record, walks from the neighbour's exit until it stands on a run start of its OWN record (or
    // leaves the strip), and leaves the ops and a header: the true walk, finding its position in the header, copies the bridge and
    // the rest of the record and is through the strip without a lookup of its own.
    if(strip + 2u >= pd.v_strips || mode == 2u) return;  // (the last strip has no record: the walk begins there)
    u32x4 na, nb;
    {
        const uint64_t np = reinterpret_cast<uint64_t>(lp_splice_area(bnd, pd, strip + 1u) + kSpOps);
        bool there = false;
        // (the neighbour's record is ~12 us + its walk behind this strip's own fill; a neighbour without one says so; and the wait is
        // bounded at ~0.2 ms -- this wavefront may have other items to draw: no bridge then)
        for(uint32_t spins = 0; spins < 128u && !there; ++spins) {
            asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(na), "=&v"(nb) : "v"(np) : "memory");
            if(__builtin_amdgcn_readfirstlane(static_cast<int>(na.x)) == static_cast<int>(kSpNoRecord)) return;
            there = __builtin_amdgcn_readfirstlane(static_cast<int>(na.x != 0xffffffffu && nb.w == (na.x ^ na.y ^ kSpCheck))) != 0;
            if(!there) __builtin_amdgcn_s_sleep(32);
        }
        if(!there) return;
    }
    const uint32_t bx_i = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(nb.x)));
    const int bx_mv = __builtin_amdgcn_readfirstlane(static_cast<int>(nb.y));
    if(bx_i <= kWave) return;
    int bst;  // the state after arriving at (bx_i, j_e) by that move: this strip's own decision word (common.hpp state_after)
    {
        uint64_t aidx;
        uint32_t ashift;
        lp_lookup_addr<W>(sd, bx_i - 1u, j_e - 1u, bx_mv, aidx, ashift);
        const uint32_t aw = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(fl[aidx])));
        if(bx_mv == COATI_HIP_OP_INS) {
            bst = ((aw >> ashift) & 1u) ? COATI_HIP_OP_MATCH : COATI_HIP_OP_INS;
        } else {
            const uint32_t two = (aw >> ashift) & 3u;
            bst = !(two & 2u) ? COATI_HIP_OP_MATCH : ((two & 1u) ? COATI_HIP_OP_INS : COATI_HIP_OP_DEL);
        }
    }
    LpWalk b;
    lp_walk_init(b, bx_i, j_e, kSpOpsB, bst);
    const u32x4 rs_b = lp_rsrc(area + kSpBridgeOps, kSpOpsB);
    bool merged = false, over_b = false;
    uint32_t a_cnt = 0;
    __builtin_amdgcn_s_waitcnt(0x0f70);
    while(b.st != kWalkEnd && b.i > kWave && b.j > kWave && !b.exited) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(mya.x == b.i && mya.y == b.j && mya.z == static_cast<uint32_t>(b.st));
        if(m != 0ull) {
            a_cnt = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mya.w), __builtin_ctzll(m)));
            merged = true;
            break;
        }
        if(b.pos32 < kWave) {
            over_b = true;
            break;
        }
        lp_walk_step<W, 2>(b, lane, fl, sd, win_bytes, rs_b, 0u, col0);
    }
    asm volatile("s_nop 4\n\tbuffer_store_byte %0, %1, %2, 0 offen sc1\n\ts_waitcnt vmcnt(0)" ::"v"(b.pend_st), "v"(b.pend_off), "s"(rs_b) : "memory");
    if(over_b || (!merged && !b.exited)) return;
    if(lane == 0) {
        const uint32_t leave_mv = merged ? w.exit_mv : b.exit_mv, exit_i = merged ? w.i : b.i;
        const u32x4 h0{bx_i, j_e, static_cast<uint32_t>(bst) | (leave_mv << 8) | (merged ? 1u << 16 : 0u), kSpOpsB - b.pos32};
        const u32x4 h1{a_cnt, kSpOps - w.pos32, exit_i, bx_i ^ j_e ^ kSpCheckB};
        const uint64_t dst = reinterpret_cast<uint64_t>(area + kSpBridgeHeader);
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1\n\ts_nop 1\n\ts_waitcnt vmcnt(0)" ::"v"(dst), "v"(h0), "v"(h1) : "memory");
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);
}
// entering strip `strip` (1 <= strip < last): the list of its record's run starts, one entry per lane (its lines were asked for a
// strip ago), then the requests for the recorded ops of THIS strip and for the list of the strip after it.  Plain loads: the
// walk's L2 was invalidated before it began, it touches a record once, and the records were stored through the L2 by their
// writers.  The order matters: a writer stores its ops, waits for them, then stores the list's entries -- so a list that was
// read with valid entries proves the ops complete, provided the ops are read AFTER it (hence not a strip ahead with the list:
// a record still being written when its lines were asked for reads as absent or partial -- no match, the walk goes through that
// strip itself -- but ops asked for before the list was seen could be stale under a list that is not).
__device__ __forceinline__ void lp_list_enter(LpWalk& w, int lane, float* __restrict__ bnd, const PairDesc& pd, uint32_t strip, uint32_t lds_sink) {
    w.list_strip = strip;
    uint8_t* area = lp_splice_area(bnd, pd, strip);
    uint8_t* narea = lp_splice_area(bnd, pd, strip >= 2u ? strip - 1u : strip);
    const uint64_t cur = reinterpret_cast<uint64_t>(area + kSpOps + static_cast<uint32_t>(lane) * 32u);
    const uint64_t hdr = reinterpret_cast<uint64_t>(area + kSpBridgeHeader);
    const uint64_t opsp = reinterpret_cast<uint64_t>(area + static_cast<uint32_t>(lane) * 16u);
    const uint64_t nxt = reinterpret_cast<uint64_t>(narea + kSpOps + static_cast<uint32_t>(lane) * 16u);
    const uint64_t nhdr = reinterpret_cast<uint64_t>(narea + kSpBridgeHeader);
    uint32_t keep;
    static_assert(kSpOps == 1024 && kSpEntries * 32u == 2048, "one request of 1 KB for the ops, two for the list");
    asm volatile("global_load_dwordx4 %[ea], %[cur], off\n\tglobal_load_dwordx4 %[eb], %[cur], off offset:16\n\t"
                 "global_load_dwordx4 %[ha], %[hdr], off\n\tglobal_load_dwordx4 %[hb], %[hdr], off offset:16\n\t"
                 "s_waitcnt vmcnt(0)\n\t"
                 "s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %[opsp], off\n\tglobal_load_lds_dwordx4 %[nxt], off\n\tglobal_load_lds_dwordx4 %[nxt], off offset:1024\n\t"
                 "global_load_lds_dword %[nhdr], off\n\t"
                 "s_mov_b32 m0, %[keep]"
                 : [ea] "=&v"(w.ea), [eb] "=&v"(w.eb), [ha] "=&v"(w.ha), [hb] "=&v"(w.hb), [keep] "=&s"(keep)
                 : [cur] "v"(cur), [hdr] "v"(hdr), [opsp] "v"(opsp), [nxt] "v"(nxt), [nhdr] "v"(nhdr), [lds] "s"(lds_sink)
                 : "memory");
}
// The true walk takes a strip's recorded ops: `n_b` bytes of the bridge (the walk found its position in the bridge's header), then
// `n_a` bytes of the record behind its `a_cnt`-th op (it met the record there: w.hit, or the bridge did) -- and goes on from the
// exit (xi, mv).  One memory round trip per 256 bytes of either: the recorded bytes and the decision word of the cell the
// record left the strip into are loaded together; the copies' stores are left in flight (the next lookup's wait covers them: the
// counter retires in order).
template <int W>
__device__ __forceinline__ void lp_splice_apply(LpWalk& w, int lane, const GapConsts& k, const PairDesc& pd, const uint32_t* __restrict__ flags,
                                                float* __restrict__ bnd, const u32x4& rs_ops, uint64_t sd, uint32_t n_b, uint32_t a_cnt,
                                                uint32_t a_total, bool with_a, uint32_t xi, int mv) {
    constexpr uint32_t kCols = kWave * W;
    const uint32_t strip = (w.j - 1u) / kCols;
    const uint32_t n_a = with_a ? a_total - a_cnt : 0u;
    const u32x4 rs_rec = lp_rsrc(lp_splice_area(bnd, pd, strip), kSpStrideBytes);
    const uint32_t xj = strip * kCols;  // the matrix cell the record arrived at when it left the strip: (xi, xj)
    const bool body = xi >= 1u && xj >= 1u;  // (a margin cell: by formula, below)
    uint64_t aidx = 0;
    uint32_t ashift = 0;
    if(body) lp_lookup_addr<W>(sd, xi - 1u, xj - 1u, mv, aidx, ashift);
    const uint64_t ap = reinterpret_cast<uint64_t>(flags + pd.flags_off + aidx);
    uint32_t aword = 0;
    asm volatile("s_nop 4\n\tbuffer_store_byte %0, %1, %2, 0 offen" ::"v"(w.pend_st), "v"(w.pend_off), "s"(rs_ops) : "memory");  // (the iteration before)
    const uint32_t n_max = n_a > n_b ? n_a : n_b;
    for(uint32_t done = 0; done < n_max || done == 0u; done += 4u * kWave) {
        const uint32_t o = done + static_cast<uint32_t>(lane);  // byte of a piece, from its left (lowest) end
        const uint32_t sa = o < n_a ? kSpOps - a_total + o : kLpDrop, sb = o < n_b ? kSpBridgeOps + kSpOpsB - n_b + o : kLpDrop;
        const uint32_t da = w.pos32 - n_b - n_a + o, db = w.pos32 - n_b + o;
        uint32_t a0, a1, a2, a3, b0, b1, b2, b3;
        asm volatile("s_nop 4\n\tbuffer_load_ubyte %0, %9, %11, 0 offen\n\tbuffer_load_ubyte %1, %9, %11, 0 offen offset:64\n\t"
                     "buffer_load_ubyte %2, %9, %11, 0 offen offset:128\n\tbuffer_load_ubyte %3, %9, %11, 0 offen offset:192\n\t"
                     "buffer_load_ubyte %4, %10, %11, 0 offen sc1\n\tbuffer_load_ubyte %5, %10, %11, 0 offen offset:64 sc1\n\t"
                     "buffer_load_ubyte %6, %10, %11, 0 offen offset:128 sc1\n\tbuffer_load_ubyte %7, %10, %11, 0 offen offset:192 sc1\n\t"
                     "global_load_dword %8, %12, off\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(aword)
                     : "v"(sa), "v"(sb), "s"(rs_rec), "v"(ap)
                     : "memory");
        auto to = [](uint32_t o_, uint32_t n, uint32_t d) { return o_ < n ? d : kLpDrop; };
        const uint32_t da0 = to(o, n_a, da), da1 = to(o + kWave, n_a, da + kWave), da2 = to(o + 2u * kWave, n_a, da + 2u * kWave),
                       da3 = to(o + 3u * kWave, n_a, da + 3u * kWave);
        const uint32_t db0 = to(o, n_b, db), db1 = to(o + kWave, n_b, db + kWave), db2 = to(o + 2u * kWave, n_b, db + 2u * kWave),
                       db3 = to(o + 3u * kWave, n_b, db + 3u * kWave);
        asm volatile("s_nop 4\n\tbuffer_store_byte %0, %4, %8, 0 offen\n\tbuffer_store_byte %1, %5, %8, 0 offen\n\t"
                     "buffer_store_byte %2, %6, %8, 0 offen\n\tbuffer_store_byte %3, %7, %8, 0 offen"
                     :
                     : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(da0), "v"(da1), "v"(da2), "v"(da3), "s"(rs_ops)
                     : "memory");
        asm volatile("s_nop 4\n\tbuffer_store_byte %0, %4, %8, 0 offen\n\tbuffer_store_byte %1, %5, %8, 0 offen\n\t"
                     "buffer_store_byte %2, %6, %8, 0 offen\n\tbuffer_store_byte %3, %7, %8, 0 offen"
                     :
                     : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(db0), "v"(db1), "v"(db2), "v"(db3), "s"(rs_ops)
                     : "memory");
    }
    w.pos32 -= n_a + n_b;
    w.i = xi;
    w.j = xj;
    if(body) {  // the state after arriving at (xi, xj) by a move of kind mv: common.hpp state_after on the word loaded above
        const uint32_t aw = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(aword)));
        if(mv == COATI_HIP_OP_INS) {
            w.st = ((aw >> ashift) & 1u) ? COATI_HIP_OP_MATCH : COATI_HIP_OP_INS;
        } else {
            const uint32_t two = (aw >> ashift) & 3u;
            w.st = !(two & 2u) ? COATI_HIP_OP_MATCH : ((two & 1u) ? COATI_HIP_OP_INS : COATI_HIP_OP_DEL);
        }
    } else {
        w.st = __builtin_amdgcn_readfirstlane(arrival_state(k, 1u, flags, pd, xi, xj, mv));
        __builtin_amdgcn_s_waitcnt(0x0f70);  // (the compiler's own load: see walk_pair_lp)
    }
    w.hit = false;
    w.fresh = true;
    w.pend_off = kLpDrop;
#ifdef COATI_FILL_TRACE
    w.tr_hits += 1;
#endif
}

// traceback<tropical> (align_pair.cc:249-303) for a pair whose strips all have W columns per lane: common.hpp's
// wave-cooperative walk_pair (64 lanes look up the states after 1..64 more moves of the current kind, a ballot finds where
// the run ends) with the cell address computed for this one shape -- shifts by constants, one strip size, no compact layout,
// the kind of move as a wave-uniform branch.  A long pair's walk is a chain of ~(moves/64 + 2 x gap runs) such iterations on a
// wavefront that is alone on its SIMD: their instruction count is its time (160 kb pair: 2.4 ms with the general walker).
template <int W>
__device__ __forceinline__ void walk_pair_lp(int lane, const GapConsts& k, const PairDesc& pd, uint32_t pair, int start_state,
                                             const uint32_t* __restrict__ flags, uint8_t* __restrict__ ops,
                                             uint64_t* __restrict__ ops_start, uint32_t* __restrict__ ops_len, uint32_t lds_scratch,
                                             float* __restrict__ bnd, bool splice) {
    const uint32_t la = pd.la, lb = pd.lb;
    const uint32_t* __restrict__ fl = flags + pd.flags_off;
    const uint64_t sd = strip_dwords(la, W);
    // 256 bytes of LDS the walk may scribble on (the wavefront's own table: the kernel reloads it for the next item)
    const uint32_t lds_sink = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(lds_scratch)));
    uint32_t i = la, j = lb;  // matrix coordinates of the last cell (gap_len 1: body cell (i-1, j-1))
    uint64_t pos = pd.ops_off + la + lb;
    int st = (i < 1 && j < 1) ? kWalkEnd : start_state;
    // (The loop for the margins' neighbourhood, below.)  The decision words along the path were written tens of milliseconds ago:
    // a lookup would be a miss all the way to HBM.  So every 64 diagonal moves the wavefront ASKS for the words of the diagonal
    // 64 ... 230 moves ahead -- two loads per lane, nobody waits for them (their values are folded into `sink` one window later) --
    // and the lookups find them in the L2.
    uint32_t pf_window = 0xffffffffu, pf0 = 0u, pf1 = 0u, sink = 0u;
    // Round 6: the walk away from the margins (i, j > 64: every lane's cell is a body cell) as a loop of its own
    // (lp_walk_iter above).
    {
        LpWalk w;
        lp_walk_init(w, i, j, la + lb /* pos - pd.ops_off: within the pair's slot */, st);
#ifdef COATI_FILL_TRACE
        const uint64_t tr_begin = __builtin_amdgcn_s_memtime();
#endif
        const u32x4 rs_ops = lp_rsrc(ops + pd.ops_off, static_cast<uint64_t>(la) + lb);
        const uint32_t win_bytes = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(std::min<uint64_t>(2u * sd * 4u, 0x7ffffff0ull))));
        // (every vector memory operation the COMPILER knows of -- the fill's last stores -- is waited for here, as an instruction
        // its bookkeeping understands: it does not count the loop's asm operations, and a store it believes outstanding at the
        // loop's head would make it drain the counter -- our ops store included -- in every iteration)
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
        constexpr uint32_t kCols = kWave * W;
        while(w.st != kWalkEnd && w.i > kWave && w.j > kWave) {
            const uint32_t strip_cur = (w.j - 1u) / kCols;
            if(splice && strip_cur + 1u < pd.v_strips && strip_cur > 0u) {  // (the last strip has no record: the walk begins there; nor has the first)
                if(strip_cur != w.list_strip) {
                    lp_list_enter(w, lane, bnd, pd, strip_cur, lds_sink);
                    // does the walk stand where the strip's bridge began (it does after a splice in the strip before, once the walks
                    // have met: the bridge started from that record's exit)?  Then the strip is the bridge + the rest of the record.
                    const uint32_t h_i = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w.ha.x)));
                    const uint32_t h_j = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w.ha.y)));
                    const uint32_t h_f = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w.ha.z)));
                    const uint32_t h_c = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w.hb.w)));
                    if(h_i == w.i && h_j == w.j && (h_f & 0xffu) == static_cast<uint32_t>(w.st) && h_c == (w.i ^ w.j ^ kSpCheckB)) {
                        const uint32_t n_b = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w.ha.w)));
                        const uint32_t a_cnt = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w.hb.x)));
                        const uint32_t a_total = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w.hb.y)));
                        const uint32_t x_i = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w.hb.z)));
                        lp_splice_apply<W>(w, lane, k, pd, flags, bnd, rs_ops, sd, n_b, a_cnt, a_total, ((h_f >> 16) & 1u) != 0u, x_i,
                                           static_cast<int>((h_f >> 8) & 0xffu));
#ifdef COATI_FILL_TRACE
                        w.tr_wait += 1;  // (strips taken by their bridge)
#endif
                        continue;
                    }
                }
                // is the walk on one of the record's run starts?  (all 64 entries at once)
                const unsigned long long m = __builtin_amdgcn_ballot_w64(w.ea.x == w.i && w.ea.y == w.j && w.ea.z == static_cast<uint32_t>(w.st) &&
                                                                       w.eb.w == (w.i ^ w.j ^ kSpCheck));
                if(m != 0ull) {
                    const int e = __builtin_ctzll(m);
                    const uint32_t a_cnt = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(w.ea.w), e));
                    const uint32_t x_i = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(w.eb.x), e));
                    const int x_mv = __builtin_amdgcn_readlane(static_cast<int>(w.eb.y), e);
                    const uint32_t a_total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(w.eb.z), e));
                    lp_splice_apply<W>(w, lane, k, pd, flags, bnd, rs_ops, sd, 0u, a_cnt, a_total, true, x_i, x_mv);
                    continue;
                }
            }
            lp_walk_step<W, 0>(w, lane, fl, sd, win_bytes, rs_ops, lds_sink, 0u);
        }
        asm volatile("s_nop 4\n\tbuffer_store_byte %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" ::"v"(w.pend_st), "v"(w.pend_off), "s"(rs_ops) : "memory");
#ifdef COATI_FILL_TRACE
        if(lane == 0 && la > 100000u) {
            g_lp_trace[4000 * 4 + 0] = w.tr_iter;
            g_lp_trace[4000 * 4 + 1] = w.tr_hits | (w.tr_wait << 32);
            g_lp_trace[4000 * 4 + 2] = __builtin_amdgcn_s_memtime() - tr_begin;
            g_lp_trace[4000 * 4 + 3] = w.tr_ask;
        }
#endif
        i = w.i;
        j = w.j;
        st = w.st;
        pos = pd.ops_off + w.pos32;
    }
    // the margins' neighbourhood (and W-generic reference form of the loop above): the last <= 64 rows or columns
    while(st != kWalkEnd) {
        const uint32_t di = st != COATI_HIP_OP_INS ? 1u : 0u, dj = st != COATI_HIP_OP_DEL ? 1u : 0u;
        const uint32_t step = static_cast<uint32_t>(lane) + 1u;
        const bool valid = di * step <= i && dj * step <= j;
        const uint32_t ci = i - di * step, cj = j - dj * step;  // where the walk is after `step` more moves of kind st
        int next = kWalkEnd;
        const bool body = valid && ci >= 1 && cj >= 1;
        uint32_t word = 0u, shift = 0u;
        if(body) {  // the lookup's load first ...
            uint64_t idx;
            lp_lookup_addr<W>(sd, ci - 1, cj - 1, st, idx, shift);
            word = fl[idx];
        }
        // ... then the requests for the words further ahead (loads retire in order: issued BEFORE the lookup they would make it
        // wait for memory too) ...
        if(((i + j) >> 7) != pf_window) {  // (wave-uniform)
            pf_window = (i + j) >> 7;
            sink ^= pf0 ^ pf1;  // last window's loads: long since back
            pf0 = pf1 = 0u;
#pragma unroll
            for(uint32_t half = 0; half < 2u; ++half) {
                uint64_t idx;
                if(lp_ahead_addr<W>(sd, i, j, static_cast<uint32_t>(lane) + kWave * half + (W == 3 ? 0u : half), idx)) {
                    const uint32_t v = fl[idx];
                    if(half == 0) pf0 = v;
                    else pf1 = v;
                }
            }
        }
        // ... then the lookup's value
        if(body) {
            if(st == COATI_HIP_OP_INS) {
                next = ((word >> shift) & 1u) ? COATI_HIP_OP_MATCH : COATI_HIP_OP_INS;
            } else {
                const uint32_t two = (word >> shift) & 3u;
                next = !(two & 2u) ? COATI_HIP_OP_MATCH : ((two & 1u) ? COATI_HIP_OP_INS : COATI_HIP_OP_DEL);
            }
        } else if(valid && (ci >= 1 || cj >= 1)) {  // a margin cell: by formula (align_pair.cc:82-91)
            float m, d, in;
            margin_mdi(k, 1u, ci, cj, m, d, in);
            next = decide_after(k, st, m, d, in);
        }
        if(di > i || dj > j) break;  // (never walk off the matrix)
        const unsigned long long cont = __builtin_amdgcn_ballot_w64(valid && next == st);
        const uint32_t run = cont == ~0ull ? kWave : static_cast<uint32_t>(__builtin_ctzll(~cont));
        const uint32_t moves = run == kWave ? kWave : run + 1u;
        if(static_cast<uint32_t>(lane) < moves) ops[pos - 1 - static_cast<uint32_t>(lane)] = static_cast<uint8_t>(st);
        pos -= moves;
        i -= di * moves;
        j -= dj * moves;
        if(run < kWave) st = __builtin_amdgcn_readlane(next, static_cast<int>(run));
    }
    sink ^= pf0 ^ pf1;
    if(lane == 0) {
        ops_start[pair] = pos;
        ops_len[pair] = static_cast<uint32_t>(pd.ops_off + la + lb - pos);
    }
    // (keeps the requests above alive: no lane has this number)
    if(lane == 64 + static_cast<int>(sink & 1u)) ops_len[pair] = sink;
}

__global__ __launch_bounds__(kFillWaves* kWave, 3) void viterbi_lp(
    const float* __restrict__ table, GapConsts k, const PairDesc* __restrict__ pairs, const WorkItem* __restrict__ items,
    uint32_t n_items, uint32_t* __restrict__ queue, uint32_t* __restrict__ progress, const uint8_t* __restrict__ a_cat,
    const uint8_t* __restrict__ b_cat, uint32_t* __restrict__ flags, float* __restrict__ bnd, float* __restrict__ scores,
    uint8_t* __restrict__ ops, uint64_t* __restrict__ ops_start, uint32_t* __restrict__ ops_len, uint32_t pair_tables, uint32_t splice_mode) {
    __shared__ float tab_all[kFillWaves][kTabRows * kTabStride];
    extern __shared__ float lp_dynamic_lds[];  // pair_tables: one pair table per wavefront; else padding (launch_viterbi_lp)
    const int lane_id = threadIdx.x & (kWave - 1);
    float* tab = tab_all[threadIdx.x / kWave];
    float* ptab = lp_dynamic_lds + (threadIdx.x / kWave) * (kTabRows * kLpPairStride / 4);
    const uint32_t lds_pair = pair_tables != 0 ? static_cast<uint32_t>(reinterpret_cast<uintptr_t>(ptab)) : 0u;
    uint32_t tab_held = 0xffffffffu;
    const char* tab_bytes = reinterpret_cast<const char*>(tab);
    const uint32_t lds_tab = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(tab));  // LDS byte address
#ifdef COATI_FILL_TRACE
    const uint32_t trace_wave = (blockIdx.x * kFillWaves + threadIdx.x / kWave) & 4095u;
    if(lane_id == 0) g_lp_trace[trace_wave * 4] = __builtin_amdgcn_s_memrealtime();
#endif
    for(;;) {
        int lane = lane_id;
        asm volatile("" : "+v"(lane));  // (see viterbi_l1)
        uint32_t ticket = atomicAdd(queue, lane == 0 ? 1u : 0u);
        ticket = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ticket)));
        if(ticket >= n_items) break;
        const WorkItem item = items[ticket];
        const uint32_t pair = item.pair, strip = item.strip;
        const PairDesc pd = pairs[pair];
        bool handoff_ok = true;
        if(pd.table != tab_held) {  // (wave-uniform)
            const float* __restrict__ src = table + static_cast<size_t>(pd.table) * kTabFloats;
            for(int idx = lane; idx < kTabFloats; idx += kWave) {
                const int r = idx / kTabCols, c = idx - r * kTabCols;
                tab[r * kTabStride + c] = src[idx];
            }
            if(pair_tables != 0) {
                // scores of two adjacent A/C/G/T columns side by side: [row][b0 * 4 + b1] = (s(row, b0), s(row, b1))
                for(int idx = lane; idx < kTabRows * 16; idx += kWave) {
                    const int r = idx >> 4, p = idx & 15;
                    const float* __restrict__ row = src + r * kTabCols;
                    ptab[r * (kLpPairStride / 4) + 2 * p] = row[p >> 2];
                    ptab[r * (kLpPairStride / 4) + 2 * p + 1] = row[p & 3];
                }
            }
            tab_held = pd.table;
        }
        if(pd.la > 0 && pd.lb > 0) {  // (every strip of a pair has the pair's shape here)
            if(pd.v_wmain == 2)
                handoff_ok = fill_strip_lp<2>(k, pd, pair, strip, ticket, lane, lds_tab, lds_pair, tab_bytes, a_cat + pd.a_off, b_cat + pd.b_off,
                                              flags, bnd, scores, progress);
            else if(pd.v_wmain == 3)
                handoff_ok = fill_strip_lp<3>(k, pd, pair, strip, ticket, lane, lds_tab, lds_pair, tab_bytes, a_cat + pd.a_off, b_cat + pd.b_off,
                                              flags, bnd, scores, progress);
            else
                handoff_ok = fill_strip_lp<4>(k, pd, pair, strip, ticket, lane, lds_tab, lds_pair, tab_bytes, a_cat + pd.a_off, b_cat + pd.b_off,
                                              flags, bnd, scores, progress);
        }
#ifdef COATI_FILL_TRACE
        if(lane_id == 0) {
            g_lp_trace[trace_wave * 4 + 1] = __builtin_amdgcn_s_memrealtime();
            g_lp_trace[trace_wave * 4 + 3] = strip;
        }
#endif
        if(strip + 1 < pd.v_strips) {
            // the strip's speculative walk for the spliced traceback (lp_spec_walk), while the strips after it still fill
            if(splice_mode != 0u && pd.la > 0 && pd.lb > 0) {
                if(pd.v_wmain == 2) lp_spec_walk<2>(lane, pd, strip, flags, bnd, lds_tab, splice_mode);
                else if(pd.v_wmain == 3) lp_spec_walk<3>(lane, pd, strip, flags, bnd, lds_tab, splice_mode);
                else lp_spec_walk<4>(lane, pd, strip, flags, bnd, lds_tab, splice_mode);
                tab_held = 0xffffffffu;  // (its run starts were collected in this wavefront's table)
            }
            continue;  // the pair's traceback runs on the wavefront of its LAST strip
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if(pd.v_strips > 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if(pd.la > 0 && pd.lb > 0) {
            // max_mdi of the terminal-adjusted last cell == its "after match" decision (common.hpp viterbi_finish)
            const int start_state = __builtin_amdgcn_readfirstlane(state_after(flags, pd, pd.la - 1, pd.lb - 1, COATI_HIP_OP_MATCH));
            if(pd.v_wmain == 2)
                walk_pair_lp<2>(lane, k, pd, pair, start_state, flags, ops, ops_start, ops_len, lds_tab, bnd, splice_mode != 0u && pd.v_strips > 1);
            else if(pd.v_wmain == 3)
                walk_pair_lp<3>(lane, k, pd, pair, start_state, flags, ops, ops_start, ops_len, lds_tab, bnd, splice_mode != 0u && pd.v_strips > 1);
            else
                walk_pair_lp<4>(lane, k, pd, pair, start_state, flags, ops, ops_start, ops_len, lds_tab, bnd, splice_mode != 0u && pd.v_strips > 1);
            tab_held = 0xffffffffu;  // (the walk's requests ahead land in this wavefront's table: reload it for the next item)
        } else {
            viterbi_finish(lane, k, 1u, pd, pair, flags, ops, ops_start, ops_len, scores);  // (margins only)
        }
        if(!handoff_ok && lane == 0) scores[pair] = __builtin_nanf("");  // a producer never arrived (spin bound)
#ifdef COATI_FILL_TRACE
        if(lane_id == 0) g_lp_trace[trace_wave * 4 + 2] = __builtin_amdgcn_s_memrealtime();
#endif
    }
}

}  // namespace

#ifdef COATI_FILL_TRACE
extern "C" int coati_hip_debug_trace_lp(unsigned long long* out) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lp_trace), sizeof(g_lp_trace));
    if(e != hipSuccess) return static_cast<int>(e);
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_lp_trace));
    if(e != hipSuccess) return static_cast<int>(e);
    return static_cast<int>(hipMemset(p, 0, sizeof(g_lp_trace)));  // next launch starts clean
}
#endif

// The launch has one workgroup (one wavefront per SIMD) on every CU while the items fit, up to three after that,
// enforced by LDS padding like viterbi_l1's.
hipError_t launch_viterbi_lp(const BatchDeviceView& v, hipStream_t stream) {
    hipError_t e = zero_queue_and_progress(v, v.n_items, stream);  // ticket counter + polled words: zero every launch
    if(e != hipSuccess) return e;
    if(v.bnd_bytes != 0) {  // the boundary values validate themselves: everything starts as the NaN pattern 0xffffffff
        e = hipMemsetAsync(v.bnd, 0xff, v.bnd_bytes, stream);
        if(e != hipSuccess) return e;
    }
    const uint32_t kCUs = device_cu_count(), kSimds = kCUs * 4;
    int best = static_cast<int>(std::clamp<uint64_t>((static_cast<uint64_t>(v.n_items) + kSimds - 1) / kSimds, 1, 3));
    if(env_options().lp_blocks_per_cu != 0) best = std::clamp(env_options().lp_blocks_per_cu, 1, 3);  // (COATI_HIP_LP_BLOCKS_PER_CU: experiment)
    constexpr size_t kStatic = kFillWaves * kTabRows * kTabStride * sizeof(float);
    constexpr size_t kPerBlock[4] = {0, 96 * 1024, 72 * 1024, 52 * 1024};
    // one workgroup per CU: the dynamic LDS holds a pair table per wavefront (99.6 KB, which also keeps a second workgroup
    // off the CU); more: padding only, single-column gathers.  COATI_HIP_LP_PAIRTAB=0: never (A/B)
    // the spliced traceback (lp_spec_walk): on for plans with multi-strip pairs; COATI_HIP_LP_SPLICE = 0 / 1 / miss forces
    const uint32_t splice_mode = env_options().lp_splice >= 0 ? static_cast<uint32_t>(env_options().lp_splice) : (v.multi_strip ? 1u : 0u);
    const bool pair_ok = !env_options().lp_pairtab_off;
    const bool pair_tables = best == 1 && pair_ok;
    const size_t dyn = pair_tables ? static_cast<size_t>(kFillWaves) * kTabRows * kLpPairStride : kPerBlock[best] - ((kStatic + 255) / 256) * 256;
    if(dyn > 48 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(viterbi_lp), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(dyn));
        if(e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(viterbi_lp, dim3(kCUs * static_cast<uint32_t>(best)), dim3(kFillWaves * kWave), dyn, stream, v.table, v.k, v.pairs,
                       v.items, v.n_items, v.queue, v.progress, v.a_cat, v.b_cat, v.flags, v.bnd, v.scores, v.ops, v.ops_start,
                       v.ops_len, pair_tables ? 1u : 0u, splice_mode);
    return hipGetLastError();
}

}  // namespace coati_hip_detail
