// viterbi_lp: the gap_len-1 Viterbi fill for a FEW LONG pairs (BASELINE configs[2]: one 160 kb pair) --
// the batches whose strip plan has fewer strips than the GPU has SIMDs, so that every strip is 4 columns per
// lane wide and every wavefront is alone on its SIMD.  Same recurrence, same five decision bits per cell in
// the same HBM layout as viterbi_l1.hip (the traceback of common.hpp reads both), same strip pipeline through
// self-validating boundary values; what differs is how a step is issued.
//
// What it replaces in the reference: forward_impl<tropical, align_pair_work_mem_t> (src/lib/align_pair.cc:62-139)
// and traceback<tropical> (align_pair.cc:249-303) for one pair that the CPU tool cannot hold in memory.
//
// A lone wavefront issues one instruction per ~4.3 cycles WHATEVER the instruction
// (profiles/r03/ubench_issue_model.txt), so its time is its instruction count.  viterbi_l1's 4-column step is
// ~175 instructions (27 per cell + ~67 of hand-off, stores, bookkeeping); here a step is ~87:
//   * the cell's eleven additions are six (v_pk_add_f32 does two fp32 additions -- IEEE, the same bits -- in one
//     instruction) and the five sign differences of the decisions three: 19 instructions per cell, LDS gather
//     included (gen_viterbi_lp.py has the list);
//   * 16 steps are ONE block of hand-allocated instruction text: lane 0 takes the strip's left boundary straight
//     from lane j of the chunk registers by a row_shl:j DPP, stores use immediate offsets, cell 3 ping-pongs its
//     state between two register pairs so that the diagonal hand-off needs no copy;
//   * the left boundary arrives in 16-row chunks that are loaded one block AHEAD (at the top of the block before)
//     and checked after it, so a strip follows its left neighbour at 63 + 32 steps, not 63 + 64 + a memory round
//     trip per chunk -- with 626 strips in a 160 kb pair the sum of those lags is a third of the time.
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
    uint32_t offx, offz, offb;  // per-lane offsets: boundary X / Z stores (lane 63, or kLpDrop), decision rows
    u32x4 rs_in, rs_out, rs_bits, rs_a;
    float mx[W], my[W];  // the lane's margin-row state (taken at step == lane)
};

// 16 wavefront steps from `kbase`.  In: the chunk (bx, bz, ach: lane j < 16 holds the strip's left boundary of row
// kbase + j and the table row offset of ancestor row kbase + j + 1).  Out: the raw next chunk (rows kbase + 16 ...).
template <int W, bool kFirst>
__device__ __forceinline__ void lp_block(const LpStrip<W>& sp, LaneState<W>& st, uint32_t& arow, float (&s)[W], uint32_t kbase,
                                         int lane, uint32_t la, float bx, float bz, uint32_t ach, uint32_t& nx, uint32_t& nz,
                                         uint32_t& na) {
    static_assert(W == 4 || W == 2, "gen_viterbi_lp.py writes these two shapes");
    uint32_t arb;
    const uint32_t next = kbase + kLpRows + static_cast<uint32_t>(lane);
    const uint32_t vin_x = next * 4u, vin_z = (la + 1u + next) * 4u, vin_a = next + 1u;
    const uint32_t so_bits = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>((kbase / (32u / W)) * (kPairDwords * 4u))));
    const uint32_t lrel = static_cast<uint32_t>(lane) - kbase;  // (first blocks) the lane starts at step kbase + lrel
    // (main blocks) lane 63 did body row kbase + j - 63 at step j: X of its last column goes to bnd_x[row + 1], Z to bnd_z[row]
    const uint32_t so_out = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>((kbase - (kWave - 1u)) * 4u)));
    // [X:Y] of column c in v[8+2c : 9+2c]; the last column's pair alternates with the pair after it, whose low half is
    // xlast_old between blocks; the scores in v44.. (the other set, v48.., is scratch) -- gen_viterbi_lp.py
#define COATI_LP_COMMON_OUT                                                                                                   \
    "+{v28}"(st.zlast), "+{v44}"(s[0]), "+{v45}"(s[1]), [ara] "+v"(arow), [arb] "=&v"(arb), [aa] "+v"(st.acc[ACC_A]),          \
        [ab] "+v"(st.acc[ACC_B]), [ac] "+v"(st.acc[ACC_C]), [nx] "=&v"(nx), [nz] "=&v"(nz), [na] "=&v"(na)
#define COATI_LP_COMMON_IN                                                                                                    \
    "{v2}"(sp.kv.go), "{v3}"(sp.kv.ng), "{v4}"(sp.kv.ge), "{v5}"(sp.kv.gs), [bx] "v"(bx), [bz] "v"(bz), [ach] "v"(ach),       \
        [bl0] "v"(sp.bl[0]), [bl1] "v"(sp.bl[1]), [blp0] "v"(sp.blp[0]), [offb] "v"(sp.offb), [vin_x] "v"(vin_x),             \
        [vin_z] "v"(vin_z), [vin_a] "v"(vin_a), [rs_in] "s"(sp.rs_in), [rs_bits] "s"(sp.rs_bits), [rs_a] "s"(sp.rs_a),        \
        [so_bits] "s"(so_bits)
#define COATI_LP_FIRST_IN [lrel] "v"(lrel), [mx0] "v"(sp.mx[0]), [mx1] "v"(sp.mx[1]), [my0] "v"(sp.my[0]), [my1] "v"(sp.my[1])
#define COATI_LP_MAIN_IN [offx] "v"(sp.offx), [offz] "v"(sp.offz), [rs_out] "s"(sp.rs_out), [so_out] "s"(so_out)
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
                                             uint32_t la, bool last_strip, uint32_t* fout, float* bnd_x, float* bnd_z, uint32_t ach,
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
        store_through(&bnd_x[r + 1], st.X[W - 1]);
        store_through(&bnd_z[r], st.zlast);
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
    // strip-boundary columns (layout: viterbi_l1.hip fill_strip)
    const uint64_t bstride = 2 * (static_cast<uint64_t>(la) + 1);
    float* __restrict__ bnd_x = bnd + pd.bnd_off + strip * bstride;  // written by this strip
    float* __restrict__ bnd_z = bnd_x + (la + 1);
    const float* __restrict__ in_x = bnd + pd.bnd_off + (strip - 1) * bstride;  // read by it (strip > 0)
    const float* __restrict__ in_z = in_x + (la + 1);
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
    sp.offz = publisher ? (la + 1u) * 4u : kLpDrop;
    sp.offb = static_cast<uint32_t>(lane) * 4u;
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
                        xb = __hip_atomic_load(reinterpret_cast<const uint32_t*>(in_x) + crow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        zb = __hip_atomic_load(reinterpret_cast<const uint32_t*>(in_z) + crow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        uint32_t nx, nz, na;
        lp_block<W, true>(sp, st, arow, s, kbase, lane, la, bx, bz, ach, nx, nz, na);
        // step 63 is lane 63's first row
        if(kbase + kLpRows == static_cast<uint32_t>(kWave) && publisher) {
            store_through(&bnd_x[1], st.X[W - 1]);
            store_through(&bnd_z[0], st.zlast);
        }
        take_chunk(kbase + kLpRows, nx, nz, na);
    }
    for(; kbase + kLpRows <= nsteps; kbase += kLpRows) {
        uint32_t nx, nz, na;
        lp_block<W, false>(sp, st, arow, s, kbase, lane, la, bx, bz, ach, nx, nz, na);
        take_chunk(kbase + kLpRows, nx, nz, na);
    }
    for(uint32_t kk = 0; kbase + kk < nsteps; ++kk) {
        if(kbase < static_cast<uint32_t>(kWave))
            lp_tail_step<W, true>(sp, st, arow, s, boff, lds_tab, kbase + kk, kk, lane, la, last_strip, fout, bnd_x, bnd_z, ach, bx, bz);
        else
            lp_tail_step<W, false>(sp, st, arow, s, boff, lds_tab, kbase + kk, kk, lane, la, last_strip, fout, bnd_x, bnd_z, ach, bx, bz);
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
    {
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

// traceback<tropical> (align_pair.cc:249-303) for a pair whose strips all have W columns per lane: common.hpp's
// wave-cooperative walk_pair (64 lanes look up the states after 1..64 more moves of the current kind, a ballot finds where
// the run ends) with the cell address computed for this one shape -- shifts by constants, one strip size, no compact layout,
// the kind of move as a wave-uniform branch.  A long pair's walk is a chain of ~(moves/64 + 2 x gap runs) such iterations on a
// wavefront that is alone on its SIMD: their instruction count is its time (160 kb pair: 2.4 ms with the general walker).
template <int W>
__device__ __forceinline__ void walk_pair_lp(int lane, const GapConsts& k, const PairDesc& pd, uint32_t pair, int start_state,
                                             const uint32_t* __restrict__ flags, uint8_t* __restrict__ ops,
                                             uint64_t* __restrict__ ops_start, uint32_t* __restrict__ ops_len) {
    constexpr uint32_t lgW = W == 4 ? 2u : 1u, lg_mc = 5u - lgW, lg_ma = 4u - lgW, lg_cols = 6u + lgW;
    const uint32_t la = pd.la, lb = pd.lb;
    const uint32_t* __restrict__ fl = flags + pd.flags_off;
    const uint64_t sd = strip_dwords(la, W);
    uint32_t i = la, j = lb;  // matrix coordinates of the last cell (gap_len 1: body cell (i-1, j-1))
    uint64_t pos = pd.ops_off + la + lb;
    int st = (i < 1 && j < 1) ? kWalkEnd : start_state;
    // The decision words along the path were written tens of milliseconds ago: every iteration's lookup is a miss all the way
    // to HBM, ~0.75 us, and a real pair has thousands of runs (the reference's 160 kb sample: 9 521 runs, 10 333 iterations, ~8 ms
    // of a 44.6-ms launch).  So every 64 diagonal moves the wavefront ASKS for the words of the diagonal 64 ... 230 moves ahead
    // -- 26 groups of steps x their 5 rows, two loads per lane, nobody waits for them (their values are folded into `sink` one
    // window later) -- and the lookups find them in the L2.
    uint32_t pf_window = 0xffffffffu, pf0 = 0u, pf1 = 0u, sink = 0u;
    while(st != kWalkEnd) {
        const uint32_t di = st != COATI_HIP_OP_INS ? 1u : 0u, dj = st != COATI_HIP_OP_DEL ? 1u : 0u;
        const uint32_t step = static_cast<uint32_t>(lane) + 1u;
        const bool valid = di * step <= i && dj * step <= j;
        const uint32_t ci = i - di * step, cj = j - dj * step;  // where the walk is after `step` more moves of kind st
        int next = kWalkEnd;
        const bool body = valid && ci >= 1 && cj >= 1;
        uint32_t word = 0u, shift = 0u;
        if(body) {  // the lookup's load first ...
            const uint32_t bi = ci - 1, bj = cj - 1;
            const uint32_t strip = bj >> lg_cols, colin = bj & ((1u << lg_cols) - 1u), t = colin >> lgW, c = colin & (W - 1u);
            const uint32_t kstep = bi + t, g = kstep >> lg_mc, q = kstep & ((1u << lg_mc) - 1u);
            const uint32_t* __restrict__ grp = fl + (strip * sd + static_cast<uint64_t>(g) * kPairDwords + t);
            if(st == COATI_HIP_OP_INS) {  // (wave-uniform)
                word = grp[4 * kWave];
                shift = 31u - ((q << lgW) + c);
            } else {
                const uint32_t half = q >> lg_ma, tt = q & ((1u << lg_ma) - 1u);
                word = grp[half * (2u * kWave) + (st == COATI_HIP_OP_DEL ? kWave : 0)];
                shift = 30u - 2u * ((tt << lgW) + c);
            }
        }
        // ... then the requests for the words further ahead (loads retire in order: issued BEFORE the lookup they would make it
        // wait for memory too) ...
        if(((i + j) >> 7) != pf_window) {  // (wave-uniform)
            pf_window = (i + j) >> 7;
            sink ^= pf0 ^ pf1;  // last window's loads: long since back
            pf0 = pf1 = 0u;
#pragma unroll
            for(uint32_t half = 0; half < 2u; ++half) {
                const uint32_t m = static_cast<uint32_t>(lane) / 5u + 13u * half, row = static_cast<uint32_t>(lane) % 5u;
                const uint32_t ahead = 64u + (m * 32u) / 5u;  // moves along the diagonal: a group of 8 steps is 6.4 of them (W = 4)
                if(i > ahead && j > ahead) {
                    const uint32_t bi = i - ahead - 1u, bj = j - ahead - 1u;
                    const uint32_t strip = bj >> lg_cols, t = (bj & ((1u << lg_cols) - 1u)) >> lgW, g = (bi + t) >> lg_mc;
                    const uint32_t v = fl[strip * sd + static_cast<uint64_t>(g) * kPairDwords + row * kWave + t];
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
    uint8_t* __restrict__ ops, uint64_t* __restrict__ ops_start, uint32_t* __restrict__ ops_len, uint32_t pair_tables) {
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
        if(strip + 1 < pd.v_strips) continue;  // the pair's traceback runs on the wavefront of its LAST strip
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if(pd.v_strips > 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if(pd.la > 0 && pd.lb > 0) {
            // max_mdi of the terminal-adjusted last cell == its "after match" decision (common.hpp viterbi_finish)
            const int start_state = __builtin_amdgcn_readfirstlane(state_after(flags, pd, pd.la - 1, pd.lb - 1, COATI_HIP_OP_MATCH));
            if(pd.v_wmain == 2)
                walk_pair_lp<2>(lane, k, pd, pair, start_state, flags, ops, ops_start, ops_len);
            else
                walk_pair_lp<4>(lane, k, pd, pair, start_state, flags, ops, ops_start, ops_len);
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
    const bool pair_ok = !env_options().lp_pairtab_off;
    const bool pair_tables = best == 1 && pair_ok;
    const size_t dyn = pair_tables ? static_cast<size_t>(kFillWaves) * kTabRows * kLpPairStride : kPerBlock[best] - ((kStatic + 255) / 256) * 256;
    if(dyn > 48 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(viterbi_lp), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(dyn));
        if(e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(viterbi_lp, dim3(kCUs * static_cast<uint32_t>(best)), dim3(kFillWaves * kWave), dyn, stream, v.table, v.k, v.pairs,
                       v.items, v.n_items, v.queue, v.progress, v.a_cat, v.b_cat, v.flags, v.bnd, v.scores, v.ops, v.ops_start,
                       v.ops_len, pair_tables ? 1u : 0u);
    return hipGetLastError();
}

}  // namespace coati_hip_detail
