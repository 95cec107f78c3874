// Shared definitions of libcoati_hip.so: constants, the per-pair descriptor, the
// HBM layouts, the decision-bit lookups and the wave-cooperative traceback
// walker (used by both Viterbi kernels).
#ifndef COATI_HIP_COMMON_HPP
#define COATI_HIP_COMMON_HPP

#include "coati_hip.h"

#include <hip/hip_runtime.h>

#include "glibc_math.hpp"

#include <cfloat>
#include <cstdint>

namespace coati_hip_detail {

constexpr int kWave = 64;
constexpr int kW = 16;                       // columns per lane
constexpr int kStrip = kWave * kW;           // columns per strip (1024)
constexpr int kTabRows = COATI_HIP_TABLE_ROWS;
constexpr int kTabCols = COATI_HIP_TABLE_COLS;
constexpr int kTabStride = 17;               // LDS row stride in floats (bank spread, measured)
constexpr int kTabFloats = COATI_HIP_TABLE_ROWS * COATI_HIP_TABLE_COLS;  // one table in HBM (2745)
constexpr int kPairDwords = 5 * kWave;       // dwords one pair of wavefront steps stores (320)
constexpr int kFillWaves = 4;                // sequence pairs per workgroup
constexpr float kLowest = -FLT_MAX;          // semiring zero(), semiring.hpp:83,113

// Decision bits per cell.  max_mdi (align_pair.cc:210-224) is the arg-max of its three
// arguments with ties M over D over I.  Bit = 1 means:
//   M1: the M argument is NOT the maximum after a match move   (so the state is D or I)
//   M2: the D argument is NOT the maximum after a match move   (M1 & M2: the state is I)
//   D1, D2: the same two facts after a deletion move (align_pair.cc:285-287)
//   IM: M beats I after an insertion move (max_mi, align_pair.cc:230-232; tie -> I)
// Three per-lane accumulators: A = (M1,M2) pairs, B = (D1,D2) pairs, C = IM.
enum : int { ACC_A = 0, ACC_B = 1, ACC_C = 2, kAccs = 3 };

struct GapConsts {
    float ng, gs, go, ge;  // no_gap, gap_stop, gap_open, gap_extend (log space)
};

// The four gap constants in VECTOR registers.  On gfx950 a v_add_f32 whose constant operand is an SGPR issues
// at the 4-cycle rate of v_max_f32, not at the 2-cycle rate of an add between VGPRs (round 3,
// tools/ubench/gen_issue.py -> profiles/r03/ubench_issue_model.txt: "pure v_add_f32" 2.5 cycles per
// instruction and SIMD, "pure v_add_f32 sgpr" 4.4; the 15-instruction cell 3.03 cycles per instruction with
// SGPR constants, 2.03 with VGPR constants).  Ten of the cell's adds take a constant, so the cells read them
// from four VGPRs that are loaded once per work item and made opaque (or the compiler re-materialises the
// v_mov from the SGPR inside the loop when registers are short).
struct GapVec {
    float ng, gs, go, ge;
};
__device__ __forceinline__ GapVec gap_vec(const GapConsts& k) {
    GapVec v{k.ng, k.gs, k.go, k.ge};
    asm volatile("" : "+v"(v.ng), "+v"(v.gs), "+v"(v.go), "+v"(v.ge));
    return v;
}

struct PairDesc {
    uint64_t a_off, b_off;  // into the concatenated code arrays
    uint64_t flags_off;     // dwords into the bit-plane arena
    uint64_t bnd_off;       // floats into the strip-boundary arena
    uint64_t ops_off;       // slot start in the ops arena (slot = la + lb bytes)
    uint64_t mdi_off;       // floats into the Forward M/D/I arena
    uint32_t la, lb;
    // Viterbi strip plan (decision-bit layout): v_strips strips; all but the last are 64*v_wmain
    // columns wide (v_wmain columns per lane), the last one has v_wlast columns per lane.
    uint32_t v_strips;
    uint8_t v_wmain, v_wlast;
    uint16_t table;  // which of the model's substitution tables this pair uses
    // 0: one stored cell per matrix cell (layout below).  L (2 or 3): "compact" layout of viterbi_k:
    // only the live cells (bi - bj) % L == 0 are stored; the plan counts BLOCK columns (lb / L).
    uint32_t v_compact;
    // Forward M/D/I layout: 0 = every body cell, strips of 64 << f_wlog2 columns (forward_l1: 4, 8 or
    // 16 columns per lane; dp_generic: 16); L (2 or 3) = live cells only, forward_k's block-column
    // strips (forward_k.hip)
    uint16_t f_compact;
    uint8_t f_wlog2;  // log2 of the columns per lane of the Forward strips (0 .. 4)
    // viterbi_ck: the pair's (single) strip is cut into this many ROW parts, each its own work item, continued by
    // whichever wavefront takes the next part (0 or 1: not cut).  Finer items for the ragged end of a launch.
    uint8_t v_parts;
};
// PairDesc::v_parts: the number of row parts in the low four bits; bit 7 = TAPERED parts (below); bits 4-6 = how many
// 64-step chunks the LAST part is shorter than an equal split would make it (round 5: the wavefront of the last part also
// walks the pair -- ~0.17 ms, two chunks' worth -- and the items at the end of the queue are the last parts)
constexpr uint8_t kCkPartsTaper = 0x80;
constexpr uint32_t kCkWalkItemsFlag = 0x80000000u;  // in BatchDeviceView::ck_split_items: the cut pairs' tracebacks are items of their own (part number = number of parts)
__host__ __device__ inline uint32_t ck_parts_count(uint8_t v_parts) { return v_parts & 0x0fu; }
__host__ __device__ inline uint32_t ck_parts_short_last(uint8_t v_parts) { return (v_parts >> 4) & 7u; }
// steps [begin, end) of row part `part` of a strip of nsteps wavefront steps: whole 64-step chunks.  Equal parts, or --
// tapered -- parts whose lengths fall linearly (weights parts, parts - 1, ... 1): what is left of a launch when its
// ticket queue runs dry is at most one item per wavefront, and the items at the end of the queue are the LAST parts
// of the cut pairs, so short last parts shorten the ragged end while long first parts keep the hand-overs few.
__host__ __device__ inline uint32_t ck_part_cut(uint32_t chunks, uint32_t parts, uint32_t p, bool taper, uint32_t short_last = 0) {
    if(p >= parts) return chunks;
    if(!taper) {  // (short_last: the first parts share what the last one gives up; no part is ever left without a chunk)
        const uint32_t cut = p * (chunks + short_last) / parts, cap = chunks > parts - p ? chunks - (parts - p) : 0u;
        return cut < cap ? cut : cap;
    }
    const uint32_t total = parts * (parts + 1u) / 2u, upto = p * parts - p * (p - 1u) / 2u;  // (p = 0: 0 - 0)
    return upto * chunks / total;
}
__host__ __device__ inline void ck_part_range(uint32_t nsteps, uint8_t v_parts, uint32_t part, uint32_t& begin, uint32_t& end) {
    const uint32_t chunks = (nsteps + 63u) / 64u, parts = ck_parts_count(v_parts);
    const bool taper = (v_parts & kCkPartsTaper) != 0;
    begin = 64u * ck_part_cut(chunks, parts, part, taper, ck_parts_short_last(v_parts));
    end = part + 1u == parts ? nsteps : 64u * ck_part_cut(chunks, parts, part + 1u, taper, ck_parts_short_last(v_parts));
}
// every part of a strip of nsteps steps is at least one chunk long
__host__ __device__ inline bool ck_parts_fit(uint32_t nsteps, uint32_t parts, bool taper) {
    const uint32_t chunks = (nsteps + 63u) / 64u;
    return taper ? chunks >= parts * (parts + 1u) / 2u && chunks >= 2u * parts : chunks >= 2u * parts;
}
// the largest shortening <= want of the last part that leaves EVERY part of the real cut (ck_part_cut) at least two chunks:
// the last part begins at (parts - 1) * (chunks + sl) / parts, which reaches `chunks` when (parts - 1) * sl >= chunks
__host__ __device__ inline uint32_t ck_fit_short_last(uint32_t chunks, uint32_t parts, uint32_t want) {
    if(parts < 2u) return 0u;
    for(uint32_t sl = want < 7u ? want : 7u; sl > 0u; --sl) {
        bool ok = true;
        for(uint32_t p = 0; p < parts && ok; ++p) {
            const uint32_t b = p * (chunks + sl) / parts, e = p + 1u == parts ? chunks : (p + 1u) * (chunks + sl) / parts;
            ok = e >= b + 2u;
        }
        if(ok) return sl;
    }
    return 0u;
}
constexpr uint32_t kCkPartStateDwords = 3u * 64u;  // what a part leaves for the next besides the row checkpoint: per lane xlast_old, zlast, table row

// HBM layout of the decision bits of one strip of one pair.  A strip is 64*W descendant
// columns, W = 4, 8 or 16 columns per lane (W = 16 is the full-speed shape; narrower strips are
// for short descendants, for the remainder of a long one and to get more strips in flight when a
// batch has few pairs).  Wavefront step k = body_row + lane.  Steps are stored in groups of
// mC = 32/W steps, 320 dwords per group:
//   [g*320 +   0 + lane]  A of the group's first half      [g*320 + 128 + lane]  A of its second half
//   [g*320 +  64 + lane]  B of the first half              [g*320 + 192 + lane]  B of the second half
//   [g*320 + 256 + lane]  C of the whole group
// An A/B dword holds mA = 16/W steps: step t' (0..mA-1) of its half, column c (0..W-1) at
// position p = t'*W + c: first test at bit 31-2p, second test at bit 30-2p.  The C dword holds
// step q (0..mC-1) of the group, column c at bit 31 - (q*W + c).
// (W = 16: one step per A/B dword, even/odd step = first/second half, C = two steps.)
// Every store is a fully coalesced 256-byte row: 5 bits per DP cell.
// 3 columns per lane (viterbi_lp only, round 6): 3 does not divide the 32-bit words above, so the bits are kept PER COLUMN, in
// groups of 16 steps, LANE-major, kLp3GroupDwords = 512 dwords (2 048 bytes) per group:
//   [g*512 +       3*lane + c]  A of column c (dword: step tt of the group at bits 31-2tt, 30-2tt)
//   [g*512 + 192 + 3*lane + c]  B of column c (same positions)
//   bytes g*2048 + 1536 + 8*lane + 2*c: C of column c (a 16-bit short: step tt at bit 15-tt; 2 bytes per lane unused)
// -- 5.33 bits per cell, three stores per 16 steps (12-, 12- and 8-byte pieces per lane; the 4-column layout: ten), and the
// cells a walk looks up along a diagonal -- a lane's three columns, the neighbouring lanes -- share a cache line.
constexpr uint32_t kLp3GroupDwords = 512;
// viterbi_lp's spliced traceback (round 6; viterbi_lp.hip: lp_spec_walk): behind the boundary arrays of a pair of several
// strips every strip has a record area -- kSpOps bytes of ops its speculative walk recorded (right to left), then a list of up to
// kSpEntries run starts, 32 bytes each: {i, j, state, ops recorded before, exit row, arriving move, ops in all, check word}.
// The areas start every launch as 0xff bytes with the boundary arrays (launch_viterbi_lp): an entry that was not written, or
// only half, matches nothing.
constexpr uint32_t kSpOps = 1024, kSpEntries = 64;
// ... then the BRIDGE (viterbi_lp.hip: lp_spec_walk, second half): kSpOpsB bytes of ops of a second speculative walk, the one that
// starts where the RIGHT neighbour's recorded walk left that strip and runs until it meets this strip's own record (or leaves the
// strip), and its 32-byte header {entry row, entry column, entry state | leaving move << 8 | merged << 16, ops in the bridge, ops the
// own record had made where they met, ops of the own record in all, exit row, check word}.
constexpr uint32_t kSpOpsB = 512, kSpBridgeOps = kSpOps + kSpEntries * 32u, kSpBridgeHeader = kSpBridgeOps + kSpOpsB, kSpStrideBytes = 4096;
static_assert(kSpBridgeHeader + 32u <= kSpStrideBytes, "record area");
__host__ __device__ inline uint64_t lp_splice_first_float(uint32_t la, uint32_t strips) {  // (16-byte aligned)
    return ((static_cast<uint64_t>(strips) - 1u) * 2u * (static_cast<uint64_t>(la) + 1u) + 3u) & ~3ull;
}
__host__ __device__ inline uint64_t lp_splice_floats(uint32_t strips) { return strips > 1u ? static_cast<uint64_t>(strips) * (kSpStrideBytes / 4u) + 4u : 0u; }
// viterbi_ck's spliced traceback for pairs of several strips (round 6, second half; viterbi_ck.hip: ck_walk_pair's modes): the same
// idea on top of the checkpoint walk.  Behind the boundary arrays every strip has a record area of kCkRecBytes:
//   [0, kCkRecOps)                      ops of the strip's speculative walk, right to left
//   [kCkRecList, + 64 * 16)             its run starts {i, j, state, ops recorded before}
//   [kCkRecHead, + 32)                  {exit row, exit column, arriving move, ops in all, magic}
//   [kCkRecBridgeOps, + kCkRecOpsB)     the BRIDGE: ops of the walk from the right neighbour's recorded exit to this strip's record
//   [kCkRecBridgeHead, + 32)            {entry row, entry column, entry move, ops in the bridge, ops the own record had made where
//                                        they met (0xffffffff: they did not), exit row / column / move when they did not meet}
// The areas start every launch as 0xff bytes with the boundary arrays (launch_viterbi_ck): what was not written matches nothing.
constexpr uint32_t kCkRecOps = 4096, kCkRecList = kCkRecOps, kCkRecHead = kCkRecList + 64u * 16u, kCkRecOpsB = 2048,
                   kCkRecBridgeOps = kCkRecHead + 64u, kCkRecBridgeHead = kCkRecBridgeOps + kCkRecOpsB, kCkRecBytes = 8192;
static_assert(kCkRecBridgeHead + 64u <= kCkRecBytes, "record area");
constexpr uint32_t kCkRecMagic = 0x636b7370u;
__host__ __device__ inline uint64_t ck_rec_first_float(uint32_t la, uint32_t strips) {  // (16-byte aligned)
    return ((static_cast<uint64_t>(strips) - 1u) * 2u * (static_cast<uint64_t>(la) + 1u) + 3u) & ~3ull;
}
__host__ __device__ inline uint64_t ck_rec_floats(uint32_t strips) { return strips > 1u ? static_cast<uint64_t>(strips) * (kCkRecBytes / 4u) : 0u; }
__host__ __device__ inline uint64_t strip_dwords(uint32_t la, uint32_t w = kW) {
    if(w == 3u) return static_cast<uint64_t>((la + kWave - 1 + 15u) / 16u) * kLp3GroupDwords;
    const uint32_t mc = 32u / w;
    return static_cast<uint64_t>((la + kWave - 1 + mc - 1) / mc) * kPairDwords;
}
// viterbi_ck.hip (gap_len 1) keeps no per-cell bits: per strip of W columns per lane it stores
//   colin  float2[c][lane][k%16] what lane `lane` received from its left neighbour at wavefront step k = c*kCkRows + k%16
//                               (diagonal X, left Z); k < la + 63.  TILE-major (round 5): the 16 values of a (band, lane)
//                               tile are one 128-byte line, which is what a recompute reads (step-major, rounds 2-4: 8 bytes
//                               out of each of 16 512-byte rows -- 1.45 GB fetched per 10 000-pair launch for 0.5 GB used).
//                               The row parts of a CUT pair keep the step-major float2[k][lane]: their stores go through
//                               the L2 one by one, and a step's lanes are neighbours there
//   rowck  float4[c][lane][q]   the lane state before step c*kCkRows: X[0..W) then Y[0..W) as W/2
//                               float4 (q), one band c per kCkRows steps (tile-major too; cut pairs [c][q][lane])
// from which any (band, lane) tile can be recomputed on its own.  8/W + 8/kCkRows bytes per cell.
constexpr uint32_t kCkRowsLog2 = 4, kCkRows = 1u << kCkRowsLog2;
// A single-strip pair needs its checkpoints only until its own wavefront has walked it, so a large
// batch does not reserve them per pair: PairDesc::flags_off == kCkWaveSlot means "the slot of the
// wavefront that processes the pair" (BatchDeviceView::ck_slot_dwords dwords per resident wavefront at
// the start of the arena).  Pairs of several strips (other wavefronts read their checkpoints) and the
// pairs of small batches keep per-pair storage.
constexpr uint64_t kCkWaveSlot = ~0ull;
constexpr uint32_t kCkBandOff = 0xffffffffu;  // banded checkpoints (viterbi_ck.hip): no band, every tile keeps its checkpoints
__host__ __device__ constexpr uint32_t ck_rowck_quads(uint32_t w) { return w / 2; }
__host__ __device__ inline uint32_t ck_bands(uint32_t la) { return (la + kWave + kCkRows - 1) / kCkRows; }
__host__ __device__ inline uint64_t ck_colin_dwords(uint32_t la) { return static_cast<uint64_t>(ck_bands(la)) * (kCkRows * 2 * kWave); }
__host__ __device__ inline uint64_t ck_strip_dwords(uint32_t la, uint32_t w) {
    return ck_colin_dwords(la) + static_cast<uint64_t>(ck_bands(la)) * (2 * w * kWave);
}
// Compact layout (gap_len L = 2, 3; viterbi_k.hip): live body cell (bi, bj), bi = p*L + r,
// bj = q*L + r.  A lane owns W <= 16 block columns q; wavefront step k = p + lane.  Per strip and
// step S = 2L + (L+1)/2 rows of 64 dwords:
//   row r          A bits of phase r: column c at bits 31-2c, 30-2c
//   row L + r      B bits of phase r, same positions
//   row 2L + h     C bits of phases 2h and 2h+1: column c at bit 31-2c (phase 2h) and 30-2c
//                  (phase 2h+1); if phase 2h+1 does not exist (L odd) column c is bit 31-c
__host__ __device__ constexpr uint32_t compact_slots(uint32_t L) { return 2 * L + (L + 1) / 2; }
__host__ __device__ inline uint64_t compact_strip_dwords(uint32_t la, uint32_t L) {
    return static_cast<uint64_t>(la / L + kWave - 1) * compact_slots(L) * kWave;
}
__host__ __device__ inline uint32_t n_strips(uint32_t lb) { return (lb + kStrip - 1) / kStrip; }
// Viterbi strip plan of a descendant of lb columns with w_main columns per lane in every strip
// but the last: the last strip takes the narrowest shape (4, 8, 16 columns per lane, at most
// w_main) that holds the remainder.  (w_main = 2: viterbi_lp, a few long pairs.)
inline void viterbi_strip_plan(uint32_t lb, uint32_t w_main, uint32_t& strips, uint32_t& w_last) {
    const uint32_t full = kWave * w_main;
    if(w_main < 4) {  // (viterbi_lp's 2-column strips: one shape throughout)
        strips = (lb + full - 1) / full;
        w_last = w_main;
        return;
    }
    const uint32_t whole = lb / full, rem = lb % full;
    if(rem == 0 && whole > 0) {
        strips = whole;
        w_last = w_main;
        return;
    }
    strips = whole + 1;
    w_last = 4;
    while(w_last < w_main && kWave * w_last < rem) w_last *= 2;
}

// lane l receives lane l-1's `v`; lane 0 receives `lane0` (DPP keeps `old` where
// the shift has no source lane).
__device__ __forceinline__ float shift_in(float v, float lane0) {
    const int r = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, lane0), __builtin_bit_cast(int, v),
                                              0x138 /*wave_shr:1*/, 0xf, 0xf, false);
    return __builtin_bit_cast(float, r);
}
__device__ __forceinline__ uint32_t shift_in(uint32_t v, uint32_t lane0) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(lane0), static_cast<int>(v),
                                                             0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float read_lane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ uint32_t read_lane(uint32_t v, int lane) {
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), lane));
}

// ---------------------------------------------------------------------------
// decision-bit lookups (layout above); (bi, bj) are BODY coordinates
// ---------------------------------------------------------------------------
struct CellAddr {
    uint64_t idx_a, idx_b, idx_c;  // dword indices of the cell's A, B and C words
    uint32_t sh_ab, sh_c;          // shifts that bring the cell's bits to the bottom
};
__device__ __forceinline__ CellAddr cell_addr(const PairDesc& pd, uint32_t bi, uint32_t bj) {
    if(pd.v_compact != 0) {  // viterbi_k: live cells only
        const uint32_t L = pd.v_compact;
        const uint32_t q = bj / L, r = bi % L, p = bi / L;
        const uint32_t full = kWave * pd.v_wmain;
        uint32_t strip = q / full, w = pd.v_wmain;
        if(strip + 1 >= pd.v_strips) {
            strip = pd.v_strips - 1;
            w = pd.v_wlast;
        }
        const uint32_t colin = q - strip * full, t = colin / w, c = colin % w;
        const uint64_t base = pd.flags_off + strip * compact_strip_dwords(pd.la, L) +
                              static_cast<uint64_t>(p + t) * (compact_slots(L) * kWave) + t;
        const bool paired = (r | 1u) < L;  // phase r shares its C word with a partner phase
        return {base + r * kWave, base + (L + r) * kWave, base + (2 * L + r / 2) * kWave, 30u - 2u * c,
                paired ? 31u - 2u * c - (r & 1u) : 31u - c};
    }
    const uint32_t full = kWave * pd.v_wmain;
    uint32_t strip = bj / full, w = pd.v_wmain;
    if(pd.v_wmain == 3u) {  // per-column words (layout above); every strip of the pair has this shape
        const uint32_t colin = bj - strip * full, t = colin / 3u, c = colin - 3u * t;
        const uint32_t kstep = bi + t, g = kstep >> 4, tt = kstep & 15u;
        const uint64_t base = pd.flags_off + strip * strip_dwords(pd.la, 3u) + static_cast<uint64_t>(g) * kLp3GroupDwords;
        return {base + 3u * t + c, base + 3u * kWave + 3u * t + c, base + 6u * kWave + 2u * t + (c >> 1),
                30u - 2u * tt, ((c & 1u) << 4) + 15u - tt};
    }
    if(strip + 1 >= pd.v_strips) {
        strip = pd.v_strips - 1;
        w = pd.v_wlast;
    }
    const uint32_t lg = 31u - static_cast<uint32_t>(__clz(static_cast<int>(w)));  // w = 4, 8, 16
    const uint32_t colin = bj - strip * full;
    const uint32_t t = colin >> lg, c = colin & (w - 1u);
    const uint32_t kstep = bi + t;
    const uint32_t lg_mc = 5u - lg, lg_ma = 4u - lg;          // steps per C dword / per A,B dword
    const uint32_t g = kstep >> lg_mc, q = kstep & ((1u << lg_mc) - 1u);
    const uint32_t half = q >> lg_ma, tt = q & ((1u << lg_ma) - 1u);
    const uint64_t base = pd.flags_off + strip * strip_dwords(pd.la, pd.v_wmain) + static_cast<uint64_t>(g) * kPairDwords + t;
    return {base + half * (2u * kWave), base + half * (2u * kWave) + kWave, base + 4u * kWave,
            30u - 2u * ((tt << lg) + c), 31u - ((q << lg) + c)};
}
// two-bit decision of accumulator A (which = 0) or B (which = 1)
__device__ __forceinline__ uint32_t pair_bits(const uint32_t* __restrict__ flags, const CellAddr& ca, int which) {
    const uint32_t w = flags[which ? ca.idx_b : ca.idx_a];
    return (w >> ca.sh_ab) & 3u;  // bit1 = M argument is not the maximum, bit0 = D argument is not
}
__device__ __forceinline__ uint32_t im_bit(const uint32_t* __restrict__ flags, const CellAddr& ca) {
    return (flags[ca.idx_c] >> ca.sh_c) & 1u;
}
// state entered after a move of kind `moved` arrives at body cell (bi, bj)
__device__ __forceinline__ int state_after(const uint32_t* __restrict__ flags, const PairDesc& pd, uint32_t bi,
                                           uint32_t bj, int moved) {
    const CellAddr ca = cell_addr(pd, bi, bj);
    if(moved == COATI_HIP_OP_INS) return im_bit(flags, ca) ? COATI_HIP_OP_MATCH : COATI_HIP_OP_INS;
    const uint32_t two = pair_bits(flags, ca, moved == COATI_HIP_OP_DEL ? 1 : 0);
    if(!(two & 2u)) return COATI_HIP_OP_MATCH;  // the M argument is the maximum (ties: M first)
    return (two & 1u) ? COATI_HIP_OP_INS : COATI_HIP_OP_DEL;
}

// ---------------------------------------------------------------------------
// margins (align_pair.cc:82-91): values of matrix cell (i, j) with i < L or j < L
// ---------------------------------------------------------------------------
__device__ __forceinline__ void margin_mdi(const GapConsts& k, uint32_t L, uint32_t i, uint32_t j, float& m,
                                           float& d, float& in) {
    const uint32_t start = L - 1;
    m = d = in = kLowest;
    if(i == start && j == start) m = 0.0f;
    if(j == start && i > start && (i - start) % L == 0) d = (k.ng + k.go) + k.ge * static_cast<float>(i - 1);
    if(i == start && j > start && (j - start) % L == 0) in = k.go + k.ge * static_cast<float>(j - 1);
}
// max_mdi (align_pair.cc:210-224): ties M over D over I
__device__ __forceinline__ int max_mdi(float cm, float cd, float ci) {
    int w = COATI_HIP_OP_MATCH;
    float best = cm;
    if(cd > best) {
        best = cd;
        w = COATI_HIP_OP_DEL;
    }
    return ci > best ? COATI_HIP_OP_INS : w;
}
// the three "which state next" rules of align_pair.cc:275-296 on a cell's M/D/I
__device__ __forceinline__ int decide_after(const GapConsts& k, int moved, float m, float d, float in) {
    if(moved == COATI_HIP_OP_INS) return (m + k.go) > (in + k.ge) ? COATI_HIP_OP_MATCH : COATI_HIP_OP_INS;
    if(moved == COATI_HIP_OP_MATCH) return max_mdi((m + k.ng) + k.ng, d + k.gs, (in + k.gs) + k.ng);
    return max_mdi((m + k.ng) + k.go, d + k.ge, (in + k.gs) + k.go);
}

constexpr int kWalkEnd = 3;
// State the reference's walk is in after arriving at MATRIX cell (i, j) by a move of
// kind `moved`: body cells from the stored bits, margin cells by formula.
__device__ __forceinline__ int arrival_state(const GapConsts& k, uint32_t L, const uint32_t* __restrict__ flags,
                                             const PairDesc& pd, uint32_t i, uint32_t j, int moved) {
    if(i < L && j < L) return kWalkEnd;  // loop condition of align_pair.cc:268
    if(i >= L && j >= L) return state_after(flags, pd, i - L, j - L, moved);
    float m, d, in;
    margin_mdi(k, L, i, j, m, d, in);
    return decide_after(k, moved, m, d, in);
}

// traceback<tropical> (align_pair.cc:249-303) by one WAVEFRONT.
// A walk is a chain of dependent loads, but it consists of long runs of the same
// move.  So the 64 lanes speculate: lane l looks up the state the walk would be
// in after l+1 further moves of the current kind; a ballot finds the first lane
// where the run ends; all moves up to there are emitted at once (coalesced
// byte stores) and the walk jumps.  Memory round trips per pair drop from
// len_a+len_b to about (number of runs + length/64).
// Ops (one byte per alignment column; a gap move emits L of them) are written
// right-to-left into the pair's slot so they end up in alignment order.  All
// lanes must call this (it uses ballots).  `start_state` is max_mdi of the
// terminal-adjusted last cell.
__device__ __forceinline__ void walk_pair(int lane, const GapConsts& k, uint32_t L, const PairDesc& pd,
                                          uint32_t pair, int start_state, const uint32_t* __restrict__ flags,
                                          uint8_t* __restrict__ ops, uint64_t* __restrict__ ops_start,
                                          uint32_t* __restrict__ ops_len) {
    const uint32_t la = pd.la, lb = pd.lb;
    uint32_t i = la + L - 1, j = lb + L - 1;  // matrix coordinates of the last cell
    uint64_t pos = pd.ops_off + la + lb;
    int st = (i < L && j < L) ? kWalkEnd : start_state;
    while(st != kWalkEnd) {
        const uint32_t di = st == COATI_HIP_OP_MATCH ? 1u : (st == COATI_HIP_OP_DEL ? L : 0u);
        const uint32_t dj = st == COATI_HIP_OP_MATCH ? 1u : (st == COATI_HIP_OP_INS ? L : 0u);
        const uint32_t width = st == COATI_HIP_OP_MATCH ? 1u : L;  // alignment columns per move
        // lane l: where the walk is after l+1 more moves of kind st, and in which state
        const uint32_t step = static_cast<uint32_t>(lane) + 1u;
        const bool valid = di * step <= i && dj * step <= j;
        int next = kWalkEnd;
        if(valid) next = arrival_state(k, L, flags, pd, i - di * step, j - dj * step, st);
        if(di > i || dj > j) break;  // cannot happen for decision bits of a finite path; never walk off the matrix
        const unsigned long long cont = __builtin_amdgcn_ballot_w64(valid && next == st);
        const uint32_t run = cont == ~0ull ? kWave : static_cast<uint32_t>(__builtin_ctzll(~cont));  // lanes that continue
        const uint32_t moves = run == kWave ? kWave : run + 1u;
        for(uint32_t q = lane; q < moves * width; q += kWave) ops[pos - 1 - q] = static_cast<uint8_t>(st);
        pos -= static_cast<uint64_t>(moves) * width;
        i -= di * moves;
        j -= dj * moves;
        if(run < kWave) st = __builtin_amdgcn_readlane(next, static_cast<int>(run));
    }
    if(lane == 0) {
        ops_start[pair] = pos;
        ops_len[pair] = static_cast<uint32_t>(pd.ops_off + la + lb - pos);
    }
}

// Terminal adjustment (align_pair.cc:130-138) + score + max_mdi of a last cell with
// UNADJUSTED values (m, d, in).
__device__ __forceinline__ int terminal_state(const GapConsts& k, float m, float d, float in, float& score) {
    const float tm = (m + k.ng) + k.ng, td = d + k.gs, ti = (in + k.gs) + k.ng;
    score = fmaxf(fmaxf(tm, td), ti);
    return max_mdi(tm, td, ti);
}

// Viterbi epilogue of one pair (all lanes): start state + score where the fill did
// not produce them (no body cells), then the walk.
__device__ __forceinline__ void viterbi_finish(int lane, const GapConsts& k, uint32_t L, const PairDesc& pd,
                                               uint32_t pair, const uint32_t* __restrict__ flags,
                                               uint8_t* __restrict__ ops, uint64_t* __restrict__ ops_start,
                                               uint32_t* __restrict__ ops_len, float* __restrict__ scores) {
    int start_state;
    if(pd.la > 0 && pd.lb > 0) {
        // max_mdi of the terminal-adjusted last cell == its "after match" decision
        start_state = __builtin_amdgcn_readfirstlane(
            state_after(flags, pd, pd.la - 1, pd.lb - 1, COATI_HIP_OP_MATCH));
    } else {
        float m, d, in, score;
        margin_mdi(k, L, pd.la + L - 1, pd.lb + L - 1, m, d, in);
        start_state = terminal_state(k, m, d, in, score);
        if(lane == 0) scores[pair] = score;
    }
    walk_pair(lane, k, L, pd, pair, start_state, flags, ops, ops_start, ops_len);
}

// Forward: HBM layout of the fp32 M/D/I of the body cells of one strip of one pair:
//   float[(((k * W + c) * 64 + lane) * 3 + mat], W = 4, 8 or 16 columns per lane, k = wavefront step = body_row + lane,
//   mat 0/1/2 = M/D/I, c = the lane's column.  Every (k, c) is one coalesced 768-byte row
//   written by one 12-byte store per lane.  12 bytes per cell.
constexpr int kMdiStepFloats = 3 * kW * kWave;  // 3072
__host__ __device__ inline uint64_t strip_mdi_floats(uint32_t la) {
    return static_cast<uint64_t>(la + kWave) * kMdiStepFloats;
}
// the same for strips of w columns per lane (forward_l1 narrows its strips when a batch has few pairs)
__host__ __device__ inline uint64_t strip_mdi_floats_w(uint32_t la, uint32_t w) {
    return static_cast<uint64_t>(la + kWave) * (3 * w * kWave);
}
__host__ __device__ inline uint32_t fwd_strips_w(uint32_t lb, uint32_t w) { return (lb + kWave * w - 1) / (kWave * w); }
// Quad strips (a handful of pairs; forward_l1.hip: forward_quad_strip): four lanes per column -- one each for the M, the D and
// the I sum of a cell --, kFwdQuadCols columns per wavefront.  The M/D/I LAYOUT stays that of 1-column strips (f_wlog2 = 0):
// only the work items, their boundary arrays and their progress words are per quad strip.
constexpr uint32_t kFwdQuadCols = kWave / 4;
__host__ __device__ inline uint32_t fwd_quad_strips(uint32_t lb) { return (lb + kFwdQuadCols - 1) / kFwdQuadCols; }
// Compact Forward layout (gap_len L = 2, 3; forward_k.hip): live cells (p*L + r, q*L + r) only.  A
// lane owns Wf block columns (one shape per L), step k = p + lane:
//   float[((((k * L + r) * Wf + c) * 64 + lane) * 3 + mat]
__host__ __device__ constexpr uint32_t fwd_compact_w(uint32_t L) { return L == 3 ? 6u : 8u; }
__host__ __device__ inline uint64_t fwd_compact_strip_floats(uint32_t la, uint32_t L) {
    return static_cast<uint64_t>(la / L + kWave - 1) * (3 * L * fwd_compact_w(L)) * kWave;
}
__host__ __device__ inline uint32_t fwd_compact_strips(uint32_t lb, uint32_t L) {
    const uint32_t per = kWave * fwd_compact_w(L);
    return (lb / L + per - 1) / per;
}
// M, D, I of one cell are adjacent (12 bytes): the fill writes them with one 12-byte store per lane (a
// wavefront's store covers 768 contiguous bytes) and a sampler step reads them with one load.
struct Mdi {
    float m, d, in;
};
__device__ __forceinline__ uint64_t mdi_index(const PairDesc& pd, uint32_t bi, uint32_t bj, int mat) {
    if(pd.f_compact != 0) {
        const uint32_t L = pd.f_compact, wf = fwd_compact_w(L), per = kWave * wf;
        const uint32_t q = bj / L, r = bi % L, p = bi / L;
        const uint32_t strip = q / per, colin = q % per, t = colin / wf, c = colin % wf;
        return pd.mdi_off + strip * fwd_compact_strip_floats(pd.la, L) +
               (((static_cast<uint64_t>(p + t) * L + r) * wf + c) * kWave + t) * 3 + mat;
    }
    const uint32_t wl = pd.f_wlog2, w = 1u << wl, colin = bj & (kWave * w - 1);
    const uint32_t strip = bj >> (6 + wl), t = colin >> wl, c = bj & (w - 1);
    return pd.mdi_off + strip * strip_mdi_floats_w(pd.la, w) + ((static_cast<uint64_t>(bi + t) * w + c) * kWave + t) * 3 + mat;
}
// a live cell of the Forward layout?  (every cell when the layout is not compact)
__device__ __forceinline__ bool mdi_stored(const PairDesc& pd, uint32_t bi, uint32_t bj) {
    return pd.f_compact == 0 || (bi % pd.f_compact) == (bj % pd.f_compact);
}

// log-semiring plus = log_sum_exp (semiring.hpp:86-121, utils.hpp:134-156), BIT-EXACT: glibc's expf and
// log1pf restated in glibc_math.hpp, the reference's branch at y <= -16 included.  `exp_tab` is the
// 32-entry table of expf (kernels keep a copy in LDS, see load_exp_table).
__device__ __forceinline__ float log_plus_exact(float a, float b, const uint64_t* exp_tab) {
    const float hi = fmaxf(a, b);
    const float y = -fabsf(a - b);
    const float e = libm::expf_nonpos(y, exp_tab);
    // (e >= expf(-16) = 1.1e-7 on the log1pf side.)  A wavefront whose lanes all either skip log1pf (y <= -16) or take its
    // k = 0 route (e < 0.41422) runs that route alone -- the same operations on the same values, so the same bits; the
    // reference's own early-out at y <= -16 (utils.hpp:141) is almost never wave-uniform (< 1 % of the instructions:
    // profiles/r04/forward_y_histogram.txt), this one is in 25-40 %.
    const bool small = y <= -16.0f || e < libm::u2f(0x3ed413d7u);
    float l;
    if(__builtin_amdgcn_ballot_w64(small) == __builtin_amdgcn_ballot_w64(true))
        l = libm::log1pf_small(e);
    else
        l = libm::log1pf_mid(e);
    return hi + (y <= -16.0f ? e : l);
}
// Two independent plus operations side by side (round 5): plus(a0, b0) and plus(a1, b1), each exactly as log_plus_exact
// computes it -- the two dependent chains (fp64 expf, reciprocal + Newton steps, the degree-7 polynomial) in ONE basic block,
// so that the scheduler can overlap them.  The wave-uniform choice of log1pf's k = 0 route is made for both at once.
__device__ __forceinline__ void log_plus_exact_x2(float a0, float b0, float a1, float b1, const uint64_t* exp_tab, float& out0, float& out1) {
    const float hi0 = fmaxf(a0, b0), hi1 = fmaxf(a1, b1);
    const float y0 = -fabsf(a0 - b0), y1 = -fabsf(a1 - b1);
    const float e0 = libm::expf_nonpos(y0, exp_tab), e1 = libm::expf_nonpos(y1, exp_tab);
    const bool small = (y0 <= -16.0f || e0 < libm::u2f(0x3ed413d7u)) && (y1 <= -16.0f || e1 < libm::u2f(0x3ed413d7u));
    float l0, l1;
    if(__builtin_amdgcn_ballot_w64(small) == __builtin_amdgcn_ballot_w64(true)) {
        l0 = libm::log1pf_small(e0);
        l1 = libm::log1pf_small(e1);
    } else {
        libm::log1pf_mid_x2(e0, e1, l0, l1);
    }
    out0 = hi0 + (y0 <= -16.0f ? e0 : l0);
    out1 = hi1 + (y1 <= -16.0f ? e1 : l1);
}
__device__ __forceinline__ void load_exp_table(uint64_t* lds_tab, int tid) {
    constexpr uint64_t kTab[32] = {COATI_EXP2F_TABLE};
    if(tid < 32) lds_tab[tid] = kTab[tid];
}

// The same with the hardware exp2/log2 instead (opt-in, COATI_HIP_FORWARD_FAST=1: ~3x faster, not
// bit-identical to the CPU); see forward_l1.hip for the derivation and the error bound.
__device__ __forceinline__ float log_plus(float a, float b) {
    constexpr float kLog2e = 1.44269504088896340736f, kLn2 = 0.69314718055994530942f;
    const float hi = fmaxf(a, b);
    const float t = -fabsf(a - b) * kLog2e;        // <= 0 (or -inf); abs/neg are source modifiers
    const float e = __builtin_amdgcn_exp2f(t);     // v_exp_f32: exp(-|a-b|) in [0, 1]
    const float u = 1.0f + e;
    const float resid = e - (u - 1.0f);            // exact
    const float l2 = __builtin_amdgcn_logf(u);     // v_log_f32: log2(u) in [0, 1]
    // (gfx950's v_log_f32 is biased low by up to 2.9e-8 on [1.4, 2], v_exp_f32 is centred: tools/ubench/trans_err.hip ->
    // profiles/r03/ubench_transcendental_error.txt.  Putting the measured bias back in -- three more instructions --
    // changed NO figure of the fuzz campaign (tools/fuzz_sample.py, 900 000 samples: the same worst deviations to
    // the digit): what separates this build from the CPU is not the accuracy of log1p(exp()) but the 1e-8 noise of the
    // CPU's own libm against the rounding grid of values of magnitude ~100 -- DESIGN.md 3.2b.  Left out.)
    return hi + __builtin_fmaf(l2, kLn2, resid);
}

// The tolerance build's three-way sum, plus(plus(a, b), c) of align_pair.cc:97-119 as ONE log-sum-exp (round 6): the largest
// of the three contributes exp(0) = 1, so two exponentials and one logarithm where the nested form takes four transcendental
// instructions -- and those, at a quarter of the vector rate, are what the tolerance build's cell consists of (10 of them in
// ~70 instructions; 8 now).  Mathematically the same sum with one rounding fewer; not the reference's bits (nor is log_plus).
__device__ __forceinline__ float log_plus3(float a, float b, float c) {
    constexpr float kLog2e = 1.44269504088896340736f, kLn2 = 0.69314718055994530942f;
    const float hi = __builtin_fmaxf(__builtin_fmaxf(a, b), c);                      // v_max3_f32
    const float lo = __builtin_fminf(__builtin_fminf(a, b), c);                      // v_min3_f32
    const float mid = __builtin_amdgcn_fmed3f(a, b, c);                              // v_med3_f32
    const float e1 = __builtin_amdgcn_exp2f((mid - hi) * kLog2e);                    // in [0, 1]
    const float e2 = __builtin_amdgcn_exp2f((lo - hi) * kLog2e);                     // in [0, e1]
    const float t = e1 + e2;                                                         // in [0, 2]
    const float u = 1.0f + t;
    const float resid = t - (u - 1.0f);  // the rounding residue of 1 + t (exact up to t = 1, within an ulp of u above)
    return hi + __builtin_fmaf(__builtin_amdgcn_logf(u), kLn2, resid);
}

// ---------------------------------------------------------------------------
// The COATI_HIP_* environment, read ONCE per process (first use) into one struct -- not on every batch_create or launch:
// the one-pair call path is 0.27 ms.  These are A/B, test and experiment switches; what an embedder needs is a model
// option (coati_hip_model_set_option: persistent-call form, checkpoint band).  coati_hip_debug_reload_env() reads the
// environment again (the Python test plumbing calls it before every entry, so tests can switch kernels in one process).
struct EnvOptions {
    bool force_generic = false;      // COATI_HIP_FORCE_GENERIC: dp_generic for every gap_len (second implementation)
    bool viterbi_bits = false;       // COATI_HIP_VITERBI_BITS: viterbi_l1 / viterbi_lp instead of viterbi_ck
    bool viterbi_ck = false;         // COATI_HIP_VITERBI_CK: viterbi_ck whatever the planner would choose
    bool l1_lp_off = false;          // COATI_HIP_L1_LP=0: viterbi_l1 where viterbi_lp would run
    bool l1_progress = false;        // COATI_HIP_L1_PROGRESS: progress-word boundary protocol in viterbi_l1
    bool ck_per_pair = false;        // COATI_HIP_CK_PER_PAIR: no per-wavefront checkpoint slots
    int stream_tail_units_x10 = 0;   // COATI_HIP_STREAM_TAIL_UNITS (x 10): size of the streamed call's last chunk in units of 10^9 cells (0: 2.6)
    int stream_parts = -1;           // COATI_HIP_STREAM_PARTS: -1 (default) = ONE large last chunk of a streamed call cut into 3 row parts, 22 .. 28 = into 2 .. 8,
                                     // 0 = no row parts, 1 = row parts in the last ~1 000-pair chunks (round 3's form)
    int stream_helpers = 55;         // COATI_HIP_STREAM_HELPERS (A/B): 1 = chunks planned ahead, 2 = results unstaged, 4 = slots allocated on the model's helper threads,
                                     // 8 = a trace line per call, 16 / 32 = the kernel stores results straight into the caller's page-locked arrays / the slot's staging block
    bool pipe_no_d2h = false;        // COATI_HIP_PIPE_NO_D2H: chunk pipeline without downloads (timing experiment)
    bool sample_sequential = false;  // COATI_HIP_SAMPLE_SEQUENTIAL: one serial walker per pair
    bool sample_table_off = false;   // COATI_HIP_SAMPLE_TABLE=0: exact-stream sampler without the step table
    bool fwd_wide_build = false;     // COATI_HIP_FWD_WIDE_BUILD: forward_l1's 16-column build for narrow strips too
    bool lp_pairtab_off = false;     // COATI_HIP_LP_PAIRTAB=0
    bool lp3_off = false;            // COATI_HIP_LP3=0: the planner never chooses 3-column strips (A/B: the round-5 plan)
    int lp_splice = -1;              // COATI_HIP_LP_SPLICE: 0 = viterbi_lp walks every strip itself, 1 = spliced traceback, 2 ("miss") = records that never match (tests); -1: the default (on for multi-strip pairs)
    int ck_splice = -1;              // COATI_HIP_CK_SPLICE: the same for viterbi_ck's multi-strip pairs (resident launches): 0 = the last strip's wavefront walks the pair alone, 1 = spliced, 2 ("miss") = records that never match, 3 ("nobridge") = records without bridges; -1: on
    bool forward_fast = false;       // COATI_HIP_FORWARD_FAST: hardware exp / log in the log-semiring plus (not a parity mode)
    bool timing = false;             // COATI_HIP_TIMING: host stage times on stderr
    bool pipe_timing = false;        // COATI_HIP_PIPE_TIMING: timeline of a one-shot call on stderr
    bool sdma_off = false;           // HSA_ENABLE_SDMA=0: copies are kernels -- no persistent-kernel call form
    int pipe = 0;                    // COATI_HIP_PIPE: 0 = by the input, 1 = "chunks", 2 = "stream"
    int strip_w = 0;                 // COATI_HIP_STRIP_W: 2 / 4 / 8 / 16 columns per lane in every strip but the last
    int fwd_w = 0;                   // COATI_HIP_FWD_W: 1 / 2 / 4 / 8 / 16
    int fwd_quad = -1;               // COATI_HIP_FWD_QUAD: 0 / 1 force the quad strips of forward_l1 off / on (where the plan is 1 column per lane)
    int fill_blocks_per_cu = 0;      // COATI_HIP_FILL_BLOCKS_PER_CU
    int lp_blocks_per_cu = 0;        // COATI_HIP_LP_BLOCKS_PER_CU
    long long tail_pairs = -1;       // COATI_HIP_TAIL_PAIRS (-1: the planner's rule)
    uint32_t ck_band = 64;           // COATI_HIP_CK_BAND: default of COATI_HIP_OPT_CK_BAND (0 -> kCkBandOff)
    uint32_t ck_debug = 0;           // COATI_HIP_CK_DEBUG: bit 0 fill only, bit 1 traceback statistics
    bool ck_split_set = false;       // COATI_HIP_CK_SPLIT="pairs,parts[,t]"
    uint64_t ck_split_pairs = 0, ck_split_parts = 3;
    bool ck_split_taper = false;
    int ck_fuse = -1;                  // COATI_HIP_CK_FUSE=0 / 1: a two-strip pair whose second strip is narrow as TWO items / both strips by the wavefront that draws the first (-1: the planner's rule)
    int ck_walk_items = -1;            // COATI_HIP_CK_WALK_ITEMS=0 / 1: the traceback of a pair cut into row parts stays with its last part / is an item of its own (-1: the planner's rule)
    uint32_t ck_split_short_last = 0;  // COATI_HIP_CK_SPLIT="pairs,parts,s<k>": the last part k chunks shorter (0-7)
    uint32_t spec_cands = 3u << 16;  // COATI_HIP_SPEC_CANDS (196 608: 16 x 1 000 samples 6.1 ms; 2^17: 6.4, 2^18: 6.4, 2^16: 7.8 -- tools/sample_bench.py, round 4)
    double spec_z = 2.0;             // COATI_HIP_SPEC_Z
    uint32_t sample_band = 64;       // COATI_HIP_SAMPLE_BAND: half width, in diagonals, of the sampler's step table (tests: 1 or 2 force the walkers' own entries)
    bool spec_host_rounds = false;   // COATI_HIP_SPEC_HOST_ROUNDS: the sampler's speculation rounds planned and resolved on the host (round 3's loop; A/B, tests)
    long double stream_unit = 0;     // COATI_HIP_STREAM_UNIT (cells; 0: the default)
    uint64_t mem_budget = 0;         // COATI_HIP_MEM_BUDGET (bytes; 0: none)
};
const EnvOptions& env_options();

// ---------------------------------------------------------------------------
// launchers implemented in the kernel translation units
// ---------------------------------------------------------------------------
// One unit of work of the persistent kernels: one strip of one pair.  The strips of a pair are
// consecutive items in ascending order.
struct WorkItem {
    uint32_t pair, strip;
};
// viterbi_ck, round 6 -- FUSED two-strip pairs.  A pair a little wider than one strip (the bench's synthetic set: 333 of 10 000
// descendants are 1 025-1 082 nt long) is a full strip and one of a few columns; as two items on two wavefronts the second spends
// the first's whole time following it -- a wavefront slot that mostly polls --, and the pair pays a release and two acquires.  In a
// launch with (many) more items than wavefronts the planner marks the two items (upper half of `strip`, which only pairs cut into
// row parts use otherwise): the wavefront that draws the first fills BOTH strips, one after the other, and walks the pair;
// whoever draws the second moves on.  Same values, same order of operations per cell; nothing waits for another wavefront.
constexpr uint32_t kCkFusedFirst = 0xfffeu, kCkFusedSecond = 0xffffu;

struct BatchDeviceView {
    const float* table;
    GapConsts k;
    uint32_t gap_len;
    const PairDesc* pairs;
    const uint32_t* order;
    uint32_t n_pairs;
    uint32_t* queue;
    const WorkItem* items;  // viterbi_l1: (pair, strip) work list of the Viterbi strip plan, longest pairs first
    uint32_t n_items;
    const WorkItem* fwd_items;  // forward_l1: the same for fixed 1024-column strips
    uint32_t n_fwd_items;
    uint32_t* progress;     // rows of each item whose boundary column is published
    const uint8_t *a_cat, *b_cat;
    uint32_t* flags;
    float* bnd;
    uint64_t bnd_bytes;
    float* scores;
    uint8_t* ops;
    uint64_t* ops_start;
    uint32_t* ops_len;
    uint32_t* wscratch;  // viterbi_ck: per-wavefront scratch of the traceback (decision bits of one round)
    uint64_t ck_slot_dwords;  // viterbi_ck: size of one per-wavefront checkpoint slot (0: every pair has its own storage)
    uint32_t ck_split_items;  // viterbi_ck: row part p > 0 of a cut pair waits for the item this many tickets before its own
    float* mdi;        // Forward: fp32 M/D/I of every body cell
    float* final_mdi;  // Forward: terminal-adjusted M, D, I of the last cell, 3 floats per pair
    uint32_t fwd_wlog2_max;  // forward_l1: the widest strip shape of the batch (log2 of the columns per lane)
    uint32_t ck_band;        // viterbi_ck: half width of the kept checkpoint band in wavefront steps, kCkBandOff = keep everything
    uint32_t long_pairs;     // decision-bit plan of a few long pairs (every strip 4 columns per lane): viterbi_lp fills it
    uint32_t multi_strip;    // the Viterbi plan has pairs of several strips (their boundary arrays start every launch as NaN patterns)
    uint32_t fwd_quad;       // forward_l1: the items are QUAD strips (kFwdQuadCols columns, four lanes per column: forward_l1.hip)
    uint32_t fwd_fast;       // Forward kernels: hardware exp2 / log2 (COATI_HIP_FORWARD_TOLERANCE) instead of the libm restatements
};
// Before every launch of a persistent kernel: the ticket counter and the polled progress words start at zero.  The
// planner lays the counter out right in front of the words (plan.hip), so this is ONE fill, not two.
inline hipError_t zero_queue_and_progress(const BatchDeviceView& v, uint32_t n_words, hipStream_t stream) {
    const size_t words = n_words < 4u ? 4u : n_words;
    char* q = reinterpret_cast<char*>(v.queue);
    char* p = reinterpret_cast<char*>(v.progress);
    if(p > q && p - q <= 4096) return hipMemsetAsync(q, 0, static_cast<size_t>(p - q) + sizeof(uint32_t) * words, stream);
    const hipError_t e = hipMemsetAsync(v.queue, 0, sizeof(uint32_t), stream);
    return e != hipSuccess ? e : hipMemsetAsync(v.progress, 0, sizeof(uint32_t) * words, stream);
}
hipError_t launch_viterbi_l1(const BatchDeviceView& v, hipStream_t stream);
// viterbi_lp.hip: the same fill (same decision-bit layout, same strip pipeline) issued for wavefronts that are alone
// on their SIMDs; 4-column strips only
hipError_t launch_viterbi_lp(const BatchDeviceView& v, hipStream_t stream);
// Compute units of the current device (hipDeviceAttributeMultiprocessorCount), at most 256: the persistent
// kernels size their grids to what is RESIDENT (a partitioned or CU-masked device has fewer than the MI355X's
// 256; a grid sized for 256 would oversubscribe it and the residency the queue discipline assumes would not
// hold), and the per-wavefront scratch areas are sized for 256 CUs.
inline uint32_t device_cu_count() {
    int dev = 0, n = 0;
    if(hipGetDevice(&dev) != hipSuccess) return 256u;
    if(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256u;
    return static_cast<uint32_t>(n < 256 ? n : 256);
}

// viterbi_ck.hip: lean fill + checkpoint traceback (`flags` is the checkpoint arena); shared_tab: the
// model has one substitution table.  The scratch is ck_scratch_waves() x ck_scratch_dwords_per_wave().
hipError_t launch_viterbi_ck(const BatchDeviceView& v, bool shared_tab, hipStream_t stream);
uint32_t ck_scratch_waves();
uint64_t ck_scratch_dwords_per_wave();
// viterbi_ck_stream (viterbi_ck.hip): one persistent launch fed chunk by chunk; the control block lives in HBM
constexpr int kCkStreamSlots = 12;
void ck_stream_host_set_slots(void* host, uint32_t n_slots);  // before the launch: chunk ci uses table entry ci % n_slots
uint64_t ck_stream_ctl_bytes();
uint64_t ck_stream_error_offset();
// the page-locked, device-visible block host and kernel talk through (host = its host address, host_dev = its
// device address): chunk table entries, {chunks, items} announced, closed; completion word per slot
uint64_t ck_stream_host_bytes();
// Results the kernel stores straight into page-locked host memory instead of the chunk's workspace (device-visible addresses, null =
// the workspace): the ops at the chunk's base, the three short arrays at the chunk's first pair; start_add is added to every ops
// offset the kernel reports (the chunk's base in the caller's ops array, when ops_start is the caller's own array)
struct CkStreamDirect {
    uint8_t* ops = nullptr;
    float* scores = nullptr;
    uint64_t* ops_start = nullptr;
    uint32_t* ops_len = nullptr;
    uint64_t start_add = 0;
};
void ck_stream_fill_chunk(void* host, void* host_dev, int slot, const void* arena, const BatchDeviceView& v, uint32_t n_pairs,
                          uint32_t first_ticket, uint32_t chunk_no, const CkStreamDirect& direct = CkStreamDirect{});
void ck_stream_host_announce(void* host, uint32_t chunks, uint32_t items);
void ck_stream_host_close(void* host);
unsigned long long ck_stream_host_bad(void* host, int slot);  // (0, or the kernel's report of a code out of range: viterbi_ck.hip ck_report_bad)
double ck_stream_host_done_ms(void* host, int slot);  // (ms after the kernel started: when the slot's chunk was complete)
volatile uint32_t* ck_stream_host_done_flag(void* host, int slot);
// wave_ck: ck_scratch_waves() checkpoint slots of wave_slot_dwords each; wave_scratch: as many traceback scratch areas
hipError_t launch_viterbi_ck_stream(const float* table, GapConsts k, bool shared_tab, void* ctl, const void* host_words, uint32_t* wave_ck,
                                    uint64_t wave_slot_dwords, uint32_t* wave_scratch, uint32_t band, hipStream_t stream);
uint32_t ck_band_setting();  // the default half width of the kept checkpoint band (COATI_HIP_CK_BAND, else 64 steps)
hipError_t launch_ck_all_flags(const BatchDeviceView& v, uint32_t pair, uint32_t* scratch, uint32_t n_waves, uint8_t* out,
                               hipStream_t stream);
hipError_t launch_dp_generic(const BatchDeviceView& v, bool forward, hipStream_t stream);
hipError_t launch_forward_l1(const BatchDeviceView& v, bool one_table, hipStream_t stream);
hipError_t launch_viterbi_k(const BatchDeviceView& v, bool narrow_only, hipStream_t stream);
hipError_t launch_forward_k(const BatchDeviceView& v, hipStream_t stream);
// COATI_HIP_FORWARD_FAST=1: Forward with the hardware exp2/log2 (fast, within 1e-5 of the CPU) instead
// of the bit-exact libm restatements (read once).
bool forward_fast_math();
hipError_t launch_sampleback(const BatchDeviceView& v, uint32_t n_samples, bool independent, uint64_t* rng_states,
                             const uint64_t* sample_base, uint8_t* ops, uint64_t* ops_start, uint32_t* ops_len,
                             float* log_weights, hipStream_t stream);
hipError_t launch_rng_f24(const uint64_t state[2], uint32_t n, float* d_out, hipStream_t stream);

// Speculative exact-stream sampling (sampleback.hip): a walker that assumes its sample starts
// `offset` draws after the pair's chunk origin, and the record that promotes a candidate to a result.
struct SpecCandidate {
    uint32_t pair, offset;
    uint64_t slot;  // start of its la + lb bytes in the temporary ops arena
};
struct SpecCommit {
    uint32_t cand, pad_;
    uint64_t slot_end;   // end of the sample's slot in the result ops arena
    uint64_t out_index;  // pair * n_samples + sample
};
hipError_t launch_spec_walk(const BatchDeviceView& v, const uint64_t* origin_state, const uint64_t* mult_pow,
                            const SpecCandidate* cands, uint32_t n_cands, uint8_t* tmp_ops, uint64_t* c_start,
                            uint32_t* c_len, float* c_lw, uint32_t* c_draws, hipStream_t stream);
hipError_t launch_spec_commit(const SpecCommit* commits, uint32_t n_commits, const uint8_t* tmp_ops,
                              const uint64_t* c_start, const uint32_t* c_len, const float* c_lw, uint8_t* ops,
                              uint64_t* ops_start, uint32_t* ops_len, float* log_weights, hipStream_t stream);

// round 4: the speculation loop of the table path on the DEVICE (sampleback.hip "device rounds"): per pair what the host
// loop keeps (sample_host.hip), a round's windows, and three launches per round with no host round trip in between
struct SpecPairState {
    uint64_t origin;     // draws consumed by the samples resolved so far
    uint32_t done, cnt;  // samples resolved; observations in the running estimate
    double mean, m2;     // draws per sample: Welford
    uint32_t n_cands, n_windows, rank, n_draws;  // this round: candidates, windows, which share of the candidate array, draws of the stream its walks may take
};
struct SpecWindow {  // the candidates of one (pair, sample-in-chunk): offsets lo .. hi from the pair's origin
    uint32_t first, lo, hi;
};
struct SpecRound {  // written by the plan launch: unfinished pairs before the round, candidates per share, pairs in the round, draws per slice
    uint32_t active, share, ranked, slice;
};
constexpr uint32_t kSpecDrawFloats = 16u << 20;  // the round's table of stream draws (64 MB), cut into one slice per pair of the round
constexpr uint32_t kSpecChunkMax = 512;  // samples speculated per pair and round, at most
hipError_t launch_spec_round(const BatchDeviceView& v, const uint64_t* tab_off, uint32_t band, const void* steps, const uint64_t* state0, const uint64_t* mult_pow,
                             uint32_t n_samples, uint32_t max_cands, uint32_t max_width, double z, SpecPairState* states, SpecWindow* windows,
                             uint32_t* rank_pair, SpecRound* round, float* draw_table, const uint64_t* thr_off, const void* thr_m, uint32_t* c_draws,
                             uint64_t* sample_off, hipStream_t stream);

// The step table covers a BAND of diagonals around the straight line of a pair: d = i - j from min(0, la - lb) - B to
// max(0, la - lb) + B (clipped to the matrix).  A sampled path leaves it only by an excess of > B gap columns of one kind;
// the walkers compute the entries of cells outside it as they go (sampleback.hip: step_entry).  16 pairs of 1 kb at B = 64:
// 150 MB and 0.15 ms to build instead of 1.15 GB and 0.8 ms.
struct StepBand {
    int64_t dhi;      // the largest d = i - j in the band: column j of row i is entry k = j - i + dhi of the row
    uint32_t width;   // diagonals in the band = entries per row
    uint32_t diag_len;  // min(la, lb): entries per diagonal of the diagonal-major threshold array
};
__host__ __device__ inline StepBand step_band(uint32_t la, uint32_t lb, uint32_t half) {
    if(la == 0 || lb == 0) return StepBand{0, 0, 0};
    const int64_t delta = static_cast<int64_t>(la) - static_cast<int64_t>(lb);
    int64_t dlo = (delta < 0 ? delta : 0) - static_cast<int64_t>(half), dhi = (delta > 0 ? delta : 0) + static_cast<int64_t>(half);
    if(dlo < 1 - static_cast<int64_t>(lb)) dlo = 1 - static_cast<int64_t>(lb);
    if(dhi > static_cast<int64_t>(la) - 1) dhi = static_cast<int64_t>(la) - 1;
    return StepBand{dhi, static_cast<uint32_t>(dhi - dlo + 1), la < lb ? la : lb};
}

// round 4: the step table of the exact-stream sampler (sampleback.hip): thresholds and log-weight increments per
// (body cell, state), 24 bytes each, row-major per pair from entry tab_off[pair]; gap_len 1
uint64_t step_entry_bytes();
hipError_t launch_step_table(const BatchDeviceView& v, const uint64_t* tab_off, uint64_t max_cells, uint32_t band, void* steps, const uint64_t* thr_off,
                             void* thr_m, hipStream_t stream);  // thr_m (may be null): the M-state thresholds again, 12 bytes per body cell, diagonal-major (sampleback.hip)

hipError_t launch_spec_len(const BatchDeviceView& v, const uint64_t* tab_off, uint32_t band, const void* steps, const uint64_t* origin_state,
                           const uint64_t* mult_pow, const SpecCandidate* cands, uint32_t n_cands, uint32_t* c_draws, hipStream_t stream);
hipError_t launch_final_walk(const BatchDeviceView& v, const uint64_t* tab_off, uint32_t band, const void* steps, const uint64_t* start_state,
                             const uint64_t* mult_pow, const uint64_t* sample_offset, const uint64_t* sample_base, uint32_t n_samples, uint8_t* ops, uint64_t* ops_start,
                             uint32_t* ops_len, float* log_weights, hipStream_t stream);

}  // namespace coati_hip_detail
#endif
