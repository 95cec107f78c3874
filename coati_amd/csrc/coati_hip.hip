// libcoati_hip.so -- MI355X (gfx950 / CDNA4) implementation of COATi's marginal
// pairwise DP hot path behind the C ABI of include/coati_hip.h.
//
// What it replaces in the reference (all CPU, one pair per process):
//   forward_impl<tropical, align_pair_work_mem_t>   src/lib/align_pair.cc:62-139
//   traceback<tropical> / max_mdi / max_mi          src/lib/align_pair.cc:210-303
//
// Design (see DESIGN.md for the derivations):
//   * one sequence pair per 64-lane wavefront; a lane owns 16 consecutive
//     descendant columns and walks down the ancestor rows, skewed by one row per
//     lane (anti-diagonal wavefront).  All M/D/I state lives in registers;
//     nothing of the fp32 matrices ever reaches memory.
//   * neighbour hand-off between lanes is a DPP `wave_shr:1` move (no LDS).
//   * the 183x15 substitution table is staged in LDS (row stride 17 floats so
//     that a wave's 64 different rows spread over the 32 banks).
//   * the traceback is NOT an arg-max recorded in the fill.  The reference
//     re-derives each decision from the stored scores of the cell it arrives at
//     (align_pair.cc:275-296), so the kernel evaluates exactly those five
//     comparisons per cell and stores them as five bit-planes: the 64-bit
//     `v_cmp` lane masks are moved to lanes with v_writelane and stored as one
//     coalesced 640-byte row per wavefront step (5 bits per cell in HBM).
//   * a second kernel walks the bit-planes (one lane per pair) and emits one op
//     byte per alignment column.
//
// fp32 only, adds/max/compares in the reference's evaluation order; built with
// -ffp-contract=off so nothing is fused.
#include "coati_hip.h"

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdarg>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

// ============================================================================
// device side
// ============================================================================
namespace {

constexpr int kWave = 64;
constexpr int kW = 16;                       // columns per lane
constexpr int kStrip = kWave * kW;           // columns per strip (1024)
constexpr int kTabRows = COATI_HIP_TABLE_ROWS;
constexpr int kTabCols = COATI_HIP_TABLE_COLS;
constexpr int kTabStride = 17;               // LDS row stride in floats (bank spread, measured)
constexpr int kPairDwords = 5 * kWave;       // dwords one pair of wavefront steps stores (320)
constexpr int kFillWaves = 4;                // sequence pairs per workgroup
constexpr float kLowest = -FLT_MAX;          // semiring zero(), semiring.hpp:83,113

// Decision bits per cell.  Bit = 1 means:
//   M1: D beats M after a match move     (max_mdi first test,  align_pair.cc:213-216)
//   M2: I beats max(M,D) after a match   (max_mdi second test, align_pair.cc:217-219)
//   D1, D2: the same two tests after a deletion move (align_pair.cc:285-287)
//   IM: M beats I after an insertion move (max_mi, align_pair.cc:230-232; tie -> I)
// Three per-lane accumulators: A = (M1,M2) pairs, B = (D1,D2) pairs, C = IM.
enum : int { ACC_A = 0, ACC_B = 1, ACC_C = 2, kAccs = 3 };

struct GapConsts {
    float ng, gs, go, ge;  // no_gap, gap_stop, gap_open, gap_extend (log space)
};

struct PairDesc {
    uint64_t a_off, b_off;  // into the concatenated code arrays
    uint64_t flags_off;     // dwords into the bit-plane arena
    uint64_t bnd_off;       // floats into the strip-boundary arena
    uint64_t ops_off;       // slot start in the ops arena (slot = la + lb bytes)
    uint32_t la, lb;
};

// HBM layout of the decision bits of one strip (1024 columns) of one pair, per
// pair of wavefront steps kp = k >> 1 (k = body_row + lane), 320 dwords:
//   [kp*320 +   0 + lane]  A of the even step     [kp*320 + 128 + lane]  A of the odd step
//   [kp*320 +  64 + lane]  B of the even step     [kp*320 + 192 + lane]  B of the odd step
//   [kp*320 + 256 + lane]  C: bits 31..16 even step, 15..0 odd step
// In A/B the lane's column c (0..15) holds its first test at bit 31-2c and its
// second at bit 30-2c; in C column c is bit 15-c of its half.
// Every store is a fully coalesced 256-byte row: 5 bits per DP cell.
__host__ __device__ inline uint64_t strip_dwords(uint32_t la) {
    return static_cast<uint64_t>((la + kWave) / 2) * kPairDwords;
}
__host__ __device__ inline uint32_t n_strips(uint32_t lb) { return (lb + kStrip - 1) / kStrip; }

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
// Register state of one lane: its 16 columns of the row it processed last.
struct LaneState {
    float X[kW];  // max((M+ng)+ng, D+gs, (I+gs)+ng): feeds M of the next diagonal cell
    float Y[kW];  // max((M+ng)+go, D+ge, (I+gs)+go): the D value of the cell below (gap_len 1)
    float xlast_old;  // X[15] of the row before: the right neighbour's diagonal input
    float zlast;      // max(M+go, I+ge) of column 15: the right neighbour's I value
    uint32_t acc[kAccs];    // decision bits, shifted in cell by cell
};

// One DP cell for gap_len == 1 (align_pair.cc:97-124 with look_back = 1, where
// power(gap_extend, 0) is -0.0f and adding it is the identity).  Because fp32
// addition is monotone, max(x1+s, x2+s, x3+s) == max(x1,x2,x3)+s bit for bit,
// so M = X(diagonal cell) + s.
//
// The five decisions are max_mdi / max_mi (align_pair.cc:210-232) on the
// expressions of align_pair.cc:275-296.  Each `p > q` is taken as the sign bit
// of q - p: exact, because fp32 subtraction of two finite numbers is zero only
// when they are equal (gradual underflow is on) and no -0.0f occurs here.  On
// gfx950 v_sub_f32 issues at twice the rate of v_cmp_f32 and the bit is
// deposited with a single v_alignbit_b32 (measured: tools/ubench).
// One DP cell = ONE asm block of 27 VALU instructions with a fixed order and a
// hand register allocation.  Why not leave it to the compiler (all measured or
// observed):
//  * on gfx950 v_add/v_sub_f32 and v_add_u32 issue every 2 cycles, v_max_f32 and
//    v_alignbit_b32 have a 4-cycle initiation interval; a slow op costs nothing
//    extra only if fast ops sit on both sides of it.  The order below alternates
//    them (F/S) and keeps every consumer >= 2 instructions behind its producer.
//  * hipcc batches the X/Y maxes of a whole row (32 back-to-back v_max), and once
//    values are opaque adds canonicalising v_max around every fmaxf (IEEE mode);
//    between adjacent dependent inline-asm statements the hazard recognizer
//    inserts s_nop.  One block per cell has none of that.
// No instruction here has a software-visible hazard (no trans ops, no DPP or
// readlane consumer inside).  The deposit of the cell's last decision (D2) is
// carried in `pend` into the next cell; the LDS address of this column's score
// for the NEXT wavefront step is computed here, the ds_read is issued by the
// compiler right after the block (so that it also places the s_waitcnt).
#define COATI_CELL_HEAD                                                                     \
    "v_add_f32 %[t0], %[diag], %[s]\n\t"      /* F  M  = diag + s                        */ \
    "v_add_f32 %[t1], %[ge], %[zl]\n\t"       /* F  z2 = I + ge                          */
#define COATI_CELL_PEND                                                                     \
    "v_alignbit_b32 %[aB], %[aB], %[pend], 31\n\t" /* S  D2 of the previous cell         */
#define COATI_CELL_BODY                                                                     \
    "v_add_f32 %[t2], %[gs], %[zl]\n\t"       /* F  i1 = I + gs                          */ \
    "v_add_f32 %[t3], %[go], %[t0]\n\t"       /* F  z1 = M + go                          */ \
    "v_add_f32 %[t0], %[ng], %[t0]\n\t"       /* F  m1 = M + ng                          */ \
    "v_max_f32 %[zl], %[t3], %[t1]\n\t"       /* S  Z  = max(z1,z2) -> I of next column  */ \
    "v_add_f32 %[t4], %[ng], %[t0]\n\t"       /* F  x1 = m1 + ng                         */ \
    "v_add_f32 %[t5], %[gs], %[y]\n\t"        /* F  x2 = D + gs                          */ \
    "v_sub_f32 %[t1], %[t1], %[t3]\n\t"       /* F  z2 - z1  (sign: z1 > z2)             */ \
    "v_max_f32 %[t3], %[t4], %[t5]\n\t"       /* S  xm = max(x1,x2)                      */ \
    "v_add_f32 %[pend], %[ng], %[t2]\n\t"     /* F  x3 = i1 + ng                         */ \
    "v_alignbit_b32 %[aC], %[aC], %[t1], 31\n\t" /* S  IM                                */ \
    "v_sub_f32 %[t1], %[t4], %[t5]\n\t"       /* F  x1 - x2  (sign: x2 > x1)             */ \
    "v_max_f32 %[x], %[t3], %[pend]\n\t"      /* S  X  = max(xm,x3)                      */ \
    "v_sub_f32 %[t4], %[t3], %[pend]\n\t"     /* F  xm - x3  (sign: x3 > xm)             */ \
    "v_add_f32 %[t5], %[go], %[t0]\n\t"       /* F  y1 = m1 + go                         */ \
    "v_alignbit_b32 %[aA], %[aA], %[t1], 31\n\t" /* S  M1                                */ \
    "v_add_f32 %[t1], %[ge], %[y]\n\t"        /* F  y2 = D + ge                          */ \
    "v_add_f32 %[t3], %[go], %[t2]\n\t"       /* F  y3 = i1 + go                         */ \
    "v_max_f32 %[t0], %[t5], %[t1]\n\t"       /* S  ym = max(y1,y2)                      */ \
    "v_sub_f32 %[t2], %[t5], %[t1]\n\t"       /* F  y1 - y2  (sign: y2 > y1)             */ \
    "v_alignbit_b32 %[aA], %[aA], %[t4], 31\n\t" /* S  M2                                */ \
    "v_add_u32 %[addr], %[lds], %[boff]\n\t"  /* F  LDS address of next step's score     */ \
    "v_max_f32 %[y], %[t0], %[t3]\n\t"        /* S  Y  = max(ym,y3)                      */ \
    "v_sub_f32 %[pend], %[t0], %[t3]\n\t"     /* F  ym - y3  (sign: y3 > ym), carried    */ \
    "v_alignbit_b32 %[aB], %[aB], %[t2], 31"  /* S  D1                                   */

template <int C>
__device__ __forceinline__ void cell_l1(const GapConsts& k, LaneState& st, float& diag, float& zl, float& pend,
                                        float& s, uint32_t lds_next_row, uint32_t boff) {
    float x_new, t0, t1, t2, t3, t4, t5;
    uint32_t addr;
#define COATI_CELL_OPERANDS                                                                              \
    : [x] "=&v"(x_new), [y] "+v"(st.Y[C]), [zl] "+v"(zl), [pend] "+v"(pend), [aA] "+v"(st.acc[ACC_A]),   \
      [aB] "+v"(st.acc[ACC_B]), [aC] "+v"(st.acc[ACC_C]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), \
      [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [addr] "=&v"(addr)                                  \
    : [diag] "v"(diag), [s] "v"(s), [lds] "v"(lds_next_row), [boff] "v"(boff), [ng] "s"(k.ng),          \
      [gs] "s"(k.gs), [go] "s"(k.go), [ge] "s"(k.ge)
    if constexpr(C > 0) {
        asm volatile(COATI_CELL_HEAD COATI_CELL_PEND COATI_CELL_BODY COATI_CELL_OPERANDS);
    } else {
        asm volatile(COATI_CELL_HEAD COATI_CELL_BODY COATI_CELL_OPERANDS);
    }
#undef COATI_CELL_OPERANDS
    diag = st.X[C];  // the next column's diagonal input is this column's previous-row X
    st.X[C] = x_new;
    // s was consumed by the block's first instruction: reuse it for the next step's score
    s = *reinterpret_cast<const __attribute__((address_space(3))) float*>(addr);
}

template <int... C>
__device__ __forceinline__ void row_l1(const GapConsts& k, LaneState& st, float diag, float zl,
                                       float (&s)[kW], uint32_t lds_next_row, const uint32_t (&boff)[kW],
                                       std::integer_sequence<int, C...>) {
    st.xlast_old = st.X[kW - 1];
    float pend = 0.0f;
    (cell_l1<C>(k, st, diag, zl, pend, s[C], lds_next_row, boff[C]), ...);
    asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(st.acc[ACC_B]) : "v"(pend));  // D2 of column 15
    st.zlast = zl;
}

struct CellAddr {
    uint64_t pair_base;  // dword index of the step pair
    uint32_t odd, t, c;
};
__device__ __forceinline__ CellAddr cell_addr(uint64_t base, uint32_t la, uint32_t bi, uint32_t bj) {
    const uint32_t strip = bj / kStrip, t = (bj % kStrip) / kW, c = bj % kW;
    const uint32_t kstep = bi + t;
    return {base + strip * strip_dwords(la) + static_cast<uint64_t>(kstep >> 1) * kPairDwords, kstep & 1u, t, c};
}
// two-bit decision (first test, second test) of accumulator A (which = 0) or B (which = 1)
__device__ __forceinline__ uint32_t pair_bits(const uint32_t* __restrict__ flags, const CellAddr& ca, int which) {
    const uint32_t w = flags[ca.pair_base + ca.odd * (2 * kWave) + which * kWave + ca.t];
    return (w >> (30 - 2 * ca.c)) & 3u;  // bit1 = first test, bit0 = second test
}
__device__ __forceinline__ uint32_t im_bit(const uint32_t* __restrict__ flags, const CellAddr& ca) {
    const uint32_t w = flags[ca.pair_base + 4 * kWave + ca.t];
    return (w >> ((ca.odd ? 0u : 16u) + (kW - 1 - ca.c))) & 1u;
}

__device__ __forceinline__ int state_after(const uint32_t* __restrict__ flags, uint64_t base,
                                           uint32_t la, uint32_t bi, uint32_t bj, int moved) {
    const CellAddr ca = cell_addr(base, la, bi, bj);
    if(moved == COATI_HIP_OP_INS) return im_bit(flags, ca) ? COATI_HIP_OP_MATCH : COATI_HIP_OP_INS;
    const uint32_t two = pair_bits(flags, ca, moved == COATI_HIP_OP_DEL ? 1 : 0);
    if(two & 1u) return COATI_HIP_OP_INS;  // second test: I beats max(M,D)
    return (two & 2u) ? COATI_HIP_OP_DEL : COATI_HIP_OP_MATCH;
}

// State the reference's walk is in after arriving at matrix cell (i, j) by a move
// of kind `moved` (align_pair.cc:275-296).  On column 0 only D, on row 0 only I
// is finite (align_pair.cc:82-91), so margin cells need no stored bits.
constexpr int kWalkEnd = 3;
__device__ __forceinline__ int arrival_state(const uint32_t* __restrict__ flags, uint64_t base, uint32_t la,
                                             uint32_t i, uint32_t j, int moved) {
    if(i == 0 && j == 0) return kWalkEnd;
    if(j == 0) return COATI_HIP_OP_DEL;
    if(i == 0) return COATI_HIP_OP_INS;
    return state_after(flags, base, la, i - 1, j - 1, moved);
}

// traceback<tropical> (align_pair.cc:249-303, gap_len 1) by one WAVEFRONT.
// A walk is a chain of dependent loads, but it consists of long runs of the same
// move.  So the 64 lanes speculate: lane l looks up the state the walk would be
// in after l+1 further moves of the current kind; a ballot finds the first lane
// where the run ends; all moves up to there are emitted at once (coalesced
// byte stores) and the walk jumps.  Memory round trips per pair drop from
// len_a+len_b to about (number of runs + length/64).
// Ops are written right-to-left into the pair's slot so they end up in alignment
// order; all lanes must call this (it uses ballots).
__device__ __forceinline__ void walk_pair_l1(int lane, const GapConsts& k, const PairDesc& pd, uint32_t pair,
                                             const uint32_t* __restrict__ flags, uint8_t* __restrict__ ops,
                                             uint64_t* __restrict__ ops_start, uint32_t* __restrict__ ops_len,
                                             float* __restrict__ scores) {
    const uint32_t la = pd.la, lb = pd.lb;
    uint32_t i = la, j = lb;  // matrix coordinates (row 0 / column 0 are the margins)
    uint64_t pos = pd.ops_off + la + lb;
    int st;
    if(la == 0 || lb == 0) {
        // No body cell: the last cell is a margin cell (align_pair.cc:82-91,130-138).
        float m = kLowest, d = kLowest, in = kLowest;
        if(la == 0 && lb == 0) m = 0.0f;
        if(la > 0) d = (k.ng + k.go) + k.ge * static_cast<float>(la - 1);
        if(lb > 0) in = k.go + k.ge * static_cast<float>(lb - 1);
        const float tm = (m + k.ng) + k.ng, td = d + k.gs, ti = (in + k.gs) + k.ng;
        if(lane == 0) scores[pair] = fmaxf(fmaxf(tm, td), ti);
        st = la > 0 ? COATI_HIP_OP_DEL : (lb > 0 ? COATI_HIP_OP_INS : kWalkEnd);
    } else {
        // max_mdi of the terminal-adjusted last cell == its "after match" decision
        st = __builtin_amdgcn_readfirstlane(state_after(flags, pd.flags_off, la, la - 1, lb - 1, COATI_HIP_OP_MATCH));
    }
    while(st != kWalkEnd) {
        const uint32_t di = st != COATI_HIP_OP_INS ? 1u : 0u, dj = st != COATI_HIP_OP_DEL ? 1u : 0u;
        // lane l: where the walk is after l+1 more moves of kind st, and in which state
        const uint32_t step = static_cast<uint32_t>(lane) + 1u;
        const bool valid = di * step <= i && dj * step <= j;
        int next = kWalkEnd;
        if(valid) next = arrival_state(flags, pd.flags_off, la, i - di * step, j - dj * step, st);
        const unsigned long long cont = __builtin_amdgcn_ballot_w64(valid && next == st);
        const uint32_t run = cont == ~0ull ? kWave : static_cast<uint32_t>(__builtin_ctzll(~cont));  // lanes that continue
        const uint32_t moves = run == kWave ? kWave : run + 1u;
        if(static_cast<uint32_t>(lane) < moves) ops[pos - 1 - lane] = static_cast<uint8_t>(st);
        pos -= moves;
        i -= di * moves;
        j -= dj * moves;
        if(run < kWave) st = __builtin_amdgcn_readlane(next, static_cast<int>(run));
    }
    if(lane == 0) {
        ops_start[pair] = pos;
        ops_len[pair] = static_cast<uint32_t>(pd.ops_off + la + lb - pos);
    }
}

// Read-only per-strip context of one wavefront.
struct StripCtx {
    GapConsts k;
    uint32_t la, col0, nsteps, pair, lds_tab;
    int lane, last_lane, last_c;
    bool last_strip;
    uint32_t* fout;
    float *bnd_x, *bnd_z, *scores;
};

// Up to 64 wavefront steps.  kFirst: chunk 0 only, where lane l starts (takes its
// margin-row state) at step l.  The main loop is a separate instantiation so
// that it carries no trace of the start-up code (spill reloads there would put
// an s_waitcnt vmcnt(0) -- a wait for the previous step's HBM stores -- into
// every step).
template <bool kFirst>
__device__ __forceinline__ void run_chunk(const StripCtx& cx, LaneState& st, uint32_t& arow, float (&s)[kW],
                                          const uint32_t (&boff)[kW], uint32_t kbase, uint32_t a_chunk, float bx,
                                          float bz) {
    const GapConsts& k = cx.k;
    const int lane = cx.lane;
    const uint32_t kend = min(static_cast<uint32_t>(kWave), cx.nsteps - kbase);
    for(uint32_t kk = 0; kk < kend; ++kk) {
        const uint32_t kstep = kbase + kk;
        if constexpr(kFirst) {
            if(kk == static_cast<uint32_t>(lane)) {
                // This lane starts now: state of the margin row (matrix row 0,
                // align_pair.cc:88-90): M = D = lowest, I = go + ge*float(j-1).
                uint32_t bj0 = cx.col0 + lane * kW;
                asm volatile("" : "+v"(bj0));  // compute in place: hoisted, these 32 values get spilled
#pragma unroll
                for(int c = 0; c < kW; ++c) {
                    const float im = k.go + k.ge * static_cast<float>(bj0 + c);
                    const float i1 = im + k.gs;
                    st.X[c] = i1 + k.ng;
                    st.Y[c] = i1 + k.go;
                }
                if(!cx.last_strip && lane == kWave - 1) cx.bnd_x[0] = st.X[kW - 1];
            }
        }
        // ---- hand-off from the left neighbour (full exec)
        const float diag = shift_in(st.xlast_old, read_lane(bx, kk));
        const float zl = shift_in(st.zlast, read_lane(bz, kk));
        const uint32_t arow_next = shift_in(arow, read_lane(a_chunk, kk));
        // ---- the 16 cells (and the LDS gather for the next step)
        row_l1(k, st, diag, zl, s, cx.lds_tab + arow_next, boff, std::make_integer_sequence<int, kW>{});
        arow = arow_next;
        // ---- decision bits: two coalesced 256-byte rows per step, a third every second step
        {
            uint32_t* dst = cx.fout + static_cast<uint64_t>(kstep >> 1) * kPairDwords + (kstep & 1u) * (2 * kWave);
            dst[0] = st.acc[ACC_A];
            dst[kWave] = st.acc[ACC_B];
            if(kstep & 1u) dst[2 * kWave] = st.acc[ACC_C];  // = pair base + 256
        }
        const int r = static_cast<int>(kstep) - lane;  // body row this lane just did
        if(!cx.last_strip && lane == kWave - 1 && r >= 0 && r < static_cast<int>(cx.la)) {
            cx.bnd_x[r + 1] = st.X[kW - 1];
            cx.bnd_z[r] = st.zlast;
        }
        if(cx.last_strip && r == static_cast<int>(cx.la) - 1 && lane == cx.last_lane) {
            // score = max(M,D,I) of the terminal-adjusted last cell
            // (align_pair.cc:130-138,265) = X of the last body cell.
            float sc = st.X[0];
#pragma unroll
            for(int c = 1; c < kW; ++c) sc = (c == cx.last_c) ? st.X[c] : sc;
            cx.scores[cx.pair] = sc;
        }
    }
}

// Viterbi fill for gap_len == 1.  PERSISTENT: the grid is sized to fill every CU
// with the same number of workgroups (host: fill_launch_shape) and each
// wavefront pulls pair indices from an atomic queue until it is empty.  (With one
// workgroup per 4 pairs the hardware dispatcher packs workgroups unevenly --
// in-kernel clocks showed SIMDs running 2x the waves of others -- and a kernel
// took ~2x the time its work implies.)  `order` lists the pairs longest first.
__global__ __launch_bounds__(kFillWaves* kWave, 3) void viterbi_l1(
    const float* __restrict__ table, GapConsts k, const PairDesc* __restrict__ pairs,
    const uint32_t* __restrict__ order, uint32_t n_pairs, uint32_t* __restrict__ queue,
    const uint8_t* __restrict__ a_cat, const uint8_t* __restrict__ b_cat,
    uint32_t* __restrict__ flags, float* __restrict__ bnd, float* __restrict__ scores,
    uint8_t* __restrict__ ops, uint64_t* __restrict__ ops_start, uint32_t* __restrict__ ops_len) {
    __shared__ float tab[kTabRows * kTabStride];
    for(int idx = threadIdx.x; idx < kTabRows * kTabCols; idx += blockDim.x) {
        const int r = idx / kTabCols, c = idx - r * kTabCols;
        tab[r * kTabStride + c] = table[idx];
    }
    __syncthreads();

    const int lane = threadIdx.x & (kWave - 1);
    const char* tab_bytes = reinterpret_cast<const char*>(tab);
    const uint32_t lds_tab = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(tab));  // LDS byte address
    for(;;) {
    uint32_t ticket = 0;
    if(lane == 0) ticket = atomicAdd(queue, 1u);
    ticket = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ticket)));
    if(ticket >= n_pairs) break;
    const uint32_t pair = order[ticket];
    const PairDesc pd = pairs[pair];
    const uint32_t la = pd.la, lb = pd.lb;
    if(la > 0 && lb > 0) {  // (without body cells only the margins are walked)
    const uint8_t* __restrict__ a = a_cat + pd.a_off;
    const uint8_t* __restrict__ b = b_cat + pd.b_off;

    const uint32_t strips = n_strips(lb);
    for(uint32_t strip = 0; strip < strips; ++strip) {
        const uint32_t col0 = strip * kStrip;
        const uint32_t ncol = min(static_cast<uint32_t>(kStrip), lb - col0);
        const uint32_t nlanes = (ncol + kW - 1) / kW;
        const uint32_t nsteps = la + nlanes - 1;
        const bool last_strip = strip + 1 == strips;
        uint32_t* __restrict__ fout = flags + pd.flags_off + strip * strip_dwords(la) + lane;
        // strip-boundary columns: [0, la] = X of the last column (index r = X of
        // body row r-1; index 0 = margin row), [la+1, 2la] = Z of body row r.
        float* __restrict__ bnd_x = bnd + pd.bnd_off;
        float* __restrict__ bnd_z = bnd_x + (la + 1);

        // byte offsets of this lane's 16 table columns
        uint32_t boff[kW];
#pragma unroll
        for(int c = 0; c < kW; ++c) {
            const uint32_t bj = col0 + lane * kW + c;
            boff[c] = bj < lb ? static_cast<uint32_t>(b[bj]) * 4u : 0u;
        }

        const StripCtx cx{k, la, col0, nsteps, pair, lds_tab, lane,
                          static_cast<int>(((lb - 1) & (kStrip - 1)) / kW), static_cast<int>((lb - 1) & (kW - 1)),
                          last_strip, fout, bnd_x, bnd_z, scores};
        LaneState st;
#pragma unroll
        for(int c = 0; c < kW; ++c) st.X[c] = st.Y[c] = 0.0f;
#pragma unroll
        for(int p = 0; p < kAccs; ++p) st.acc[p] = 0u;
        st.xlast_old = 0.0f;
        st.zlast = 0.0f;
        // table-row byte offset of the row this lane processes at the CURRENT step,
        // and the 16 substitution scores gathered for it one step earlier
        uint32_t arow = lane == 0 ? static_cast<uint32_t>(a[0]) * (kTabStride * 4u) : 0u;
        float s[kW];
#pragma unroll
        for(int c = 0; c < kW; ++c) s[c] = *reinterpret_cast<const float*>(tab_bytes + arow + boff[c]);

        for(uint32_t kbase = 0; kbase < nsteps; kbase += kWave) {
            // ---- per-64-step chunk: lane l fetches what lane 0 will need at step kbase+l
            // (boundary column) and at step kbase+l+1 (ancestor code: gathered a step ahead)
            const uint32_t crow = kbase + lane;
            uint32_t a_chunk = 0;
            float bx = kLowest, bz = kLowest;
            if(crow + 1 < la) a_chunk = static_cast<uint32_t>(a[crow + 1]) * (kTabStride * 4u);
            if(crow < la) {
                if(strip == 0) {
                    // column 0 of the matrix (align_pair.cc:82-86): M(0,0)=0, D(i,0) margin
                    if(crow == 0) {
                        bx = (0.0f + k.ng) + k.ng;
                    } else {
                        const float dm = (k.ng + k.go) + k.ge * static_cast<float>(crow - 1);
                        bx = dm + k.gs;
                    }
                } else {
                    bx = bnd_x[crow];
                    bz = bnd_z[crow];
                }
            }
            // Consume the chunk loads HERE (one wait per 64 steps), not inside the step loop.
            asm volatile("" : "+v"(a_chunk), "+v"(bx), "+v"(bz));
            if(kbase == 0)
                run_chunk<true>(cx, st, arow, s, boff, kbase, a_chunk, bx, bz);
            else
                run_chunk<false>(cx, st, arow, s, boff, kbase, a_chunk, bx, bz);
        }
        if(nsteps & 1u)  // the last (even) step has no odd partner: flush its IM bits to the high half
            fout[static_cast<uint64_t>(nsteps >> 1) * kPairDwords + 4 * kWave] = st.acc[ACC_C] << 16;
    }
    }
    // ---- traceback of this pair by the same wavefront, while its bits are still
    // in L2.  The wave reads what it wrote itself: wait until its stores are
    // acknowledged; nobody read these (128-byte aligned) lines before, so L1 is cold.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    walk_pair_l1(lane, k, pd, pair, flags, ops, ops_start, ops_len, scores);
    }  // next ticket
}

// Debug: decode one pair's bit-planes into the oracle's byte-per-cell encoding.
__global__ void decode_flags(const PairDesc* __restrict__ pairs, uint32_t pair,
                             const uint32_t* __restrict__ flags, uint8_t* __restrict__ out) {
    const PairDesc pd = pairs[pair];
    const uint64_t n = static_cast<uint64_t>(pd.la) * pd.lb;
    for(uint64_t idx = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; idx < n;
        idx += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint32_t bi = idx / pd.lb, bj = idx % pd.lb;
        const CellAddr ca = cell_addr(pd.flags_off, pd.la, bi, bj);
        const uint32_t mm = pair_bits(flags, ca, 0), dd = pair_bits(flags, ca, 1), im = im_bit(flags, ca);
        const uint32_t fm = (mm & 1u) ? 2u : (mm >> 1), fd = (dd & 1u) ? 2u : (dd >> 1);
        out[idx] = static_cast<uint8_t>(fm | (fd << 2) | ((im ^ 1u) << 4));
    }
}

}  // namespace

// ============================================================================
// host side of the C ABI
// ============================================================================
namespace {

thread_local std::string g_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if(e_ != hipSuccess)                                                                \
            return fail(e_ == hipErrorOutOfMemory ? COATI_HIP_ENOMEM : COATI_HIP_EHIP,      \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                     \
    } while(0)

// Launch shape of the persistent fill kernel: `blocks_per_cu` workgroups on each
// of the 256 CUs (one wave per SIMD each), enforced by padding the launch with
// unused dynamic LDS so that exactly that many fit.  More resident waves hide
// latency better; fewer quantise the end of a small batch more finely.  Cost of
// one pair relative to a saturated SIMD when w waves share it (measured,
// in-kernel clocks): w=1 1.85, w=2 1.10, w=3 1.03.
struct FillShape {
    uint32_t grid;
    size_t dynamic_lds;
};
FillShape fill_launch_shape(uint32_t n_pairs) {
    constexpr uint32_t kCUs = 256, kSimds = kCUs * 4;
    constexpr int kMaxBlocks = 3;  // 168 VGPRs -> 3 waves per SIMD
    constexpr double kCost[4] = {0.0, 1.85, 1.10, 1.03};
    static const int forced = [] {
        const char* e = std::getenv("COATI_HIP_FILL_BLOCKS_PER_CU");
        return e != nullptr ? std::atoi(e) : 0;
    }();
    int best = kMaxBlocks;
    if(forced >= 1 && forced <= kMaxBlocks) {
        best = forced;
    } else {
        double best_t = 1e300;
        for(int w = kMaxBlocks; w >= 1; --w) {
            const uint64_t per_simd = (static_cast<uint64_t>(n_pairs) + kSimds - 1) / kSimds;  // pairs on the busiest SIMD
            const uint64_t rounds = (per_simd + w - 1) / w;
            const double t = static_cast<double>(rounds) * w * kCost[w];
            if(t < best_t * 0.98) {
                best_t = t;
                best = w;
            }
        }
    }
    constexpr size_t kLdsPerCU = 160 * 1024, kStatic = kTabRows * kTabStride * sizeof(float);
    // LDS footprint per block that admits exactly `best` blocks per CU
    const size_t per_block = kLdsPerCU / best;
    const size_t dyn = best < 12 && per_block > kStatic + 512 ? per_block - kStatic - 512 : 0;
    return {kCUs * static_cast<uint32_t>(best), dyn};
}

bool device_is_gfx950(int dev) {
    hipDeviceProp_t prop;
    if(hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

}  // namespace

struct coati_hip_model {
    int device = 0;
    int gap_len = 1;
    GapConsts k{};
    float* d_table = nullptr;
    hipStream_t stream = nullptr;
};

struct coati_hip_batch {
    coati_hip_model* model = nullptr;
    uint64_t n_pairs = 0;
    uint64_t cells = 0;
    uint64_t ops_total = 0;    // sum(la + lb)
    uint64_t flag_dwords = 0;  // dwords in the bit-plane arena
    uint64_t bnd_floats = 0;
    uint64_t device_bytes = 0;
    std::vector<PairDesc> desc;
    // device
    PairDesc* d_desc = nullptr;
    uint32_t* d_order = nullptr;   // pair indices, most cells first
    uint32_t* d_queue = nullptr;   // ticket counter of the persistent fill kernel
    uint8_t *d_a = nullptr, *d_b = nullptr, *d_ops = nullptr;
    uint32_t* d_flags = nullptr;
    float *d_bnd = nullptr, *d_scores = nullptr;
    uint64_t* d_ops_start = nullptr;
    uint32_t* d_ops_len = nullptr;
    static constexpr int kTimingRing = 64;  // launches whose kernel times can still be read back
    hipEvent_t ev[kTimingRing][3] = {};
    uint64_t n_launches = 0;
    bool launched = false;
};

extern "C" {

uint32_t coati_hip_version(void) { return (0u << 16) | 1u; }

const char* coati_hip_last_error(void) { return g_error.c_str(); }

int coati_hip_device_count(void) {
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for(int d = 0; d < n; ++d) ok += device_is_gfx950(d) ? 1 : 0;
    return ok;
}

int coati_hip_model_create(const float* table, float no_gap, float gap_stop, float gap_open,
                           float gap_extend, int gap_len, int device, coati_hip_model_t** out) {
    if(out == nullptr) return fail(COATI_HIP_EINVAL, "model_create: out is NULL");
    *out = nullptr;
    if(table == nullptr) return fail(COATI_HIP_EINVAL, "model_create: table is NULL");
    if(gap_len < 1) return fail(COATI_HIP_EINVAL, "model_create: gap_len must be >= 1 (got %d)", gap_len);
    if(gap_len != 1)
        return fail(COATI_HIP_EINVAL, "model_create: gap_len %d not supported by this build (only 1)",
                    gap_len);
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(COATI_HIP_ENODEVICE, "model_create: no HIP device available");
    if(device < 0 || device >= n)
        return fail(COATI_HIP_EINVAL, "model_create: device %d out of range [0,%d)", device, n);
    if(!device_is_gfx950(device))
        return fail(COATI_HIP_ENODEVICE, "model_create: device %d is not gfx950 (MI355X)", device);
    auto* m = new(std::nothrow) coati_hip_model;
    if(m == nullptr) return fail(COATI_HIP_ENOMEM, "model_create: host allocation failed");
    m->device = device;
    m->gap_len = gap_len;
    m->k = GapConsts{no_gap, gap_stop, gap_open, gap_extend};
    auto cleanup = [&](int rc) {
        coati_hip_model_destroy(m);
        return rc;
    };
    hipError_t e;
    if((e = hipSetDevice(device)) != hipSuccess)
        return cleanup(fail(COATI_HIP_EHIP, "hipSetDevice: %s", hipGetErrorString(e)));
    if((e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking)) != hipSuccess)
        return cleanup(fail(COATI_HIP_EHIP, "hipStreamCreate: %s", hipGetErrorString(e)));
    const size_t bytes = sizeof(float) * kTabRows * kTabCols;
    if((e = hipMalloc(&m->d_table, bytes)) != hipSuccess)
        return cleanup(fail(COATI_HIP_ENOMEM, "hipMalloc(table): %s", hipGetErrorString(e)));
    if((e = hipMemcpy(m->d_table, table, bytes, hipMemcpyHostToDevice)) != hipSuccess)
        return cleanup(fail(COATI_HIP_EHIP, "hipMemcpy(table): %s", hipGetErrorString(e)));
    *out = m;
    return COATI_HIP_OK;
}

void coati_hip_model_destroy(coati_hip_model_t* m) {
    if(m == nullptr) return;
    (void)hipSetDevice(m->device);
    if(m->d_table != nullptr) (void)hipFree(m->d_table);
    if(m->stream != nullptr) (void)hipStreamDestroy(m->stream);
    delete m;
}

void coati_hip_batch_destroy(coati_hip_batch_t* b) {
    if(b == nullptr) return;
    if(b->model != nullptr) (void)hipSetDevice(b->model->device);
    void* ptrs[] = {b->d_order, b->d_queue, b->d_desc, b->d_a,      b->d_b,         b->d_ops,    b->d_flags,
                    b->d_bnd,  b->d_scores, b->d_ops_start, b->d_ops_len};
    for(void* p : ptrs)
        if(p != nullptr) (void)hipFree(p);
    for(auto& trio : b->ev)
        for(hipEvent_t e : trio)
            if(e != nullptr) (void)hipEventDestroy(e);
    delete b;
}

int coati_hip_batch_create(coati_hip_model_t* model, uint64_t n_pairs, const uint8_t* a_cat,
                           const uint64_t* a_off, const uint8_t* b_cat, const uint64_t* b_off,
                           coati_hip_batch_t** out) {
    if(out == nullptr) return fail(COATI_HIP_EINVAL, "batch_create: out is NULL");
    *out = nullptr;
    if(model == nullptr) return fail(COATI_HIP_EINVAL, "batch_create: model is NULL");
    if(a_off == nullptr || b_off == nullptr) return fail(COATI_HIP_EINVAL, "batch_create: offsets are NULL");
    if(n_pairs > 0xffffffffull) return fail(COATI_HIP_EINVAL, "batch_create: too many pairs");
    const uint64_t a_total = a_off[n_pairs] - a_off[0], b_total = b_off[n_pairs] - b_off[0];
    if((a_total > 0 && a_cat == nullptr) || (b_total > 0 && b_cat == nullptr))
        return fail(COATI_HIP_EINVAL, "batch_create: sequence data is NULL");

    auto* b = new(std::nothrow) coati_hip_batch;
    if(b == nullptr) return fail(COATI_HIP_ENOMEM, "batch_create: host allocation failed");
    b->model = model;
    b->n_pairs = n_pairs;
    auto cleanup = [&](int rc) {
        coati_hip_batch_destroy(b);
        return rc;
    };
    try {
        b->desc.resize(n_pairs);
    } catch(const std::bad_alloc&) {
        return cleanup(fail(COATI_HIP_ENOMEM, "batch_create: host allocation failed"));
    }
    const uint64_t L = static_cast<uint64_t>(model->gap_len);
    for(uint64_t p = 0; p < n_pairs; ++p) {
        if(a_off[p + 1] < a_off[p] || b_off[p + 1] < b_off[p])
            return cleanup(fail(COATI_HIP_EINVAL, "batch_create: offsets of pair %llu decrease",
                                static_cast<unsigned long long>(p)));
        const uint64_t la = a_off[p + 1] - a_off[p], lb = b_off[p + 1] - b_off[p];
        if(la > 0x7fffff00ull || lb > 0x7fffff00ull)
            return cleanup(fail(COATI_HIP_EINVAL, "batch_create: pair %llu too long",
                                static_cast<unsigned long long>(p)));
        // process_marginal, src/lib/utils.cc:822-835
        if(la % 3 != 0 || la % L != 0)
            return cleanup(fail(COATI_HIP_EINVAL,
                                "Length of reference sequence must be multiple of 3 and gap unit "
                                "length. (pair %llu)",
                                static_cast<unsigned long long>(p)));
        if(lb % L != 0)
            return cleanup(fail(COATI_HIP_EINVAL,
                                "Length of descendant sequence must be multiple of gap unit length. "
                                "(pair %llu)",
                                static_cast<unsigned long long>(p)));
        for(uint64_t q = a_off[p]; q < a_off[p + 1]; ++q)
            if(a_cat[q] >= kTabRows)
                return cleanup(fail(COATI_HIP_EINVAL, "batch_create: ancestor code %u out of range (pair %llu)",
                                    a_cat[q], static_cast<unsigned long long>(p)));
        for(uint64_t q = b_off[p]; q < b_off[p + 1]; ++q)
            if(b_cat[q] >= kTabCols)
                return cleanup(fail(COATI_HIP_EINVAL, "batch_create: descendant code %u out of range (pair %llu)",
                                    b_cat[q], static_cast<unsigned long long>(p)));
        PairDesc& d = b->desc[p];
        d.a_off = a_off[p] - a_off[0];
        d.b_off = b_off[p] - b_off[0];
        d.la = static_cast<uint32_t>(la);
        d.lb = static_cast<uint32_t>(lb);
        d.flags_off = b->flag_dwords;
        d.bnd_off = b->bnd_floats;
        d.ops_off = b->ops_total;
        const uint32_t ns = n_strips(d.lb);
        if(la > 0 && lb > 0) b->flag_dwords += ns * strip_dwords(d.la);
        // 128-byte aligned so that no two waves ever share a cache line of it
        if(ns > 1) b->bnd_floats += (2 * (la + 1) + 31) / 32 * 32;
        b->ops_total += la + lb;
        b->cells += la * lb;
    }

    if(hipSetDevice(model->device) != hipSuccess)
        return cleanup(fail(COATI_HIP_EHIP, "hipSetDevice failed"));
    auto dmalloc = [&](void** p, uint64_t bytes) -> hipError_t {
        if(bytes == 0) bytes = 16;
        b->device_bytes += bytes;
        return hipMalloc(p, bytes);
    };
#define B_TRY(expr)                                                                             \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if(e_ != hipSuccess)                                                                    \
            return cleanup(fail(e_ == hipErrorOutOfMemory ? COATI_HIP_ENOMEM : COATI_HIP_EHIP,  \
                                "%s failed: %s", #expr, hipGetErrorString(e_)));                \
    } while(0)
    B_TRY(dmalloc(reinterpret_cast<void**>(&b->d_desc), n_pairs * sizeof(PairDesc)));
    B_TRY(dmalloc(reinterpret_cast<void**>(&b->d_a), a_total));
    B_TRY(dmalloc(reinterpret_cast<void**>(&b->d_b), b_total));
    B_TRY(dmalloc(reinterpret_cast<void**>(&b->d_ops), b->ops_total));
    B_TRY(dmalloc(reinterpret_cast<void**>(&b->d_flags), b->flag_dwords * sizeof(uint32_t)));
    B_TRY(dmalloc(reinterpret_cast<void**>(&b->d_bnd), b->bnd_floats * sizeof(float)));
    B_TRY(dmalloc(reinterpret_cast<void**>(&b->d_scores), n_pairs * sizeof(float)));
    B_TRY(dmalloc(reinterpret_cast<void**>(&b->d_ops_start), n_pairs * sizeof(uint64_t)));
    B_TRY(dmalloc(reinterpret_cast<void**>(&b->d_ops_len), n_pairs * sizeof(uint32_t)));
    B_TRY(dmalloc(reinterpret_cast<void**>(&b->d_order), n_pairs * sizeof(uint32_t)));
    B_TRY(dmalloc(reinterpret_cast<void**>(&b->d_queue), sizeof(uint32_t)));
    if(n_pairs > 0) {
        B_TRY(hipMemcpy(b->d_desc, b->desc.data(), n_pairs * sizeof(PairDesc), hipMemcpyHostToDevice));
        // longest-processing-time-first order for the dynamic queue
        std::vector<uint32_t> order(n_pairs);
        for(uint64_t p = 0; p < n_pairs; ++p) order[p] = static_cast<uint32_t>(p);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            return static_cast<uint64_t>(b->desc[x].la) * b->desc[x].lb > static_cast<uint64_t>(b->desc[y].la) * b->desc[y].lb;
        });
        B_TRY(hipMemcpy(b->d_order, order.data(), n_pairs * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if(a_total > 0) B_TRY(hipMemcpy(b->d_a, a_cat + a_off[0], a_total, hipMemcpyHostToDevice));
    if(b_total > 0) B_TRY(hipMemcpy(b->d_b, b_cat + b_off[0], b_total, hipMemcpyHostToDevice));
    for(auto& trio : b->ev)
        for(auto& e : trio) B_TRY(hipEventCreate(&e));
#undef B_TRY
    *out = b;
    return COATI_HIP_OK;
}

uint64_t coati_hip_batch_device_bytes(const coati_hip_batch_t* b) { return b ? b->device_bytes : 0; }
uint64_t coati_hip_batch_cells(const coati_hip_batch_t* b) { return b ? b->cells : 0; }

int coati_hip_viterbi_launch(coati_hip_batch_t* b) {
    if(b == nullptr) return fail(COATI_HIP_EINVAL, "viterbi_launch: batch is NULL");
    coati_hip_model* m = b->model;
    HIP_TRY(hipSetDevice(m->device));
    const uint32_t n = static_cast<uint32_t>(b->n_pairs);
    hipEvent_t* ev = b->ev[b->n_launches % coati_hip_batch::kTimingRing];
    HIP_TRY(hipEventRecord(ev[0], m->stream));
    if(n > 0) {
        HIP_TRY(hipMemsetAsync(b->d_queue, 0, sizeof(uint32_t), m->stream));
        const FillShape shape = fill_launch_shape(n);
        if(shape.dynamic_lds > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(viterbi_l1),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shape.dynamic_lds)));
        hipLaunchKernelGGL(viterbi_l1, dim3(shape.grid), dim3(kFillWaves * kWave), shape.dynamic_lds, m->stream,
                           m->d_table, m->k, b->d_desc, b->d_order, n, b->d_queue, b->d_a, b->d_b, b->d_flags,
                           b->d_bnd, b->d_scores, b->d_ops, b->d_ops_start, b->d_ops_len);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(ev[1], m->stream));
    HIP_TRY(hipEventRecord(ev[2], m->stream));  // (the traceback is fused into the fill kernel)
    b->n_launches += 1;
    b->launched = true;
    return COATI_HIP_OK;
}

int coati_hip_batch_sync(coati_hip_batch_t* b) {
    if(b == nullptr) return fail(COATI_HIP_EINVAL, "batch_sync: batch is NULL");
    HIP_TRY(hipSetDevice(b->model->device));
    HIP_TRY(hipStreamSynchronize(b->model->stream));
    return COATI_HIP_OK;
}

int coati_hip_viterbi_fetch(coati_hip_batch_t* b, float* scores, uint8_t* ops, uint64_t ops_capacity,
                            uint64_t* ops_off, uint32_t* ops_len) {
    if(b == nullptr) return fail(COATI_HIP_EINVAL, "viterbi_fetch: batch is NULL");
    if(!b->launched) return fail(COATI_HIP_ESTATE, "viterbi_fetch: nothing was launched");
    if(ops != nullptr && ops_capacity < b->ops_total)
        return fail(COATI_HIP_EINVAL, "viterbi_fetch: ops_capacity %llu < %llu",
                    static_cast<unsigned long long>(ops_capacity),
                    static_cast<unsigned long long>(b->ops_total));
    int rc = coati_hip_batch_sync(b);
    if(rc != COATI_HIP_OK) return rc;
    const uint64_t n = b->n_pairs;
    if(n == 0) return COATI_HIP_OK;
    if(scores != nullptr) HIP_TRY(hipMemcpy(scores, b->d_scores, n * sizeof(float), hipMemcpyDeviceToHost));
    if(ops != nullptr && b->ops_total > 0) HIP_TRY(hipMemcpy(ops, b->d_ops, b->ops_total, hipMemcpyDeviceToHost));
    if(ops_off != nullptr)
        HIP_TRY(hipMemcpy(ops_off, b->d_ops_start, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if(ops_len != nullptr)
        HIP_TRY(hipMemcpy(ops_len, b->d_ops_len, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return COATI_HIP_OK;
}

int coati_hip_batch_result_ptrs(coati_hip_batch_t* b, void** scores, void** ops, uint64_t* ops_bytes,
                                void** ops_off, void** ops_len) {
    if(b == nullptr) return fail(COATI_HIP_EINVAL, "batch_result_ptrs: batch is NULL");
    if(scores != nullptr) *scores = b->d_scores;
    if(ops != nullptr) *ops = b->d_ops;
    if(ops_bytes != nullptr) *ops_bytes = b->ops_total;
    if(ops_off != nullptr) *ops_off = b->d_ops_start;
    if(ops_len != nullptr) *ops_len = b->d_ops_len;
    return COATI_HIP_OK;
}

int coati_hip_viterbi_timing(coati_hip_batch_t* b, uint32_t launches_back, float* fill_ms, float* walk_ms) {
    if(b == nullptr) return fail(COATI_HIP_EINVAL, "viterbi_timing: batch is NULL");
    if(!b->launched) return fail(COATI_HIP_ESTATE, "viterbi_timing: nothing was launched");
    if(launches_back >= coati_hip_batch::kTimingRing || launches_back >= b->n_launches)
        return fail(COATI_HIP_EINVAL, "viterbi_timing: launch %u back is not recorded", launches_back);
    int rc = coati_hip_batch_sync(b);
    if(rc != COATI_HIP_OK) return rc;
    hipEvent_t* ev = b->ev[(b->n_launches - 1 - launches_back) % coati_hip_batch::kTimingRing];
    float f = 0.f, w = 0.f;
    HIP_TRY(hipEventElapsedTime(&f, ev[0], ev[1]));
    HIP_TRY(hipEventElapsedTime(&w, ev[1], ev[2]));
    if(fill_ms != nullptr) *fill_ms = f;
    if(walk_ms != nullptr) *walk_ms = w;
    return COATI_HIP_OK;
}

int coati_hip_viterbi_last_timing(coati_hip_batch_t* b, float* fill_ms, float* walk_ms) {
    return coati_hip_viterbi_timing(b, 0, fill_ms, walk_ms);
}

int coati_hip_viterbi_batch(coati_hip_model_t* model, uint64_t n_pairs, const uint8_t* a_cat,
                            const uint64_t* a_off, const uint8_t* b_cat, const uint64_t* b_off,
                            float* scores, uint8_t* ops, uint64_t ops_capacity, uint64_t* ops_off,
                            uint32_t* ops_len) {
    if(model == nullptr) return fail(COATI_HIP_EINVAL, "viterbi_batch: model is NULL");
    if(a_off == nullptr || b_off == nullptr) return fail(COATI_HIP_EINVAL, "viterbi_batch: offsets are NULL");
    HIP_TRY(hipSetDevice(model->device));
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const uint64_t budget = static_cast<uint64_t>(free_b * 0.8);
    uint64_t ops_base = 0;  // slot start of the first pair of the current chunk
    uint64_t p0 = 0;
    while(p0 < n_pairs) {
        // grow the chunk until the workspace estimate exceeds the budget
        uint64_t p1 = p0, need = 0, chunk_ops = 0;
        while(p1 < n_pairs) {
            const uint64_t la = a_off[p1 + 1] - a_off[p1], lb = b_off[p1 + 1] - b_off[p1];
            const uint64_t w = (la > 0 && lb > 0) ? n_strips(static_cast<uint32_t>(lb)) * strip_dwords(static_cast<uint32_t>(la)) * 4 : 0;
            const uint64_t add = w + 3 * (la + lb) + 8 * (la + 1) + 128;
            if(p1 > p0 && need + add > budget) break;
            need += add;
            chunk_ops += la + lb;
            ++p1;
        }
        coati_hip_batch_t* b = nullptr;
        int rc = coati_hip_batch_create(model, p1 - p0, a_cat, a_off + p0, b_cat, b_off + p0, &b);
        if(rc != COATI_HIP_OK) return rc;
        rc = coati_hip_viterbi_launch(b);
        if(rc == COATI_HIP_OK)
            rc = coati_hip_viterbi_fetch(b, scores ? scores + p0 : nullptr, ops ? ops + ops_base : nullptr,
                                         ops ? ops_capacity - ops_base : 0, ops_off ? ops_off + p0 : nullptr,
                                         ops_len ? ops_len + p0 : nullptr);
        coati_hip_batch_destroy(b);
        if(rc != COATI_HIP_OK) return rc;
        if(ops_off != nullptr)
            for(uint64_t p = p0; p < p1; ++p) ops_off[p] += ops_base;
        ops_base += chunk_ops;
        p0 = p1;
    }
    return COATI_HIP_OK;
}

int coati_hip_debug_viterbi_flags(coati_hip_batch_t* b, uint64_t pair, uint8_t* out, uint64_t capacity) {
    if(b == nullptr || out == nullptr) return fail(COATI_HIP_EINVAL, "debug_viterbi_flags: NULL argument");
    if(!b->launched) return fail(COATI_HIP_ESTATE, "debug_viterbi_flags: nothing was launched");
    if(pair >= b->n_pairs) return fail(COATI_HIP_EINVAL, "debug_viterbi_flags: pair out of range");
    const uint64_t n = static_cast<uint64_t>(b->desc[pair].la) * b->desc[pair].lb;
    if(capacity < n) return fail(COATI_HIP_EINVAL, "debug_viterbi_flags: capacity too small");
    if(n == 0) return COATI_HIP_OK;
    int rc = coati_hip_batch_sync(b);
    if(rc != COATI_HIP_OK) return rc;
    uint8_t* d_out = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_out), n));
    hipLaunchKernelGGL(decode_flags, dim3(static_cast<uint32_t>(std::min<uint64_t>((n + 255) / 256, 4096))),
                       dim3(256), 0, b->model->stream, b->d_desc, static_cast<uint32_t>(pair), b->d_flags, d_out);
    hipError_t e = hipStreamSynchronize(b->model->stream);
    if(e == hipSuccess) e = hipMemcpy(out, d_out, n, hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if(e != hipSuccess) return fail(COATI_HIP_EHIP, "debug_viterbi_flags: %s", hipGetErrorString(e));
    return COATI_HIP_OK;
}

}  // extern "C"
