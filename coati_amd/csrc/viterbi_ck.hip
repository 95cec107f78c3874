// viterbi_ck: the Viterbi kernel for gap_len == 1 (the default and by far the common case).
// Persistent wavefronts, one strip of one sequence pair at a time per wavefront: a LEAN fill that
// keeps no per-cell traceback information, then the traceback by the same wavefront, which
// re-derives the decisions of the cells the path visits from checkpoints the fill left in HBM.
//
// What it replaces in the reference (all CPU, one pair per process):
//   forward_impl<tropical, align_pair_work_mem_t>   src/lib/align_pair.cc:62-139
//   traceback<tropical> / max_mdi / max_mi          src/lib/align_pair.cc:210-303
//
// Why (DESIGN.md §3.1 has the derivation and the measurements).  The reference's traceback
// re-derives each decision from the stored scores of the cell it arrives at
// (align_pair.cc:275-296): five comparisons per cell.  Evaluating them for EVERY cell in the fill
// (viterbi_l1.hip) costs 10 of its 25 VALU instructions per cell -- for cells of which ~0.2 % are
// ever visited.  Here the fill cell is the 15 instructions of the recurrence alone, and per
// wavefront step the fill stores
//   * what every lane RECEIVED from its left neighbour (diagonal X and left Z: 8 B per lane),
//   * every kR steps the lane state (X, Y of its W columns),
// which makes every (kR steps) x (one lane = W columns) tile of the matrix recomputable on its own.
// The traceback then works in rounds: the 64 lanes recompute, each on its own, the 64 tiles of a
// band around the predicted continuation of the path -- with the 25-instruction cell of
// viterbi_cell.hpp, i.e. the same adds in the same order, so every bit equals what a full fill would
// have stored -- and the wave-cooperative walker follows the path through those tiles until it
// leaves the band.  A 1 kb pair takes 3-4 rounds of 256 cells per lane: ~8 % of the fill's work.
//
// fp32 only, adds/max in the reference's evaluation order; -ffp-contract=off -fno-slp-vectorize.
#include "viterbi_cell.hpp"

#include <algorithm>
#include <cstdio>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

namespace coati_hip_detail {
#ifdef COATI_FILL_TRACE
// Debug build only (make trace): per-wave wall-clock stamps (s_memrealtime, 100 MHz) of the
// persistent loop, read back by tools/trace_fill.py through coati_hip_debug_trace.
__device__ unsigned long long g_ck_trace[4096 * 16];
__device__ unsigned long long g_ck_poll[4];  // sub-blocks of consumer strips, those whose boundary was not there when they began, polls made for them, -
#define COATI_CK_STAMP(slot)                                                                     \
    do {                                                                                         \
        if(lane_id == 0) /* (items from the seventh on share the last record: a wavefront's LAST stamps are always there) */ \
            g_ck_trace[trace_wave * 16 + (trace_n + (slot) < 15 ? trace_n + (slot) : 13 + (slot))] = __builtin_amdgcn_s_memrealtime(); \
    } while(0)
#else
#define COATI_CK_STAMP(slot) do { } while(0)
#endif
namespace {

// Register state of one lane of the lean fill: its W columns of the row it processed last.
template <int W>
struct CkLane {
    float X[W];  // max((M+ng)+ng, D+gs, (I+gs)+ng): feeds M of the next diagonal cell
    float Y[W];  // max((M+ng)+go, D+ge, (I+gs)+go): the D value of the cell below
    float xlast_old;  // X[W-1] of the row before: the right neighbour's diagonal input
    float zlast;      // max(M+go, I+ge) of column W-1: the right neighbour's I value
};

// The recurrence of one cell (align_pair.cc:97-124 with look_back = 1; the algebra is in
// viterbi_cell.hpp / DESIGN.md §2): 11 v_add_f32, the LDS address of this column's score for the
// next wavefront step, and the three maxima.
#define COATI_CELL_LEAN                                                                       \
    "v_add_f32 %[t0], %[diag], %[s]\n\t"       /* M  = diag + s                            */ \
    "v_add_f32 %[t1], %[ge], %[zl]\n\t"        /* z2 = I + ge                              */ \
    "v_add_f32 %[t2], %[gs], %[zl]\n\t"        /* i1 = I + gs                              */ \
    "v_add_f32 %[t3], %[go], %[t0]\n\t"        /* z1 = M + go                              */ \
    "v_add_f32 %[t0], %[ng], %[t0]\n\t"        /* m1 = M + ng                              */ \
    "v_add_u32 %[s], %[lds], %[boff]\n\t"      /* LDS address of next step's score         */ \
    "v_add_f32 %[t4], %[ng], %[t2]\n\t"        /* x3 = i1 + ng                             */ \
    "v_max_f32 %[zl], %[t3], %[t1]\n\t"        /* Z  = max(z1,z2) -> I of the next column  */ \
    "v_add_f32 %[t1], %[gs], %[y]\n\t"         /* x2 = D + gs                              */ \
    "v_add_f32 %[t3], %[ng], %[t0]\n\t"        /* x1 = m1 + ng                             */ \
    "v_add_f32 %[t2], %[go], %[t2]\n\t"        /* y3 = i1 + go                             */ \
    "v_max3_f32 %[x], %[t3], %[t1], %[t4]\n\t" /* X  = max(x1,x2,x3)                       */ \
    "v_add_f32 %[t1], %[ge], %[y]\n\t"         /* y2 = D + ge                              */ \
    "v_add_f32 %[t0], %[go], %[t0]\n\t"        /* y1 = m1 + go                             */ \
    "v_max3_f32 %[y], %[t0], %[t1], %[t2]"      /* Y  = max(y1,y2,y3)                       */

// (Five temporaries: the register-only replay of the cell, tools/ubench/gen_cell_pk.py, runs at the
// same rate whatever the order of the 15 instructions, so the order is the one with the shortest
// live ranges.  `s` is consumed by the first instruction and then carries the LDS address.)
template <int C, int W>
__device__ __forceinline__ void cell_lean(const GapVec& k, CkLane<W>& st, float& diag, float& zl, float& s,
                                          uint32_t lds_next_row, uint32_t boff) {
    float x_new, t0, t1, t2, t3, t4;
    asm volatile(COATI_CELL_LEAN
                 : [x] "=&v"(x_new), [y] "+v"(st.Y[C]), [zl] "+v"(zl), [s] "+v"(s), [t0] "=&v"(t0), [t1] "=&v"(t1),
                   [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4)
                 : [diag] "v"(diag), [lds] "v"(lds_next_row), [boff] "v"(boff), [ng] "v"(k.ng), [gs] "v"(k.gs),
                   [go] "v"(k.go), [ge] "v"(k.ge));
    diag = st.X[C];  // the next column's diagonal input is this column's previous-row X
    st.X[C] = x_new;
    // next step's score (s holds its LDS byte address now).  volatile: the read is issued HERE, a step
    // ahead of its use -- left to itself the compiler sinks all W of them to the top of the next step
    s = *reinterpret_cast<const volatile __attribute__((address_space(3))) float*>(__builtin_bit_cast(uint32_t, s));
}

template <int W, int... C>
__device__ __forceinline__ void row_lean(const GapVec& k, CkLane<W>& st, float diag, float zl, float (&s)[W],
                                         uint32_t lds_next_row, const uint32_t (&boff)[W],
                                         std::integer_sequence<int, C...>) {
    st.xlast_old = st.X[W - 1];
    (cell_lean<C, W>(k, st, diag, zl, s[C], lds_next_row, boff[C]), ...);
    st.zlast = zl;
}

// HBM accesses of the hot loop are raw buffer stores: the base sits in four SGPRs, the lane part of
// the address is ONE constant VGPR and the step part an SGPR offset -- no 64-bit VGPR pointers to
// keep (and spill) in a loop that needs every VGPR for the cells.
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// (num_records = 2 GiB: a lane whose offset register holds kCkDropOffset is OUT OF RANGE and its part of the store is
// discarded by the address unit -- how a lane that keeps no checkpoint for a band skips its stores without a branch)
constexpr uint32_t kCkDropOffset = 0x80000000u;
constexpr uint32_t kCkPrefetchAt = 10;  // the step of a 16-step sub-block at which a strip asks for its next sub-block's left boundary (ck_chunk)
// Cache policy of a raw buffer access (gfx940+: bit 0 = sc0, bit 1 = nt, bit 4 = sc1).  kAuxAgent = sc1: the store goes
// THROUGH the XCD's L2 to memory, the load is served from memory -- what an agent-scope atomic access is.  Used for
// everything a pair that is cut into row parts hands from one wavefront to another (round 5): with it a hand-over needs
// no release fence (on this chip: a write-back of the XCD's whole L2) and no acquire fence (an invalidate of it).
constexpr int kAuxPlain = 0, kAuxAgent = 16;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7ffffff0u, 0x00020000);
}

// BANDED CHECKPOINTS (round 3; the band's shape: round 4).  The checkpoints are 1.09 bytes per cell written to HBM of which
// the traceback reads ~6 %: the tiles along the path.  And the stores are what holds the clock down (the
// fill's loop holds 2.35 GHz without them, 1.9 GHz with them).  A lane therefore keeps the checkpoints of a band
// (kCkRows wavefront steps) only when the band's middle lies within `half` steps of its centre step:
//     kept(lane t, band c)  <=>  | c * kCkRows + kCkRows/2 - centre(t) | <= half
// Where the path can be: every alignment column is a diagonal step or a gap, so with delta = la - lb the path needs at
// least |delta| gap columns of one kind; a path whose gaps are (almost) all of that kind -- related sequences with a few
// indels -- lies between the diagonal through (0, 0) and the one through (la, lb): at column j its row is between
// j + min(0, delta) and j + max(0, delta).  So centre(t) = (middle column of the lane) + delta / 2 + t (the lane's
// skew), and half = band + |delta| / 2: the band setting (COATI_HIP_OPT_CK_BAND, default 64) is the slack for gaps of the
// OTHER kind, and a pair with one 300-base deletion keeps a wider band instead of being filled twice (round 3 centred
// the band on the straight line (0, 0) -> (la, lb) with a fixed half width: a bag in which a quarter of the pairs
// carry a 90-300 nt indel ran 26 % SLOWER than with everything kept; bench.py extra.band_sensitivity).
// The FILL is unchanged -- every cell is computed, scores are the same bits; only which recompute hints exist changes.
// A walk that asks for a tile that was not kept (it cannot know the decisions there) reports it, and the wavefront
// fills the pair again with band = kCkBandOff (everything kept) and walks again: exact, at twice the cost for that
// pair.  Single-strip pairs only; 0 = off.
__device__ __forceinline__ int32_t ck_lane_centre(uint32_t la, uint32_t lb, uint32_t w, uint32_t t, uint32_t col0 = 0) {
    // (col0: first column of the lane's strip -- strips after the first of a multi-strip pair)
    const int32_t delta = static_cast<int32_t>(la) - static_cast<int32_t>(lb);
    return static_cast<int32_t>(col0 + t * w + w / 2) + delta / 2 + static_cast<int32_t>(t);
}
// half width of the kept band of a pair: the setting plus half the length difference (wave-uniform)
__device__ __forceinline__ uint32_t ck_band_half(uint32_t band, uint32_t la, uint32_t lb) {
    if(band == kCkBandOff) return band;
    const uint32_t delta = la > lb ? la - lb : lb - la;
    return band + (delta + 1u) / 2u;
}
// the same for the strips of a MULTI-strip pair (round 4): a long pair's path wanders further from its diagonal than a 1 kb
// pair's, so the slack grows with the length (8 kb: 96 + 61 steps, 32 kb: 96 + 230)
__device__ __forceinline__ uint32_t ck_band_half_long(uint32_t band, uint32_t la, uint32_t lb) {
    if(band == kCkBandOff) return band;
    return ck_band_half(band, la, lb) + la / 128u;
}
__device__ __forceinline__ bool ck_tile_kept(uint32_t half, int32_t centre, int32_t c) {
    if(half == kCkBandOff) return true;
    const int64_t d = static_cast<int64_t>(c) * kCkRows + kCkRows / 2 - static_cast<int64_t>(centre);
    return (d < 0 ? -d : d) <= static_cast<int64_t>(half);
}
__device__ __forceinline__ uint32_t fbits(float x) { return __builtin_bit_cast(uint32_t, x); }

// LAYOUT of an item's checkpoints (wave-uniform; the fill and the walk evaluate it on the same values).  Tile-major (common.hpp)
// pays because a banded fill stores from ~9 lanes per step and a recompute reads whole lines.  A fill that keeps EVERYTHING
// (band off: the refill of a pair whose walk left the band, strips from 64 on, narrow strips, the debug export) stores from all
// 64 lanes: there the coalesced 512-byte row per step of rounds 2-4 is the better store (10 000 pairs with everything kept:
// 5.0 ms step-major, 6.6 ms tile-major), and the row parts of a cut pair store through the L2 (ck_step).  Both: step-major.
__device__ __forceinline__ bool ck_step_major(uint32_t band, bool through) { return through || band == kCkBandOff; }

// Read-only per-strip context of one wavefront.
struct CkCtx {
    GapConsts k;
    GapVec kv;  // the same four values in VGPRs, for the cells (viterbi_cell.hpp)
    uint32_t la, col0, nsteps, lds_tab;
    int lane;
    bool last_strip;
    // strip boundaries (wave-uniform), INTERLEAVED (round 6; viterbi_lp's layout): the pair a strip's lane 0 takes for body row r --
    // (X of row r - 1, Z of row r) of its left neighbour's last column -- is floats 2r, 2r + 1 of the boundary array: ONE aligned
    // 8-byte load per row, and what the neighbour's lane 63 makes at a step, (Z, X) of row r, is floats 2r + 1, 2r + 2: ONE store
    float* bnd_out;           // written by this strip
    const uint32_t* bnd_in;   // the left neighbour's, as bit patterns (strip > 0; kSub: loaded 16 rows at a time)
    rsrc_t bnd_rsrc;          // bnd_out behind a buffer descriptor
    bool first_strip;
    uint32_t band;   // banded checkpoints: kCkBandOff or the half width in steps (ck_band_half)
    int32_t centre;  // this lane's centre step
    bool step_major;  // layout of this item's checkpoints (ck_step_major)
};
// HBM windows of one 64-step chunk (rebased per chunk so that offsets stay far below 2^32)
struct CkChunkMem {
    rsrc_t colin;  // float2[band in chunk][lane][step in band] of this chunk's steps
    rsrc_t rowck;  // float4[band in chunk][lane][q]
};

// lane state -> row checkpoint of the band that starts at chunk step kb (state BEFORE that step)
template <int W, bool kThrough = false>
__device__ __forceinline__ void store_rowck(const CkChunkMem& mem, int lane, const CkLane<W>& st, uint32_t kb, bool keep, bool step_major) {
    constexpr int kAux = kThrough ? kAuxAgent : kAuxPlain;
    const uint32_t soff = (kb / kCkRows) * (ck_rowck_quads(W) * kWave * 16u);
    uint32_t voff = static_cast<uint32_t>(lane);
    asm volatile("" : "+v"(voff));  // (derived here, once per kCkRows steps: not another VGPR held across the hot loop)
    // (tile-major like colin: the W/2 quads of a lane are one line; step-major -- a cut pair, a fill that keeps everything --
    // they stay [q][lane]: common.hpp, ck_step_major)
    const bool sm = kThrough || step_major;  // (wave-uniform)
    const uint32_t quad_stride = sm ? kWave * 16u : 16u;
    voff = keep ? voff * (sm ? 16u : ck_rowck_quads(W) * 16u) : kCkDropOffset;
#pragma unroll
    for(int q = 0; q < W / 4; ++q) {
        const u32x4 x = {fbits(st.X[4 * q]), fbits(st.X[4 * q + 1]), fbits(st.X[4 * q + 2]), fbits(st.X[4 * q + 3])};
        const u32x4 y = {fbits(st.Y[4 * q]), fbits(st.Y[4 * q + 1]), fbits(st.Y[4 * q + 2]), fbits(st.Y[4 * q + 3])};
        __builtin_amdgcn_raw_buffer_store_b128(x, mem.rowck, voff, soff + q * quad_stride, kAux);
        __builtin_amdgcn_raw_buffer_store_b128(y, mem.rowck, voff, soff + (W / 4 + q) * quad_stride, kAux);
    }
}

// One wavefront step.  In chunk 0 lane l does its first row at step l: until then it computes on
// whatever it holds (never looked at), and at step l it takes the state of the margin row.  ONE
// instantiation serves every chunk: a separate start-up copy of the loop (as viterbi_l1.hip has)
// cost ~50 VGPRs here and with them the fourth wavefront per SIMD.
template <int W, bool kSub, bool kSingle, bool kThrough>
__device__ __forceinline__ void ck_step(const CkCtx& cx, const CkChunkMem& mem, CkLane<W>& st, uint32_t& arow,
                                        float (&s)[W], const uint32_t (&boff)[W], uint32_t kbase, uint32_t kk,
                                        uint32_t a_chunk, float bx, float bz, uint32_t colin_voff, uint32_t colin_band_soff, uint32_t colin_step_shift) {
    const GapConsts& k = cx.k;
    const int lane = cx.lane;
    const uint32_t kstep = kbase + kk;
    // (a scalar test first, marked unlikely: past the first chunk the step's usual way has no taken branch -- which costs a
    // wavefront that is alone on its SIMD ~130 cycles)
    // (kSub only: with four wavefronts per SIMD the single-strip fill measured 1 % slower that way)
    if((kSub ? __builtin_expect(kbase == 0, 0) : static_cast<long>(kbase == 0)) && kk == static_cast<uint32_t>(lane)) {
        // This lane starts now: state of the margin row (matrix row 0, align_pair.cc:88-90):
        // M = D = lowest, I = go + ge*float(j-1).
        uint32_t bj0 = cx.col0 + lane * W;
        asm volatile("" : "+v"(bj0));  // compute in place (hoisted out of the loop these 2W values get spilled)
#pragma unroll
        for(int c = 0; c < W; ++c) {
            const float im = k.go + k.ge * static_cast<float>(bj0 + c);
            const float i1 = im + k.gs;
            st.X[c] = i1 + k.ng;
            st.Y[c] = i1 + k.go;
        }
    }
    // ---- hand-off from the left neighbour (full exec)
    // (kSub: bx / bz hold the 16 rows of the current sub-block in lanes 0-15, else the chunk's 64 rows)
    const float diag = shift_in(st.xlast_old, read_lane(bx, kSub ? static_cast<int>(kk & 15u) : static_cast<int>(kk)));
    const float zl = shift_in(st.zlast, read_lane(bz, kSub ? static_cast<int>(kk & 15u) : static_cast<int>(kk)));
    const uint32_t arow_next = shift_in(arow, read_lane(a_chunk, kk));
    // ---- checkpoint: what this lane received, into the lane's 128-byte line of the band (tile-major: common.hpp; the
    // lanes that keep a band at this step are ~2 * half / (W + 1) neighbours, so a store is that many 8-byte pieces
    // which the L2 merges over the band's 16 steps)
    // (step-major items -- ck_step_major -- keep the lanes' 8 bytes of a step next to each other as in rounds 2-4,
    // float2[k][lane]: a row part of a cut pair, whose every store goes to memory on its own, and fills that keep everything)
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{fbits(diag), fbits(zl)}, mem.colin, colin_voff, colin_band_soff + (kk << colin_step_shift),
                                          kThrough ? kAuxAgent : kAuxPlain);
    // ---- the W cells (and the LDS gather for the next step)
    row_lean<W>(cx.kv, st, diag, zl, s, cx.lds_tab + arow_next, boff, std::make_integer_sequence<int, W>{});
    arow = arow_next;
    // lane 63 just did body row kstep - 63: its last column is the next strip's boundary
    // (kSingle -- the pair has one strip, known where the call is made: nothing here)
    // Round 6: ONE 8-byte write-through store per step, issued by every lane with no branch -- lanes other than 63, and every lane
    // while the row is not a body row, hold an offset out of the descriptor's range.  (Rounds 4-5: `if(lane == 63 && row in range)`
    // around two dword stores: a scalar branch, an EXEC branch and two memory instructions per step -- the trace build showed a
    // strip that PRODUCES a boundary 20 % slower than one that does not: 0.414 against 0.344 us per 8-column step.)
    if constexpr(!kSingle) {
        const uint32_t r = kstep - (kWave - 1);  // (wave-uniform; wraps while kstep < 63)
        const bool row_ok = !cx.last_strip && r < cx.la;
        uint32_t voff = static_cast<uint32_t>(lane);
        asm volatile("" : "+v"(voff));  // (derived per step: not another VGPR held across the hot loop)
        voff = (voff == static_cast<uint32_t>(kWave - 1) && row_ok) ? 0u : kCkDropOffset;
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{fbits(st.zlast), fbits(st.X[W - 1])}, cx.bnd_rsrc, voff, row_ok ? r * 8u + 4u : 0u, kAuxAgent);
    }
}

// Up to 64 wavefront steps in sub-blocks of kCkRows, each preceded by its row checkpoint.
// kSub (the strips of multi-strip pairs in the resident kernel, round 4): the strip's LEFT boundary arrives 16 rows at a
// time, as self-validating values -- the boundary arrays are filled with the NaN pattern 0xffffffff before the launch, the
// left neighbour stores its values write-through as it produces them (no per-chunk drain, no progress word), and this
// strip loads the 16 rows of its next sub-block past the L2 until none is the pattern.  A strip then follows its
// neighbour at 63 (the skew of the 64 lanes) + 16 + a round trip steps instead of 63 + 64 + a drain + a poll + an
// L2 invalidate (viterbi_l1 / viterbi_lp do the same).  Returns false if values never arrived.
template <int W, bool kSub, bool kSingle, bool kThrough>
__device__ __forceinline__ bool ck_chunk(const CkCtx& cx, const CkChunkMem& mem, CkLane<W>& st, uint32_t& arow,
                                         float (&s)[W], const uint32_t (&boff)[W], uint32_t kbase, uint32_t a_chunk,
                                         float bx, float bz, unsigned long long& pre) {
    const uint32_t kend = min(static_cast<uint32_t>(kWave), cx.nsteps - kbase);
    bool ok = true;
    for(uint32_t kb = 0; kb < kend; kb += kCkRows) {
        // (banded checkpoints: one comparison per lane and kCkRows steps; a lane outside the band stores nothing)
        const bool keep = ck_tile_kept(cx.band, cx.centre, static_cast<int32_t>((kbase + kb) / kCkRows));
        uint32_t colin_voff = static_cast<uint32_t>(cx.lane);
        asm volatile("" : "+v"(colin_voff));
        const bool sm = kThrough || cx.step_major;  // (wave-uniform)
        colin_voff = keep ? colin_voff * (sm ? 8u : kCkRows * 8u) : kCkDropOffset;
        // (scalar part of a step's address, tile-major: band kb / kCkRows of the chunk, then 8 bytes per step of the band;
        // step-major: 512 bytes per step)
        const uint32_t colin_band_soff = sm ? 0u : (kb >> kCkRowsLog2) * (kWave * kCkRows * 8u) - kb * 8u;
        const uint32_t colin_step_shift = sm ? 9u : 3u;
        store_rowck<W, kThrough>(mem, cx.lane, st, kb, keep, sm);
        const uint32_t ke = min(kb + kCkRows, kend);
        if constexpr(kSub) {
            // ---- a strip of a multi-strip pair.  Its wavefront is (nearly) alone on its SIMD: what it pays for is instructions --
            // ~4.5 cycles each whatever they are -- and TAKEN BRANCHES, ~130 cycles each (nobody hides the
            // instruction fetch).  Round 6: the sub-block's prologue has no taken branch on its usual way, a full sub-block is 16
            // steps of straight-line code (rounds 2-5: a loop of step pairs -- eight back-edges per sub-block, 8 % of a lone
            // wavefront's time), and the left boundary of the NEXT sub-block is asked for in the middle of this one (`pre`) and
            // looked at where that one begins: the round trip past the L2 is no longer part of every sub-block.
            static_assert(kCkRows == 16, "the boundary sub-blocks are the checkpoint bands");
            uint32_t row = kbase + kb + (static_cast<uint32_t>(cx.lane) & 15u);  // (lanes 16-63 repeat lanes 0-15)
            asm volatile("" : "+v"(row));
            // column 0 of the matrix (align_pair.cc:82-86): M(0,0)=0, D(i,0) margin -- computed by every strip, taken by the first
            const float margin_x = row == 0 ? (0.0f + cx.k.ng) + cx.k.ng : ((cx.k.ng + cx.k.go) + cx.k.ge * static_cast<float>(row - 1)) + cx.k.gs;
            uint32_t xb = static_cast<uint32_t>(pre), zb = static_cast<uint32_t>(pre >> 32);
            const bool have = cx.first_strip || row >= cx.la || (xb != 0xffffffffu && zb != 0xffffffffu);
#ifdef COATI_FILL_TRACE
            if(cx.lane == 0 && !cx.first_strip) atomicAdd(&g_ck_poll[0], 1ull);
#endif
            if(__builtin_expect(__builtin_amdgcn_ballot_w64(have) != ~0ull, 0)) {
                // not there yet (the strip's first sub-block; a neighbour that is slower than expected): ask until it is
#ifdef COATI_FILL_TRACE
                if(cx.lane == 0) atomicAdd(&g_ck_poll[1], 1ull);
#endif
                for(uint32_t spins = 0;; ++spins) {
#ifdef COATI_FILL_TRACE
                    if(cx.lane == 0) atomicAdd(&g_ck_poll[2], 1ull);
#endif
                    if(row < cx.la) {
                        const unsigned long long xz = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(cx.bnd_in) + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        xb = static_cast<uint32_t>(xz), zb = static_cast<uint32_t>(xz >> 32);
                    }
                    const bool valid = row >= cx.la || (xb != 0xffffffffu && zb != 0xffffffffu);
                    if(__builtin_amdgcn_ballot_w64(valid) == ~0ull) break;
                    if(spins > (1u << 24)) {
                        ok = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            bx = row < cx.la ? (cx.first_strip ? margin_x : __builtin_bit_cast(float, xb)) : kLowest;
            bz = (row < cx.la && !cx.first_strip) ? __builtin_bit_cast(float, zb) : kLowest;
            asm volatile("" : "+v"(bx), "+v"(bz));
            pre = ~0ull;
            // (16 columns per lane keep the loop of step pairs: unrolled they do not fit 128 VGPRs, and plans with such strips
            // have several wavefronts per SIMD, which hide a branch)
            if(W <= 8 && __builtin_expect(ke - kb == kCkRows, 1)) {
#pragma unroll
                for(uint32_t u = 0; u < kCkRows; ++u) {
                    if(u == kCkPrefetchAt && !cx.first_strip && row + kCkRows < cx.la)  // (W = 16: `pre` stays "not there", the sub-block asks when it begins)
                        pre = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(cx.bnd_in) + row + kCkRows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ck_step<W, kSub, kSingle, kThrough>(cx, mem, st, arow, s, boff, kbase, kb + u, a_chunk, bx, bz, colin_voff, colin_band_soff, colin_step_shift);
                }
                continue;
            }
        }
        // two steps per iteration: the new X of a column must not overwrite the old one before the
        // next column has taken it as its diagonal input; with two copies of the body the register
        // allocator ping-pongs X between two register sets instead of copying W values per step
        uint32_t kk = kb;
        for(; kk + 1 < ke; kk += 2) {
            ck_step<W, kSub, kSingle, kThrough>(cx, mem, st, arow, s, boff, kbase, kk, a_chunk, bx, bz, colin_voff, colin_band_soff, colin_step_shift);
            ck_step<W, kSub, kSingle, kThrough>(cx, mem, st, arow, s, boff, kbase, kk + 1, a_chunk, bx, bz, colin_voff, colin_band_soff, colin_step_shift);
        }
        if(kk < ke) ck_step<W, kSub, kSingle, kThrough>(cx, mem, st, arow, s, boff, kbase, kk, a_chunk, bx, bz, colin_voff, colin_band_soff, colin_step_shift);
    }
    return ok;
}

// Checkpoints of strip `strip` of a pair (dwords from the start of the PAIR's checkpoint area, which is
// PairDesc::flags_off in the arena or the slot of the wavefront that processes the pair)
__device__ __forceinline__ uint64_t ck_strip_base(const PairDesc& pd, uint32_t strip) {
    return strip * ck_strip_dwords(pd.la, pd.v_wmain);
}

// The chunks of a streamed call are not read by the host before they go up (reading every sequence byte once
// from DRAM was most of the host's planning time): the fill checks the codes it loads anyway and reports the
// first it finds out of range -- one 64-bit word in page-locked host memory: bit 63 set, bit 40 = descendant
// (else ancestor), bits 32-39 the code, bits 0-31 the chunk's pair.  (A code out of range cannot fault: it only
// indexes the LDS table, where a read past the allocation returns 0; the pair's result is garbage and the call
// fails.)  One plain store: whoever is last wins, every candidate is a true report.
__device__ __forceinline__ void ck_report_bad(unsigned long long* bad, uint32_t pair, uint32_t code, bool descendant) {
    const unsigned long long word = (1ull << 63) | (descendant ? 1ull << 40 : 0ull) | (static_cast<unsigned long long>(code & 0xffu) << 32) | pair;
    __hip_atomic_store(bad, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One work item: one strip (64*W descendant columns) of one pair, all its rows.  Returns false if
// the left neighbour's boundary column did not arrive within the spin bound.
template <int W, bool kSub = false, bool kSingle = false, bool kThrough = false>
__device__ __forceinline__ bool ck_fill_strip(const GapConsts& k, const PairDesc& pd, uint32_t pair, uint32_t strip,
                                              uint32_t ticket, int lane, uint32_t lds_tab, const char* tab_bytes,
                                              const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                              uint32_t* __restrict__ ck, float* __restrict__ bnd,
                                              float* __restrict__ scores, uint32_t* __restrict__ progress,
                                              uint32_t kbegin = 0, uint32_t kend = 0xffffffffu,
                                              unsigned long long* bad = nullptr, uint32_t band = kCkBandOff,
                                              bool defer_complete = false /* (not the last strip) the caller says "complete": ck_strip_complete; a fused pair */,
                                              bool own_left = false /* (!kSub, strip > 0) the left neighbour's boundary column was written by THIS wavefront, all of it (a fused pair): nothing to wait for */) {
    // [kbegin, kend): the steps of this item -- the whole strip, or one ROW PART of it (PairDesc::v_parts; whole
    // 64-step chunks).  A part that does not start at 0 takes over the lane state its predecessor left behind the
    // strip's checkpoints; one that does not end at the last step leaves it there.
    const uint32_t la = pd.la, lb = pd.lb;
    const uint32_t col0 = strip * (kWave * pd.v_wmain);  // every strip before this one has the main width
    const uint32_t ncol = min(static_cast<uint32_t>(kWave * W), lb - col0);
    const uint32_t nlanes = (ncol + W - 1) / W;
    const uint32_t nsteps = la + nlanes - 1;
    const bool last_strip = strip + 1 == pd.v_strips;
    uint32_t* __restrict__ ck_strip = ck + ck_strip_base(pd, strip);
    // strip-boundary columns, one array per strip boundary: [0, la] = X of the strip's last
    // column (index r = X of body row r-1; index 0 = margin row), [la+1, 2la] = Z of body row r.
    const uint64_t bstride = 2 * (static_cast<uint64_t>(la) + 1);
    float* __restrict__ bnd_out = bnd + pd.bnd_off + strip * bstride;  // written by this strip (interleaved: CkCtx)
    const float* __restrict__ bnd_in = bnd + pd.bnd_off + (strip - 1) * bstride;  // read by it (strip > 0)
    bool handoff_ok = true;

    uint32_t boff[W];  // byte offsets of this lane's W table columns
    {
        uint32_t worst = 0;
#pragma unroll
        for(int c = 0; c < W; ++c) {
            const uint32_t bj = col0 + lane * W + c;
            const uint32_t code = bj < lb ? static_cast<uint32_t>(b[bj]) : 0u;
            worst = max(worst, code);
            boff[c] = code * 4u;
        }
        if(bad != nullptr && worst >= static_cast<uint32_t>(kTabCols)) ck_report_bad(bad, pair, worst, true);
        if(bad != nullptr && kbegin == 0 && lane == 0 && a[0] >= kTabRows) ck_report_bad(bad, pair, a[0], false);
    }
    const CkCtx cx{k, gap_vec(k), la, col0, nsteps, lds_tab, lane, last_strip, bnd_out, reinterpret_cast<const uint32_t*>(bnd_in),
                   make_rsrc(bnd_out), strip == 0, band,
                   band == kCkBandOff ? 0 : ck_lane_centre(la, lb, W, static_cast<uint32_t>(lane), col0), ck_step_major(band, kThrough)};
    const uint32_t* __restrict__ rowck_strip = ck_strip + ck_colin_dwords(la);
    // the state of the margin row (matrix row 0, align_pair.cc:88-90: M = D = lowest, I = go +
    // ge*float(j-1)); a lane takes it again at its first step (ck_step)
    kend = min(kend, nsteps);
    uint32_t* __restrict__ part_state = ck_strip + ck_strip_dwords(la, W);  // [3][64]: xlast_old, zlast, table row
    CkLane<W> st;
    uint32_t arow;
    if(kbegin != 0) {
        // a continuation: X, Y are the row checkpoint of the band that starts here (the predecessor wrote it early)
        const uint32_t* __restrict__ rq_words = rowck_strip + static_cast<uint64_t>(kbegin / kCkRows) * (2 * W * kWave);
        if constexpr(kThrough) {
            // (what another wavefront -- another XCD, possibly -- stored THROUGH its L2: read past this one's)
            const rsrc_t rq = make_rsrc(rq_words);
#pragma unroll
            for(int q = 0; q < W / 4; ++q) {
                const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rq, static_cast<uint32_t>(lane) * 16u, q * (kWave * 16u), kAuxAgent);
                const u32x4 y = __builtin_amdgcn_raw_buffer_load_b128(rq, static_cast<uint32_t>(lane) * 16u, (W / 4 + q) * (kWave * 16u), kAuxAgent);
                // (through scalar copies: hipcc 7.2 compiles `__builtin_bit_cast(float, x[e])` on an ext_vector element into
                // ONE dword load splatted over the four -- DESIGN.md 5.24; found in the ISA, not by a test)
#pragma unroll
                for(int e = 0; e < 4; ++e) {
                    const uint32_t xw = x[e], yw = y[e];
                    st.X[4 * q + e] = __builtin_bit_cast(float, xw);
                    st.Y[4 * q + e] = __builtin_bit_cast(float, yw);
                }
            }
            st.xlast_old = __builtin_bit_cast(float, __hip_atomic_load(part_state + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            st.zlast = __builtin_bit_cast(float, __hip_atomic_load(part_state + kWave + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            arow = __hip_atomic_load(part_state + 2 * kWave + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const float4* __restrict__ rq = reinterpret_cast<const float4*>(rq_words);
#pragma unroll
            for(int q = 0; q < W / 4; ++q) {
                const float4 x = cx.step_major ? rq[q * kWave + lane] : rq[lane * ck_rowck_quads(W) + q];
                const float4 y = cx.step_major ? rq[(W / 4 + q) * kWave + lane] : rq[lane * ck_rowck_quads(W) + W / 4 + q];
                st.X[4 * q] = x.x, st.X[4 * q + 1] = x.y, st.X[4 * q + 2] = x.z, st.X[4 * q + 3] = x.w;
                st.Y[4 * q] = y.x, st.Y[4 * q + 1] = y.y, st.Y[4 * q + 2] = y.z, st.Y[4 * q + 3] = y.w;
            }
            st.xlast_old = __builtin_bit_cast(float, part_state[lane]);
            st.zlast = __builtin_bit_cast(float, part_state[kWave + lane]);
            arow = part_state[2 * kWave + lane];
        }
    } else {
        const uint32_t bj0 = col0 + lane * W;
#pragma unroll
        for(int c = 0; c < W; ++c) {
            const float im = k.go + k.ge * static_cast<float>(bj0 + c);
            const float i1 = im + k.gs;
            st.X[c] = i1 + k.ng;
            st.Y[c] = i1 + k.go;
        }
        if(!last_strip && lane == kWave - 1) store_through(&bnd_out[0], st.X[W - 1]);  // (X of the margin row: row 0's diagonal input)
        st.xlast_old = 0.0f;
        st.zlast = 0.0f;
        // table-row byte offset of the row this lane processes at the CURRENT step (every lane's first row is body row 0)
        arow = static_cast<uint32_t>(a[0]) * (kTabStride * 4u);
    }
    // the W substitution scores of that row, gathered one step ahead
    float s[W];
#pragma unroll
    for(int c = 0; c < W; ++c) s[c] = *reinterpret_cast<const float*>(tab_bytes + arow + boff[c]);

    unsigned long long pre_bnd = ~0ull;  // (kSub) the left boundary of the next sub-block, asked for ahead (ck_chunk)
    for(uint32_t kbase = kbegin; kbase < kend; kbase += kWave) {
        // ---- per-64-step chunk: lane l fetches what lane 0 will need at step kbase+l (boundary
        // column) and at step kbase+l+1 (ancestor code: gathered a step ahead)
        const uint32_t crow = kbase + lane;
        uint32_t a_chunk = 0;
        float bx = kLowest, bz = kLowest;
        if(crow + 1 < la) {
            const uint32_t code = a[crow + 1];
            if(bad != nullptr && code >= static_cast<uint32_t>(kTabRows)) ck_report_bad(bad, pair, code, false);  // (streamed chunks)
            a_chunk = code * (kTabStride * 4u);
        }
        if(!kSub && crow < la && strip == 0) {
            // column 0 of the matrix (align_pair.cc:82-86): M(0,0)=0, D(i,0) margin
            if(crow == 0) {
                bx = (0.0f + k.ng) + k.ng;
            } else {
                const float dm = (k.ng + k.go) + k.ge * static_cast<float>(crow - 1);
                bx = dm + k.gs;
            }
        }
        if(!kSub && strip > 0) {
            // rows kbase .. kbase+63 of the left neighbour's last column must be published
            if(!own_left) {
                handoff_ok = handoff_ok && wait_progress(progress + ticket - 1, min(la, kbase + kWave));
                if(__hip_atomic_load(progress + ticket - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kHandoffPoison) handoff_ok = false;
            }
            if(crow < la) {
                bx = bnd_in[2 * crow];
                bz = bnd_in[2 * crow + 1];
            }
        }
        // Consume the chunk loads HERE (one wait per 64 steps), not inside the step loop.
        asm volatile("" : "+v"(a_chunk), "+v"(bx), "+v"(bz));
        const CkChunkMem mem{make_rsrc(ck_strip + static_cast<uint64_t>(kbase) * (2 * kWave)),
                             make_rsrc(rowck_strip + static_cast<uint64_t>(kbase / kCkRows) * (2 * W * kWave))};
        handoff_ok = ck_chunk<W, kSub, kSingle, kThrough>(cx, mem, st, arow, s, boff, kbase, a_chunk, bx, bz, pre_bnd) && handoff_ok;
        if(!kSub && !last_strip && !defer_complete) {  // (kSub: the boundary values validate themselves: no per-chunk drain, no progress word; a fused pair's first strip: nobody is waiting)
            const uint32_t done = min(kbase + kWave, nsteps);
            if(done > kWave - 1 && done - (kWave - 1) < la) publish_progress(progress + ticket, done - (kWave - 1), lane == kWave - 1);
        }
    }
    if(kend < nsteps) {
        // a row part ends here: the lane state for whoever continues (X, Y as the row checkpoint of the next band --
        // the continuation stores the same values there again --, the rest behind the strip's checkpoints)
        const CkChunkMem next{make_rsrc(ck_strip), make_rsrc(rowck_strip + static_cast<uint64_t>(kend / kCkRows) * (2 * W * kWave))};
        store_rowck<W, kThrough>(next, lane, st, 0, true, cx.step_major);
        if constexpr(kThrough) {
            __hip_atomic_store(part_state + lane, fbits(st.xlast_old), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(part_state + kWave + lane, fbits(st.zlast), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(part_state + 2 * kWave + lane, arow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            part_state[lane] = fbits(st.xlast_old);
            part_state[kWave + lane] = fbits(st.zlast);
            part_state[2 * kWave + lane] = arow;
        }
        return handoff_ok;
    }
    // score = max(M,D,I) of the terminal-adjusted last cell (align_pair.cc:130-138,265) = X of the
    // last body cell, held by the lane that owns the last column after the strip's last step
    if(last_strip && lane == static_cast<int>((lb - 1 - col0) / W)) {
        const int last_c = static_cast<int>((lb - 1 - col0) % W);
        float sc = st.X[0];
#pragma unroll
        for(int c = 1; c < W; ++c) sc = (c == last_c) ? st.X[c] : sc;
        scores[pair] = sc;
    }
    // (wave-uniform; spliced traceback: the speculative walk of the strip comes first; the pair's last strip waits for the chain
    // where its walk leaves the strip -- CkSplice::chain)
    if((kSub || !kSingle) && defer_complete && !last_strip) return handoff_ok;
    if(kSub && strip > 0 && !defer_complete) {
        // (the chain "every earlier strip has released its checkpoints" still runs through the progress words: this strip
        // says "complete" only after its left neighbour has)
        handoff_ok = wait_progress(progress + ticket - 1, la) && handoff_ok;
        if(__hip_atomic_load(progress + ticket - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kHandoffPoison) handoff_ok = false;
    }
    if(!last_strip) {
        // The pair's traceback runs on the wavefront of the LAST strip: release this strip's
        // (plainly stored) checkpoints before saying "complete".
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        // (a strip whose own input never arrived publishes the poison value: every later strip of the pair,
        // down to the one that writes the result, then knows)
        publish_progress(progress + ticket, handoff_ok ? la : kHandoffPoison, lane == kWave - 1);
    }
    return handoff_ok;
}

// What ck_fill_strip<W, true> does last for a strip that is not its pair's last, as a call of its own (defer_complete): between
// the two the strip's wavefront walks its strip speculatively (ck_walk_pair, mode 1) -- the record is released with the checkpoints.
__device__ __forceinline__ bool ck_strip_complete(const PairDesc& pd, uint32_t strip, uint32_t ticket, int lane, uint32_t* __restrict__ progress,
                                                  bool handoff_ok) {
    if(strip > 0) {
        handoff_ok = wait_progress(progress + ticket - 1, pd.la) && handoff_ok;
        if(__hip_atomic_load(progress + ticket - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kHandoffPoison) handoff_ok = false;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    publish_progress(progress + ticket, handoff_ok ? pd.la : kHandoffPoison, lane == kWave - 1);
    return handoff_ok;
}

// ---------------------------------------------------------------------------------------------
// Traceback: tile recompute + walker
// ---------------------------------------------------------------------------------------------
// One strip of a pair as the traceback sees it (wave-uniform).
struct CkStrip {
    uint32_t strip, w, lg, col0, nlanes;
    const float2* colin;
    const float4* rowck;
};
__device__ __forceinline__ CkStrip ck_strip_of(const PairDesc& pd, const uint32_t* __restrict__ ck, uint32_t bj) {
    const uint32_t full = kWave * pd.v_wmain;
    uint32_t strip = bj / full, w = pd.v_wmain;
    if(strip + 1 >= pd.v_strips) {
        strip = pd.v_strips - 1;
        w = pd.v_wlast;
    }
    const uint32_t col0 = strip * full;
    const uint32_t ncol = min(kWave * w, pd.lb - col0);
    const uint32_t* base = ck + ck_strip_base(pd, strip);
    return {strip, w, 31u - static_cast<uint32_t>(__clz(static_cast<int>(w))), col0, (ncol + w - 1) / w,
            reinterpret_cast<const float2*>(base), reinterpret_cast<const float4*>(base + ck_colin_dwords(pd.la))};
}

// The 64 tiles of one round: a band around the predicted continuation of the path from the cell
// whose decision is pending, body coordinates (bi, bj).  A tile is (band c of kCkRows wavefront
// steps, lane t) of the strip; lane t of band c holds body rows c*kCkRows - t + [0, kCkRows) of
// columns col0 + t*W + [0, W).  Going back along the path a match moves up and left, a deletion
// up, an insertion left, so the prediction depends on the kind of run the walk is in:
//   mode D (after a match):     tile columns t0, t0-1, ... t0-20, three bands each around the
//                               diagonal through (bi, bj)
//   mode V (after a deletion):  bands c0, c0-1, ... c0-31 of tile columns t0 and t0-1
//   mode H (after an insertion): tile columns t0 ... t0-31, the band of row bi and the one above
// A wrong prediction costs a round, never correctness: a cell outside the set reads "unknown",
// the walk stops there and the next round is built around it.
struct TileSet {
    int mode;  // COATI_HIP_OP_MATCH / _DEL / _INS
    uint32_t strip, w, lg, col0, t0;
    int32_t bi, bj, c0;
    unsigned long long computed;  // slots whose tile was recomputed this round (a tile outside the kept band was not: its scratch is stale)
};
__device__ __forceinline__ int32_t tileset_cmid(const TileSet& ts, uint32_t dt) {
    const int32_t t = static_cast<int32_t>(ts.t0 - dt);
    const int32_t jr = min(ts.bj, static_cast<int32_t>(ts.col0 + ts.w * (t + 1)) - 1);  // right edge of the tile column
    const int32_t ir = ts.bi - (ts.bj - jr);                                            // the diagonal's row there
    return (ir + t - static_cast<int32_t>(ts.w / 2)) >> kCkRowsLog2;                    // band of the middle of its span
}
// slot -> tile (t, c); false if the slot is unused
__device__ __forceinline__ bool tileset_tile(const TileSet& ts, uint32_t slot, int32_t& t, int32_t& c) {
    if(ts.mode == COATI_HIP_OP_MATCH) {
        const uint32_t dt = slot / 3u;
        if(dt > 20u || dt > ts.t0) return false;
        t = static_cast<int32_t>(ts.t0 - dt);
        c = tileset_cmid(ts, dt) + static_cast<int32_t>(slot % 3u) - 1;
    } else if(ts.mode == COATI_HIP_OP_DEL) {
        const uint32_t o = slot & 1u;
        if(o > ts.t0) return false;
        t = static_cast<int32_t>(ts.t0 - o);
        c = ts.c0 - static_cast<int32_t>(slot >> 1);
    } else {
        const uint32_t dt = slot >> 1;
        if(dt > ts.t0) return false;
        t = static_cast<int32_t>(ts.t0 - dt);
        c = ((ts.bi + t) >> kCkRowsLog2) - static_cast<int32_t>(slot & 1u);
    }
    return c >= 0;
}
// tile (t, c) -> slot, or -1
__device__ __forceinline__ int tileset_slot(const TileSet& ts, uint32_t t, int32_t c) {
    if(t > ts.t0) return -1;
    const uint32_t dt = ts.t0 - t;
    if(ts.mode == COATI_HIP_OP_MATCH) {
        if(dt > 20u) return -1;
        const int32_t o = c - tileset_cmid(ts, dt);
        return (o >= -1 && o <= 1) ? static_cast<int>(3u * dt) + o + 1 : -1;
    }
    if(ts.mode == COATI_HIP_OP_DEL) {
        const int32_t u = ts.c0 - c;
        return (dt <= 1u && u >= 0 && u < 32) ? 2 * u + static_cast<int>(dt) : -1;
    }
    if(dt > 31u) return -1;
    const int32_t o = ((ts.bi + static_cast<int32_t>(t)) >> kCkRowsLog2) - c;
    return (o >= 0 && o <= 1) ? static_cast<int>(2u * dt) + o : -1;
}

// Per-wavefront scratch for the decision bits of one round: [which 0..2][step in band][slot], one
// dword each (A = (M1,M2) pairs, B = (D1,D2) pairs, C = IM; viterbi_cell.hpp).  After the W cells of
// a step the accumulators hold column cc's pair at bits 2(W-1-cc)+1, 2(W-1-cc) and its IM bit at
// bit W-1-cc (older steps sit above and are ignored).
constexpr uint32_t kCkScratchDwords = 3u * kCkRows * kWave;

// Recompute one tile per lane: lane state from the row checkpoint (or the margin row if the lane
// started inside the band), then kCkRows wavefront steps of W cells with the received values of
// the fill as left inputs, depositing the five decision bits of every cell.
template <int W>
__device__ __forceinline__ bool ck_recompute(const GapConsts& k, const PairDesc& pd, const CkStrip& sp, bool valid,
                                             int32_t t, int32_t c, uint32_t lds_tab, const char* tab_bytes,
                                             const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                             uint32_t* __restrict__ bits /* wave scratch + lane */, uint32_t band = kCkBandOff,
                                             bool through = false /* the checkpoints were stored through other wavefronts' L2s (a cut pair): read past this one's */) {
    const int32_t la = static_cast<int32_t>(pd.la);
    const int32_t k0 = c * static_cast<int32_t>(kCkRows);
    // rows this tile covers: k0 - t + [0, kCkRows)
    valid = valid && t >= 0 && t < static_cast<int32_t>(sp.nlanes) && k0 - t + static_cast<int32_t>(kCkRows) > 0 && k0 - t < la;
    // (banded checkpoints: a tile the fill kept no checkpoints for cannot be recomputed)
    if(band != kCkBandOff) valid = valid && ck_tile_kept(band, ck_lane_centre(pd.la, pd.lb, W, static_cast<uint32_t>(max(t, 0)), sp.col0), c);
    const bool computed = valid;
    if(!valid) {
        t = 0;
        c = 0;
    }
    uint32_t boff[W];
#pragma unroll
    for(int cc = 0; cc < W; ++cc) {
        const uint32_t bj = sp.col0 + static_cast<uint32_t>(t) * W + cc;
        boff[cc] = (valid && bj < pd.lb) ? static_cast<uint32_t>(b[bj]) * 4u : 0u;
    }
    // the layout the fill of this strip used (ck_step_major); step-major values are loaded with buffer loads off the strip's
    // base -- past the L2 when another wavefront stored them through its own (a cut pair) -- in 8- and 16-byte pieces
    const bool sm = ck_step_major(band, through);
    auto load_b128 = [&](const rsrc_t& rr, uint32_t voff, uint32_t soff) {
        return through ? __builtin_amdgcn_raw_buffer_load_b128(rr, voff, soff, kAuxAgent) : __builtin_amdgcn_raw_buffer_load_b128(rr, voff, soff, kAuxPlain);
    };
    LaneState<W> st;
    if(valid && t < k0) {
        const float4* rk = sp.rowck + (static_cast<uint64_t>(c) * ck_rowck_quads(W)) * kWave + t * static_cast<int32_t>(ck_rowck_quads(W));
#pragma unroll
        for(int q = 0; q < W / 4; ++q) {
            float4 x, y;
            if(sm) {  // (wave-uniform; the scalar copies: ck_fill_strip)
                const rsrc_t rr = make_rsrc(sp.rowck);
                const uint32_t voff = static_cast<uint32_t>((static_cast<uint64_t>(c) * ck_rowck_quads(W) * kWave + static_cast<uint32_t>(t)) * 16u);
                const u32x4 xv = load_b128(rr, voff, q * (kWave * 16u));
                const u32x4 yv = load_b128(rr, voff, (W / 4 + q) * (kWave * 16u));
                const uint32_t x0 = xv[0], x1 = xv[1], x2 = xv[2], x3 = xv[3], y0 = yv[0], y1 = yv[1], y2 = yv[2], y3 = yv[3];
                x = make_float4(__builtin_bit_cast(float, x0), __builtin_bit_cast(float, x1), __builtin_bit_cast(float, x2), __builtin_bit_cast(float, x3));
                y = make_float4(__builtin_bit_cast(float, y0), __builtin_bit_cast(float, y1), __builtin_bit_cast(float, y2), __builtin_bit_cast(float, y3));
            } else {
                x = rk[q], y = rk[W / 4 + q];
            }
            st.X[4 * q] = x.x, st.X[4 * q + 1] = x.y, st.X[4 * q + 2] = x.z, st.X[4 * q + 3] = x.w;
            st.Y[4 * q] = y.x, st.Y[4 * q + 1] = y.y, st.Y[4 * q + 2] = y.z, st.Y[4 * q + 3] = y.w;
        }
    } else {
        // the lane starts inside this band: margin-row state (align_pair.cc:88-90), as in ck_step
        const uint32_t bj0 = sp.col0 + static_cast<uint32_t>(t) * W;
#pragma unroll
        for(int cc = 0; cc < W; ++cc) {
            const float im = k.go + k.ge * static_cast<float>(bj0 + cc);
            const float i1 = im + k.gs;
            st.X[cc] = i1 + k.ng;
            st.Y[cc] = i1 + k.go;
        }
    }
#pragma unroll
    for(int p = 0; p < kAccs; ++p) st.acc[p] = 0u;
    st.xlast_old = 0.0f;
    st.zlast = 0.0f;
    // scores of the first row this lane will do, then (inside a step) those of the next one
    auto row_code = [&](int32_t r) { return static_cast<uint32_t>(a[min(max(r, 0), la - 1)]) * (kTabStride * 4u); };
    uint32_t arow = row_code(k0 - t);
    float s[W];
#pragma unroll
    for(int cc = 0; cc < W; ++cc) s[cc] = *reinterpret_cast<const float*>(tab_bytes + arow + boff[cc]);
    // the tile's line -- or the lane's column of the step-major rows (ck_step), one 8-byte load per step (byte offset from
    // the strip's colin: below 2 GiB up to la = 4 M rows)
    const float2* cin = sp.colin + (static_cast<uint64_t>(c) * kWave + static_cast<uint32_t>(t)) * kCkRows;
    const uint32_t through_voff = static_cast<uint32_t>((static_cast<uint64_t>(k0) * kWave + static_cast<uint32_t>(t)) * 8u);
    // the left inputs and the row code of a step are loaded one step ahead (a round is otherwise 16 dependent
    // load -> compute steps: what the traceback of the last items of a launch waits for)
    auto inputs_of = [&](int32_t ks, float2& in, uint32_t& code) {
        const int32_t kstep = k0 + ks, r = kstep - t;
        const bool act = valid && ks < static_cast<int32_t>(kCkRows) && r >= 0 && r < la;
        if(!act) {
            in = make_float2(0.0f, 0.0f);
        } else if(sm) {
            const u32x2 v = through ? __builtin_amdgcn_raw_buffer_load_b64(make_rsrc(sp.colin), through_voff, static_cast<uint32_t>(ks) * (kWave * 8u), kAuxAgent)
                                    : __builtin_amdgcn_raw_buffer_load_b64(make_rsrc(sp.colin), through_voff, static_cast<uint32_t>(ks) * (kWave * 8u), kAuxPlain);
            const uint32_t v0 = v[0], v1 = v[1];
            in = make_float2(__builtin_bit_cast(float, v0), __builtin_bit_cast(float, v1));
        } else {
            in = cin[ks];
        }
        code = row_code(r + 1);
    };
    float2 in_next;
    uint32_t code_next;
    inputs_of(0, in_next, code_next);
    const GapVec kv = gap_vec(k);
    for(int32_t ks = 0; ks < static_cast<int32_t>(kCkRows); ++ks) {
        const int32_t kstep = k0 + ks, r = kstep - t;
        const float2 in = in_next;
        const uint32_t arow_next = code_next;
        inputs_of(ks + 1, in_next, code_next);
        if(valid && r >= 0 && r < la) {
            row_l1<W>(kv, st, in.x, in.y, s, lds_tab + arow_next, boff, std::make_integer_sequence<int, W>{});
            bits[(0 * kCkRows + ks) * kWave] = st.acc[ACC_A];
            bits[(1 * kCkRows + ks) * kWave] = st.acc[ACC_B];
            bits[(2 * kCkRows + ks) * kWave] = st.acc[ACC_C];
        }
    }
    return computed;
}

// Wavefronts per workgroup.  Two, not four: a workgroup's slot on its CU is only free for the next launch's
// workgroups when ALL its wavefronts have left, and the SIMDs' issue arbitration makes a workgroup's
// wavefronts finish far apart -- with four per workgroup the ragged end of a launch overlaps the start of
// the next one (pipelined chunks, consecutive batches on two streams) noticeably worse.
constexpr int kCkWaves = 2;
constexpr int kWalkUnknown = 4;  // the cell's tile is not in the round's set


// COATI_HIP_CK_DEBUG bit 1: traceback statistics of a launch (rounds, valid tiles, walker iterations,
// pairs), printed by the launcher after the kernel -- a tuning aid, not part of the product path
__device__ unsigned int g_ck_splice_stats[4];  // strips the true walk took by their record's list, by their bridge; speculative walks, bridges walked
__device__ unsigned long long g_ck_stats[8];  // rounds, valid tiles, walker iterations, pairs, pairs filled twice (left the kept band), hand-overs of row parts, their waits in 10 ns, waits longer than 10 us

// State the walk is in after a move of kind `moved` arrives at body cell (bi, bj): from the
// round's recomputed bits, or kWalkUnknown.
__device__ __forceinline__ int ck_state_after(const PairDesc& pd, const TileSet& ts, const uint32_t* __restrict__ wbits,
                                              uint32_t bi, uint32_t bj, int moved) {
    const uint32_t full = kWave * pd.v_wmain;
    uint32_t strip = bj / full;
    if(strip + 1 >= pd.v_strips) strip = pd.v_strips - 1;
    if(strip != ts.strip) return kWalkUnknown;
    const uint32_t colin = bj - ts.col0, t = colin >> ts.lg, cc = colin & (ts.w - 1u);
    const uint32_t kstep = bi + t;
    const int slot = tileset_slot(ts, t, static_cast<int32_t>(kstep >> kCkRowsLog2));
    if(slot < 0 || ((ts.computed >> slot) & 1ull) == 0ull) return kWalkUnknown;
    const uint32_t ks = kstep & (kCkRows - 1u);
    const uint32_t which = moved == COATI_HIP_OP_INS ? 2u : (moved == COATI_HIP_OP_DEL ? 1u : 0u);
    const uint32_t word = wbits[(which * kCkRows + ks) * kWave + static_cast<uint32_t>(slot)];
    if(which == 2u) return ((word >> (ts.w - 1u - cc)) & 1u) ? COATI_HIP_OP_MATCH : COATI_HIP_OP_INS;
    const uint32_t two = (word >> (2u * (ts.w - 1u - cc))) & 3u;
    if(!(two & 2u)) return COATI_HIP_OP_MATCH;  // the M argument is the maximum (ties: M first)
    return (two & 1u) ? COATI_HIP_OP_INS : COATI_HIP_OP_DEL;
}
// the same for MATRIX cell (i, j), gap_len 1: margins by formula, (0,0) ends the walk
__device__ __forceinline__ int ck_arrival_state(const GapConsts& k, const PairDesc& pd, const TileSet& ts,
                                                const uint32_t* __restrict__ wbits, uint32_t i, uint32_t j, int moved) {
    if(i < 1 && j < 1) return kWalkEnd;  // loop condition of align_pair.cc:268
    if(i >= 1 && j >= 1) return ck_state_after(pd, ts, wbits, i - 1, j - 1, moved);
    float m, d, in;
    margin_mdi(k, 1u, i, j, m, d, in);
    return decide_after(k, moved, m, d, in);
}

struct CkWalkArgs {
    GapConsts k;
    uint32_t lds_tab;
    const char* tab_bytes;
    const uint8_t *a, *b;
    const uint32_t* ck;
    uint32_t* wbits;  // this wavefront's scratch
    bool stats;
    uint32_t band;    // banded checkpoints: what the fill of this pair kept (kCkBandOff: everything)
    unsigned long long kept_all = 0;  // multi-strip pairs: strips (bit s, s < 64) that were filled again with everything kept
    bool through = false;             // a pair cut into row parts: its checkpoints were stored through other wavefronts' L2s
    // what the fill of strip `strip` kept (strips from 64 on of a very long pair are never banded)
    __device__ __forceinline__ uint32_t band_of(uint32_t strip, uint32_t n_strips) const {
        if(n_strips == 1) return band;
        return (strip >= 64u || ((kept_all >> strip) & 1ull)) ? kCkBandOff : band;
    }
};

// traceback<tropical> (align_pair.cc:249-303) of one pair by one WAVEFRONT, in rounds (above).
// Within a round the walk is the wave-cooperative one of common.hpp: lane l looks up the state
// after l+1 further moves of the current kind, a ballot finds where the run ends.  Ops are
// written right-to-left into the pair's slot so they end up in alignment order.
// kThrough: the results are stored write-through at system scope (viterbi_ck_stream: the host's copy engine
// reads them while the kernel is still running, so they may not sit in an XCD's L2).
template <bool kThrough, typename V>
__device__ __forceinline__ void put_result(V* p, V v) {
    if constexpr(kThrough)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else
        *p = v;
}

// THE SPLICED TRACEBACK OF A MULTI-STRIP PAIR (round 6; resident launches).  A pair of S strips used to be walked by the
// wavefront of its last strip alone, ~100 recompute rounds of ~25 us in a row for a 16 kb pair while the wavefronts of the
// other strips had nothing left to do (64 x 16 kb: 2.6 of 10.9 ms).  Now every strip's wavefront, when its fill is done,
//   (1) walks its OWN strip speculatively (mode kCkSpec): from the cell where the pair's straight line (0,0) -> (la,lb) enters
//       the strip on the right, with the same rounds and the same walker, into the strip's record area (common.hpp:
//       kCkRec*) -- the ops, and every run start {i, j, state, ops so far}.  Tracebacks from neighbouring cells merge
//       after a few columns (the decisions are a function of the cell), so from the first run start the true path shares
//       with this walk on, the record IS the true path;
//   (2) walks the BRIDGE (mode kCkBridge): from where the right neighbour's record LEFT that strip (three words the
//       neighbour stores write-through, self-validating) until it meets an entry of its own record;
// and the pair's true walk (the last strip's wavefront, mode kCkTrue) copies: where it enters a strip at the cell a bridge
// started from, bridge + the record's remainder without a round of its own; else it walks with a look at the strip's
// run-start list per iteration and splices at the first hit.  Whatever does not match -- a record cut short by the band, by
// its 4 096 bytes or 64 entries, a path that never merges -- is walked as before: the records are hints, never trusted beyond
// "this (i, j, state) was reached by a walk over the same decisions".
constexpr int kCkPlain = 0, kCkSpec = 1, kCkTrue = 2, kCkBridge = 3;
struct CkSplice {
    int mode = kCkPlain;
    uint32_t strip = 0;       // (kCkSpec, kCkBridge) the strip
    uint8_t* rec = nullptr;   // the pair's record areas (strip s: rec + s * kCkRecBytes)
    bool miss = false;        // COATI_HIP_CK_SPLICE=miss: entries that match nothing (tests)
    bool bridges = false;     // (kCkTrue) look at the bridges
    // (kCkTrue, first in its own strip) what the walk waits for before it reads another strip's checkpoints
    const uint32_t* chain = nullptr;
    uint32_t* splice_stats = nullptr;
};
__device__ __forceinline__ uint32_t ck_rec_word(const uint8_t* area, uint32_t byte) {  // another wavefront's self-validating word
    return __hip_atomic_load(reinterpret_cast<const uint32_t*>(area + byte), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ck_rec_publish(uint8_t* area, uint32_t byte, uint32_t v) {
    __hip_atomic_store(reinterpret_cast<uint32_t*>(area + byte), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (kSplice: the instantiation that knows the modes -- the kernel of launches with multi-strip pairs; everywhere else they are
// compiled out, so that the headline's kernel is the code it was)
// n recorded ops into the pair's slot: four bytes per lane in flight
template <bool kThrough>
__device__ __forceinline__ void ck_copy_ops(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t n, int lane) {
    for(uint32_t q0 = 0; q0 < n; q0 += 4u * kWave) {
        uint8_t v[4];
#pragma unroll
        for(uint32_t u = 0; u < 4u; ++u) {
            const uint32_t q = q0 + u * kWave + static_cast<uint32_t>(lane);
            v[u] = q < n ? src[q] : static_cast<uint8_t>(0);
        }
#pragma unroll
        for(uint32_t u = 0; u < 4u; ++u) {
            const uint32_t q = q0 + u * kWave + static_cast<uint32_t>(lane);
            if(q < n) put_result<kThrough>(&dst[q], v[u]);
        }
    }
}

template <bool kThrough = false, bool kSplice = false>
__device__ __forceinline__ bool ck_walk_pair(int lane, const CkWalkArgs& wa, const PairDesc& pd, uint32_t pair,
                                             uint8_t* __restrict__ ops, uint64_t* __restrict__ ops_start,
                                             uint32_t* __restrict__ ops_len, uint32_t* failed_strip = nullptr,
                                             const CkSplice& sx_in = CkSplice{}) {
    CkSplice sx = sx_in;
    if constexpr(!kSplice) sx = CkSplice{};
    const uint32_t la = pd.la, lb = pd.lb;
    uint32_t i = la, j = lb;  // matrix coordinates of the cell whose decision is pending
    int moved = COATI_HIP_OP_MATCH;  // max_mdi of the terminal-adjusted last cell == its "after match" decision
    uint64_t pos = pd.ops_off + la + lb;
    uint8_t* __restrict__ out = ops;
    TileSet ts{COATI_HIP_OP_MATCH, 0xffffffffu, 16u, 4u, 0u, 0u, 0, 0, 0, 0ull};
    bool ok = true;
    // ---- the speculative walks: one strip, into its record area
    const bool spec = sx.mode == kCkSpec || sx.mode == kCkBridge;  // (wave-uniform)
    const uint32_t full = kWave * pd.v_wmain;
    const uint32_t spec_col0 = sx.strip * full;  // matrix columns <= this are left of the strip
    uint8_t* const area = spec ? sx.rec + static_cast<uint64_t>(sx.strip) * kCkRecBytes : nullptr;
    uint32_t n_log = 0, merged_at = 0xffffffffu;
    int logged = -1;
    bool stop = false;
    if(sx.mode == kCkSpec) {
        // where the straight line enters the strip (a strip that is not its pair's last has the full width)
        // ... a little BELOW it (viterbi_lp.hip, lp_spec_walk: a walk that starts below the true path climbs to it by a deletion run;
        // one from above has to wander left until the paths meet), inside the band of checkpoints the strip kept
        j = spec_col0 + full;
        const uint32_t half = wa.band_of(sx.strip, pd.v_strips);
        const uint64_t below = min(16u + (la >> 9), half / 2u);
        const uint64_t row = static_cast<uint64_t>(j) * la / lb + below;
        i = static_cast<uint32_t>(row < 1u ? 1u : (row > la ? la : row));
        out = area;
        pos = kCkRecOps;
    } else if(sx.mode == kCkBridge) {
        // where the right neighbour's record left that strip: three self-validating words (they start a launch as 0xffffffff)
        const uint8_t* nb = sx.rec + static_cast<uint64_t>(sx.strip + 1u) * kCkRecBytes + kCkRecHead;
        uint32_t xi = 0xffffffffu, xj = 0xffffffffu, xm = 0xffffffffu;
        for(uint32_t spins = 0; spins < (1u << 14); ++spins) {  // (~50 ms at most; a neighbour that never reports: no bridge)
            xi = ck_rec_word(nb, 0), xj = ck_rec_word(nb, 4), xm = ck_rec_word(nb, 8);
            if(xi != 0xffffffffu && xj != 0xffffffffu && xm != 0xffffffffu) break;
            __builtin_amdgcn_s_sleep(32);
        }
        xi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(xi)));
        xj = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(xj)));
        xm = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(xm)));
        // (only an exit INTO this strip, away from the top margin, is worth a bridge)
        if(xi == 0xffffffffu || xj == 0xffffffffu || xm > 2u || xi < 1u || xi > la || xj <= spec_col0 || xj > spec_col0 + full) return false;
        i = xi, j = xj, moved = static_cast<int>(xm);
        out = area + kCkRecBridgeOps;
        pos = kCkRecOpsB;
        if(lane == 0) {
            uint32_t* bh = reinterpret_cast<uint32_t*>(area + kCkRecBridgeHead);
            bh[0] = xi, bh[1] = xj, bh[2] = xm;
        }
    }
    const uint32_t cap_lo = spec ? 2u * kWave : 0u;  // a speculative walk stops while an iteration's 64 ops still fit
    // (speculative: the strip's top and left edges end the walk -- the margins are the true walk's)
    auto arrive = [&](uint32_t ii, uint32_t jj, int mv) {
        if(spec && (ii == 0u || jj <= spec_col0)) return static_cast<int>(kWalkUnknown);
        return ck_arrival_state(wa.k, pd, ts, wa.wbits, ii, jj, mv);
    };
    bool chain_done = sx.mode != kCkTrue || sx.chain == nullptr;
    while(i >= 1 || j >= 1) {
        if(spec && (i == 0u || j <= spec_col0 || stop)) break;
        if(i >= 1 && j >= 1) {
            // ---- a round: the tile set around body cell (i-1, j-1), recomputed one tile per lane
            const CkStrip sp = ck_strip_of(pd, wa.ck, j - 1);
            if(!chain_done && sp.strip + 1u != pd.v_strips) {
                // the true walk leaves its own strip: from here on it reads what the other strips' wavefronts wrote -- every
                // earlier strip has said "complete" (a chain through the progress words) after releasing checkpoints and record
                if(!wait_progress(sx.chain, la) || __hip_atomic_load(sx.chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kHandoffPoison) {
                    ok = false;
                    break;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                chain_done = true;
                // the inner strips' headers and the ends of their ops areas, asked for now, a strip per lane (they lie in memory,
                // written through other XCDs' L2s; the walk below takes the strips one after the other, two dependent reads each)
                {
                    const uint8_t* ar = sx.rec + static_cast<uint64_t>(min(static_cast<uint32_t>(lane), pd.v_strips - 2u)) * kCkRecBytes;
                    uint32_t acc = 0;
#pragma unroll
                    for(uint32_t k2 = 0; k2 < 12u; ++k2) {
                        const uint32_t off = k2 == 0u ? kCkRecHead : k2 == 1u ? kCkRecBridgeHead : k2 < 10u ? kCkRecOps - 128u * (k2 - 1u) : kCkRecBridgeHead - 128u * (k2 - 9u);
                        acc ^= *reinterpret_cast<const volatile uint32_t*>(ar + off);
                    }
                    asm volatile("" ::"v"(acc));
                }
            }
            if(sx.mode == kCkTrue && sx.bridges && sp.strip + 2u < pd.v_strips) {
                // ---- does a bridge start at this very cell?  Then it, and what it met of the strip's record, are the path
                const uint8_t* ar = sx.rec + static_cast<uint64_t>(sp.strip) * kCkRecBytes;
                const uint32_t* bh = reinterpret_cast<const uint32_t*>(ar + kCkRecBridgeHead);
                const uint32_t* rh = reinterpret_cast<const uint32_t*>(ar + kCkRecHead);
                const uint32_t b_i = bh[0], b_j = bh[1], b_m = bh[2], b_n = bh[3], b_rp = bh[4], b_xi = bh[5], b_xj = bh[6], b_xm = bh[7];
                const uint32_t r_xi = rh[0], r_xj = rh[1], r_xm = rh[2], r_xp = rh[3], r_magic = rh[4];
                bool take = b_i == i && b_j == j && b_m == static_cast<uint32_t>(moved) && bh[8] == kCkRecMagic && b_n <= kCkRecOpsB;
                const bool met = b_rp != 0xffffffffu;
                if(met) take = take && r_magic == kCkRecMagic && r_xp <= b_rp && b_rp <= kCkRecOps;
                else take = take && b_n > 0u && b_xm <= 2u && b_xi <= la && b_xj <= lb;
                take = take && (b_n > 0u || !met || b_rp > r_xp);  // (a bridge that moves the walk: never the same cell again)
                take = __builtin_amdgcn_readfirstlane(static_cast<int>(take)) != 0;
                if(take) {
                    const uint32_t nb = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(b_n)));
                    const uint8_t* bsrc = ar + kCkRecBridgeOps + (kCkRecOpsB - nb);  // the bridge's ops in alignment order
                    ck_copy_ops<kThrough>(out + (pos - nb), bsrc, nb, lane);
                    pos -= nb;
                    if(met) {
                        const uint32_t rp = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(b_rp)));
                        const uint32_t xp = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(r_xp)));
                        const uint32_t n = rp - xp;
                        ck_copy_ops<kThrough>(out + (pos - n), ar + xp, n, lane);
                        pos -= n;
                        i = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(r_xi)));
                        j = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(r_xj)));
                        moved = __builtin_amdgcn_readfirstlane(static_cast<int>(r_xm));
                    } else {
                        i = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(b_xi)));
                        j = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(b_xj)));
                        moved = __builtin_amdgcn_readfirstlane(static_cast<int>(b_xm));
                    }
                    if(sx.splice_stats != nullptr && lane == 0) atomicAdd(sx.splice_stats + 1, 1u);
                    continue;
                }
            }
            ts.mode = moved;
            ts.strip = sp.strip, ts.w = sp.w, ts.lg = sp.lg, ts.col0 = sp.col0;
            ts.bi = static_cast<int32_t>(i - 1), ts.bj = static_cast<int32_t>(j - 1);
            ts.t0 = (j - 1 - sp.col0) >> sp.lg;
            ts.c0 = static_cast<int32_t>((i - 1 + ts.t0) >> kCkRowsLog2);
            int32_t t = 0, c = 0;
            const bool valid = tileset_tile(ts, static_cast<uint32_t>(lane), t, c);
            if(wa.stats) {
                const unsigned long long nv = __builtin_popcountll(__builtin_amdgcn_ballot_w64(valid));
                if(lane == 0) {
                    atomicAdd(&g_ck_stats[0], 1ull);
                    atomicAdd(&g_ck_stats[1], nv);
                }
            }
            // the previous round's lookups are done (their results were consumed by ballots);
            // this round's bits are written and then read by the same wavefront through L2
            bool done;
            if(sp.w == 16)
                done = ck_recompute<16>(wa.k, pd, sp, valid, t, c, wa.lds_tab, wa.tab_bytes, wa.a, wa.b, wa.wbits + lane, wa.band_of(sp.strip, pd.v_strips), wa.through);
            else if(sp.w == 8)
                done = ck_recompute<8>(wa.k, pd, sp, valid, t, c, wa.lds_tab, wa.tab_bytes, wa.a, wa.b, wa.wbits + lane, wa.band_of(sp.strip, pd.v_strips), wa.through);
            else
                done = ck_recompute<4>(wa.k, pd, sp, valid, t, c, wa.lds_tab, wa.tab_bytes, wa.a, wa.b, wa.wbits + lane, wa.band_of(sp.strip, pd.v_strips), wa.through);
            ts.computed = __builtin_amdgcn_ballot_w64(done);
            // the wavefront reads back what it stored itself: once the stores are acknowledged its
            // loads see them (same L1, write-through)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        int st = __builtin_amdgcn_readfirstlane(arrive(i, j, moved));
        if(st == kWalkUnknown) {
            // the pending cell's own tile was not recomputed: with banded checkpoints the walk has left the kept band
            // (the caller fills the pair again with everything kept); otherwise it cannot happen.  Never spin.
            ok = false;
            if(failed_strip != nullptr && j >= 1) *failed_strip = ck_strip_of(pd, wa.ck, j - 1).strip;
            break;
        }
        // (the strip whose list the true walk looks at: the round's -- the inner loop never leaves it)
        const bool look = sx.mode == kCkTrue && i >= 1 && j >= 1 && ts.strip + 1u < pd.v_strips;
        const bool look_own = sx.mode == kCkBridge;  // the bridge looks for its strip's own record
        const uint8_t* const look_area = (look || look_own) ? sx.rec + static_cast<uint64_t>(look_own ? sx.strip : ts.strip) * kCkRecBytes : nullptr;
        while(st != kWalkEnd && st != kWalkUnknown) {
            const uint32_t di = st == COATI_HIP_OP_INS ? 0u : 1u;
            const uint32_t dj = st == COATI_HIP_OP_DEL ? 0u : 1u;
            if(di > i || dj > j) {  // cannot happen for decision bits of a finite path; never walk off the matrix
                st = kWalkEnd;
                i = j = 0;
                break;
            }
            if(look_area != nullptr) {
                // ---- is (i, j, state) a run start of the strip's recorded walk?  From there on the record is this path
                const u32x4 e = *reinterpret_cast<const u32x4*>(look_area + kCkRecList + static_cast<uint32_t>(lane) * 16u);
                const uint32_t e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3];
                const unsigned long long hit = __builtin_amdgcn_ballot_w64(e0 == i && e1 == j && e2 == static_cast<uint32_t>(st));
                if(hit != 0ull) {
                    const uint32_t rp = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(e3), static_cast<int>(__builtin_ctzll(hit))));
                    if(look_own) {  // the bridge has met the record: done
                        merged_at = rp;
                        stop = true;
                        break;
                    }
                    const uint32_t* rh = reinterpret_cast<const uint32_t*>(look_area + kCkRecHead);
                    const uint32_t xi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(rh[0])));
                    const uint32_t xj = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(rh[1])));
                    const uint32_t xm = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(rh[2])));
                    const uint32_t xp = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(rh[3])));
                    const uint32_t magic = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(rh[4])));
                    if(magic == kCkRecMagic && xp < rp && rp <= kCkRecOps && xm <= 2u && xi <= i && xj <= j) {
                        const uint32_t n = rp - xp;
                        ck_copy_ops<kThrough>(out + (pos - n), look_area + xp, n, lane);
                        pos -= n;
                        i = xi, j = xj, moved = static_cast<int>(xm);
                        st = kWalkUnknown;  // (a round around the record's exit comes next)
                        if(sx.splice_stats != nullptr && lane == 0) atomicAdd(sx.splice_stats, 1u);
                        break;
                    }
                }
            }
            if(spec) {
                if(pos < cap_lo) {  // the area is full: the record ends here, at a cell whose arriving move is `moved`
                    stop = true;
                    break;
                }
                if(sx.mode == kCkSpec && st != logged && n_log < 64u) {
                    // a run start of this walk: {i, j, state, ops so far}
                    if(lane == 0) {
                        u32x4 e = {sx.miss ? (i | 0x80000000u) : i, j, static_cast<uint32_t>(st), static_cast<uint32_t>(pos)};
                        *reinterpret_cast<u32x4*>(area + kCkRecList + n_log * 16u) = e;
                    }
                    ++n_log;
                    logged = st;
                }
            }
            // lane l: where the walk is after l+1 more moves of kind st, and in which state
            const uint32_t step = static_cast<uint32_t>(lane) + 1u;
            const bool valid = di * step <= i && dj * step <= j;
            int next = kWalkEnd;
            if(valid) next = arrive(i - di * step, j - dj * step, st);
            if(wa.stats && lane == 0) atomicAdd(&g_ck_stats[2], 1ull);
            const unsigned long long cont = __builtin_amdgcn_ballot_w64(valid && next == st);
            const uint32_t run = cont == ~0ull ? kWave : static_cast<uint32_t>(__builtin_ctzll(~cont));  // lanes that continue
            const uint32_t moves = run == kWave ? kWave : run + 1u;
            for(uint32_t q = lane; q < moves; q += kWave) put_result<kThrough>(&out[pos - 1 - q], static_cast<uint8_t>(st));
            pos -= moves;
            i -= di * moves;
            j -= dj * moves;
            moved = st;  // (i, j) was arrived at by a move of kind st: what a next round, or a record's exit, starts from
            if(run < kWave) st = __builtin_amdgcn_readlane(next, static_cast<int>(run));
        }
        if(st == kWalkEnd) break;
    }
    if(sx.mode == kCkSpec) {
        // the record's header: where the walk stopped -- (i, j), arrived at by `moved` --, the ops in all.  The first three also
        // for the LEFT neighbour's bridge, which may be waiting for them: write-through, self-validating
        if(lane == 0) {
            ck_rec_publish(area, kCkRecHead + 0, i);
            ck_rec_publish(area, kCkRecHead + 4, j);
            ck_rec_publish(area, kCkRecHead + 8, static_cast<uint32_t>(moved));
            uint32_t* rh = reinterpret_cast<uint32_t*>(area + kCkRecHead);
            rh[3] = static_cast<uint32_t>(pos), rh[4] = kCkRecMagic;
        }
        return ok;
    }
    if(sx.mode == kCkBridge) {
        if(lane == 0) {
            uint32_t* bh = reinterpret_cast<uint32_t*>(area + kCkRecBridgeHead);
            bh[3] = kCkRecOpsB - static_cast<uint32_t>(pos), bh[4] = merged_at, bh[5] = i, bh[6] = j, bh[7] = static_cast<uint32_t>(moved), bh[8] = kCkRecMagic;
        }
        return ok;
    }
    if(lane == 0) {
        put_result<kThrough>(&ops_start[pair], pos);
        put_result<kThrough>(&ops_len[pair], static_cast<uint32_t>(pd.ops_off + la + lb - pos));
    }
    return ok;
}

// Viterbi fill + traceback for gap_len == 1.  PERSISTENT: the grid is sized to fill every CU with
// the same number of workgroups (host: ck_launch_shape) and each wavefront pulls work items from
// an atomic queue until it is empty; `items` lists the pairs longest first.
// kSharedTab: the model has ONE substitution table -- one copy per workgroup in LDS (12.4 KB), which
// lets four workgroups (16 wavefronts, 4 per SIMD) share a CU.  Otherwise (per-leaf tables of
// `coati msa`) every wavefront keeps the table of its current pair.
template <bool kSharedTab, bool kMulti>
__global__ __launch_bounds__(kCkWaves* kWave, 4) void viterbi_ck(
    const float* __restrict__ table, GapConsts k, const PairDesc* __restrict__ pairs,
    const WorkItem* __restrict__ items, uint32_t n_items, uint32_t* __restrict__ queue,
    uint32_t* __restrict__ progress, const uint8_t* __restrict__ a_cat, const uint8_t* __restrict__ b_cat,
    uint32_t* __restrict__ ck, float* __restrict__ bnd, float* __restrict__ scores, uint8_t* __restrict__ ops,
    uint64_t* __restrict__ ops_start, uint32_t* __restrict__ ops_len, uint32_t* __restrict__ wscratch,
    uint64_t ck_slot_dwords, uint32_t split_items_word, uint32_t dbg, uint32_t band) {
    // (the cut pairs' tracebacks as items of their own: flag in the word's top bit, abi.hip)
    const bool walk_items = (split_items_word & kCkWalkItemsFlag) != 0u;
    // the spliced traceback of multi-strip pairs (launch_viterbi_ck): 0 off, 1 on, 2 "miss", 3 without bridges; bit 6: the launch
    // is one round of wavefronts, so a wavefront that waits for its neighbour's record keeps nothing from running
    // (kMulti: the instantiation for launches with multi-strip pairs -- their fill, ck_fill_strip<W, true>, and the spliced
    // traceback are compiled out of the other, so that the headline's kernel is the single-strip code and nothing else)
    const uint32_t splice_level = kMulti ? (dbg >> 4) & 3u : 0u;
    const bool bridges_ok = kMulti && ((dbg >> 6) & 1u) != 0u;
    const uint32_t split_items = split_items_word & ~kCkWalkItemsFlag;
    __shared__ float tab_all[kSharedTab ? 1 : kCkWaves][kTabRows * kTabStride];
    const int lane_id = threadIdx.x & (kWave - 1);
    float* tab = tab_all[kSharedTab ? 0 : threadIdx.x / kWave];
    uint32_t tab_held = 0xffffffffu;
    if constexpr(kSharedTab) {
        for(int idx = threadIdx.x; idx < kTabFloats; idx += kCkWaves * kWave) {
            const int r = idx / kTabCols, c = idx - r * kTabCols;
            tab[r * kTabStride + c] = table[idx];
        }
        __syncthreads();
        tab_held = 0u;
    }
    const char* tab_bytes = reinterpret_cast<const char*>(tab);
    const uint32_t lds_tab = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(tab));  // LDS byte address
    // (readfirstlane: the compiler must know this is wave-uniform, or every buffer descriptor derived from it
    // lands in VGPRs and each store becomes a readfirstlane loop)
    const uint32_t wave_id = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x * kCkWaves + threadIdx.x / kWave)));
    uint32_t* wbits = wscratch + static_cast<uint64_t>(wave_id) * kCkScratchDwords;
#ifdef COATI_FILL_TRACE
    const uint32_t trace_wave = (blockIdx.x * kCkWaves + threadIdx.x / kWave) & 4095u;
    uint32_t trace_n = 1;
    if(lane_id == 0) {
        g_ck_trace[trace_wave * 16] = __builtin_amdgcn_s_memrealtime();
        uint32_t hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        g_ck_trace[trace_wave * 16 + 15] = (static_cast<unsigned long long>(xcc_id) << 32) | hw_id;
    }
#endif
    uint32_t redo_ticket = 0xffffffffu;
    uint32_t redo_strip = 0;                 // multi-strip pairs: the strip to fill again (everything kept) before the walk is repeated
    unsigned long long redo_kept_all = 0ull;  // ... and the strips of the pair that have been (this one included)
    uint32_t fused_next = 0xffffffffu;       // (kMulti) the second item of a FUSED two-strip pair (common.hpp), this wavefront's next item
    bool fused_ok = true;                    // ... and whether its first strip went well
    for(;;) {
        // `lane` is made opaque in every iteration: LLVM otherwise treats `lane == 0` as a
        // loop-invariant condition and may peel/unswitch this loop per lane, after which the
        // wave-level operations inside (readfirstlane, DPP, ballots) no longer see the whole wave.
        int lane = lane_id;
        asm volatile("" : "+v"(lane));
        const bool redo = redo_ticket != 0xffffffffu;  // (wave-uniform) the previous item again: its walk left the kept checkpoint band
        const bool chained = kMulti && fused_next != 0xffffffffu;  // (wave-uniform) the second strip of a fused pair: no draw
        uint32_t ticket = atomicAdd(queue, (lane == 0 && !redo && !chained) ? 1u : 0u);  // every lane takes part; lane 0 draws
        ticket = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ticket)));
        if(redo) ticket = redo_ticket;
        if(chained) ticket = fused_next;
        redo_ticket = 0xffffffffu;
        fused_next = 0xffffffffu;
        if(ticket >= n_items) break;
        const WorkItem item = items[ticket];
        const uint32_t pair = item.pair, part = item.strip >> 16;
        const PairDesc pd = pairs[pair];
        const bool multi = kMulti && pd.v_strips > 1;
        // fused two-strip pairs (the planner marks them in launches of many items; with the spliced traceback on -- forced: the
        // planner's rule excludes it -- the marks are ignored): the drawer of the first item does the second too, its drawer moves on
        const bool fused = kMulti && multi && splice_level == 0u && (part == kCkFusedFirst || part == kCkFusedSecond);
        if(fused && part == kCkFusedSecond && !chained && !redo) continue;
        // (a multi-strip pair is walked by the wavefront of its LAST strip; when that walk leaves the band of strip s, this
        // wavefront fills strip s again -- its left boundary column is complete in memory -- and walks again)
        const uint32_t strip = (redo && multi) ? redo_strip : (item.strip & 0xffffu);
        const uint32_t fill_ticket = ticket - ((item.strip & 0xffffu) - strip);  // (the strips of a pair are consecutive items)
        if(!redo) redo_kept_all = 0ull;
        // the pair's checkpoint area: its own, or (single-strip pair of a large batch) this wavefront's slot
        uint32_t* __restrict__ ckp = ck + (pd.flags_off == kCkWaveSlot ? static_cast<uint64_t>(wave_id) * ck_slot_dwords : pd.flags_off);
        bool handoff_ok = true;
        if constexpr(!kSharedTab) {
            if(pd.table != tab_held) {  // (wave-uniform)
                const float* __restrict__ src = table + static_cast<size_t>(pd.table) * kTabFloats;
                for(int idx = lane; idx < kTabFloats; idx += kWave) {
                    const int r = idx / kTabCols, c = idx - r * kTabCols;
                    tab[r * kTabStride + c] = src[idx];
                }
                tab_held = pd.table;
            }
        }
        const uint8_t* __restrict__ a = a_cat + pd.a_off;
        const uint8_t* __restrict__ b = b_cat + pd.b_off;
        // a pair cut into row parts (the last pairs of a large batch: abi.hip, "the ragged end"): this item is steps
        // [kbegin, kend) of the pair's one strip; part p > 0 continues where part p - 1 -- split_items tickets earlier,
        // on whatever wavefront took it -- stopped
        uint32_t kbegin = 0, kend = 0xffffffffu;
        const bool cut = pd.v_parts >= 2 && !redo;  // (redo: the wavefront of the last part fills the whole pair again, all rows)
        // the traceback of a cut pair as an item of its own (walk_items): "part" number = the number of parts; nothing but the walk
        const bool walk_item = cut && walk_items && part == ck_parts_count(pd.v_parts);
        if(cut) {
            const uint32_t nlanes = (min(static_cast<uint32_t>(kWave * kW), pd.lb) + kW - 1) / kW;
            if(walk_item)
                kbegin = kend = pd.la + nlanes - 1;  // (waits until the last row part has said "all steps done")
            else
                ck_part_range(pd.la + nlanes - 1, pd.v_parts, part, kbegin, kend);
            if(part > 0) {
                const unsigned long long t_wait = (dbg & 2u) ? __builtin_amdgcn_s_memrealtime() : 0ull;
                // (no acquire: the predecessor stored the lane state and its checkpoints THROUGH its L2 and this wavefront reads the
                // state past its own -- kAuxAgent; round 5: the release / acquire fence pair per hand-over, an L2 write-back and an
                // L2 invalidate on this chip, cost a 10 000-pair launch 2.8 %)
                handoff_ok = wait_progress_relaxed(progress + ticket - split_items, kbegin);
                if(__hip_atomic_load(progress + ticket - split_items, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kHandoffPoison) handoff_ok = false;
                if((dbg & 2u) && lane == 0) {  // (statistics: how long did this part wait for its predecessor? 100 MHz clock)
                    const unsigned long long waited = __builtin_amdgcn_s_memrealtime() - t_wait;
                    atomicAdd(&g_ck_stats[5], 1ull);
                    atomicAdd(&g_ck_stats[6], waited);
                    if(waited > 1000ull) atomicAdd(&g_ck_stats[7], 1ull);
                }
            }
        }
        // banded checkpoints (above): whole single-strip pairs of the full-width shape only; the second time round
        // (redo: the walk left the kept band) everything is kept
        const uint32_t w_item = strip + 1 == pd.v_strips ? pd.v_wlast : pd.v_wmain;
        // what the item keeps: single-strip pairs of the full-width shape a band around their diagonal; the strips of a
        // multi-strip pair (round 4) a wider one, strips from 64 on everything; a redo everything
        const uint32_t band_pair = multi ? ck_band_half_long(band, pd.la, pd.lb) : ck_band_half(band, pd.la, pd.lb);
        const uint32_t band_now = (!redo && pd.la > 0 && pd.lb > 0 && (multi ? strip < 64u : w_item == 16)) ? band_pair : kCkBandOff;
        // the spliced traceback of a multi-strip pair (ck_walk_pair): the strip's wavefront says "complete" itself, after its
        // speculative walks -- and the pair's last strip only where its walk leaves the strip
        const bool spliced = multi && !redo && splice_level != 0u && !(dbg & 1u) && pd.la > 0 && pd.lb > 0;
        // (fused: nobody waits for this strip's "complete" and nothing it wrote is read by another wavefront -- no release, no
        // chain; ck_fill_strip returns before them)
        const bool defer = spliced || (fused && !redo);
        if(pd.la > 0 && pd.lb > 0 && !walk_item) {  // (without body cells only the margins are walked)
            if(cut)  // (every row part of a pair keeps the same band; the redo above refills the WHOLE pair, alone, into the pair's own storage)
                handoff_ok = ck_fill_strip<16, false, true, true>(k, pd, pair, strip, ticket, lane, lds_tab, tab_bytes, a, b, ckp, bnd, scores, progress, kbegin, kend, nullptr, band_now) && handoff_ok;
            else if(multi && w_item == 16)
                handoff_ok = ck_fill_strip<16, true>(k, pd, pair, strip, fill_ticket, lane, lds_tab, tab_bytes, a, b, ckp, bnd, scores, progress, 0, 0xffffffffu, nullptr, band_now, defer);
            else if(multi && w_item == 8)
                handoff_ok = ck_fill_strip<8, true>(k, pd, pair, strip, fill_ticket, lane, lds_tab, tab_bytes, a, b, ckp, bnd, scores, progress, 0, 0xffffffffu, nullptr, band_now, defer);
            else if(multi)
                handoff_ok = ck_fill_strip<4, true>(k, pd, pair, strip, fill_ticket, lane, lds_tab, tab_bytes, a, b, ckp, bnd, scores, progress, 0, 0xffffffffu, nullptr, band_now, defer);
            else if(w_item == 16)  // (from here on: single-strip pairs -- `multi` took the others)
                handoff_ok = ck_fill_strip<16, false, true>(k, pd, pair, strip, ticket, lane, lds_tab, tab_bytes, a, b, ckp, bnd, scores, progress, 0, 0xffffffffu, nullptr, band_now);
            else if(w_item == 8)
                handoff_ok = ck_fill_strip<8, false, true>(k, pd, pair, strip, ticket, lane, lds_tab, tab_bytes, a, b, ckp, bnd, scores, progress);
            else
                handoff_ok = ck_fill_strip<4, false, true>(k, pd, pair, strip, ticket, lane, lds_tab, tab_bytes, a, b, ckp, bnd, scores, progress);
        }
        COATI_CK_STAMP(0);  // fill of this item done
        if(cut && !walk_item && (part + 1 < ck_parts_count(pd.v_parts) || walk_items)) {
            // not the pair's last row part (or any part of a pair whose traceback is an item of its own): release what this wavefront wrote for the pair (checkpoints, lane state),
            // then say how far the pair has got (or that it is lost)
            // (no release fence: everything this part wrote for the pair went through the L2 -- ck_fill_strip's kThrough)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            publish_progress(progress + ticket, handoff_ok ? kend : kHandoffPoison, lane == kWave - 1);
#ifdef COATI_FILL_TRACE
            COATI_CK_STAMP(1);  // (trace build: a row part is an item of the timeline too)
            trace_n += 2;
#endif
            continue;
        }
        const bool inner_strip = !redo && strip + 1 < pd.v_strips;  // not the last strip of its pair: no traceback here ...
        if(fused && inner_strip) {
            // the first strip of a fused pair: its boundary column is complete in memory (stored through the L2; this wavefront
            // waits for the acknowledgements, and says "complete" for the record -- a redo of the pair looks at the word);
            // the second strip is this wavefront's next item
            publish_progress(progress + ticket, handoff_ok ? pd.la : kHandoffPoison, lane == kWave - 1);
            fused_next = ticket + 1u;
            fused_ok = handoff_ok;
        } else if(fused && !redo) {
            handoff_ok = handoff_ok && fused_ok;
        }
        if((inner_strip && !spliced) || (dbg & 1u)) {
#ifdef COATI_FILL_TRACE
            COATI_CK_STAMP(1);
            trace_n += 2;
#endif
            continue;
        }
        // ---- traceback of this pair by the wavefront of its last strip (... but, spliced, the strip's speculative walks).  What the wave wrote
        // itself: wait until the stores are acknowledged.  What other wavefronts wrote (earlier
        // strips): they released before publishing "complete", which this wave polled; acquire.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (a cut pair: its earlier parts' checkpoints were stored through other wavefronts' L2s, and this wavefront's L2 may hold
        // the lines of an earlier launch: its walk reads them past the L2 -- CkWalkArgs::through -- instead of invalidating it)
        // (spliced: the true walk waits and acquires where it leaves its own strip -- CkSplice::chain)
        // (fused, first time round: both strips' checkpoints are this wavefront's own stores)
        if(pd.v_strips > 1 && !spliced && !(fused && !redo)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if(pd.la == 0 || pd.lb == 0) {
            float m, d, in, score;
            margin_mdi(k, 1u, pd.la, pd.lb, m, d, in);
            (void)terminal_state(k, m, d, in, score);
            if(lane == 0) scores[pair] = score;
        }
        // (multi-strip: the band the pair's strips were filled with, minus the strips filled again since)
        const CkWalkArgs wa{k, lds_tab, tab_bytes, a, b, ckp, wbits, (dbg & 2u) != 0u, multi ? band_pair : band_now, redo_kept_all, cut};
        if((dbg & 2u) && lane == 0 && !inner_strip) atomicAdd(&g_ck_stats[redo ? 4 : 3], 1ull);
        uint32_t failed_strip = 0;
        CkSplice sx;
        if(multi && splice_level != 0u && pd.la > 0 && pd.lb > 0) {
            // (the true walk of a redo looks at the records too: they are walks over the same decisions whatever was kept)
            sx.mode = inner_strip ? kCkSpec : kCkTrue;
            sx.strip = strip;
            sx.rec = reinterpret_cast<uint8_t*>(bnd + pd.bnd_off + ck_rec_first_float(pd.la, pd.v_strips));
            sx.miss = splice_level == 2u;
            sx.bridges = splice_level != 3u && bridges_ok;
            sx.chain = spliced ? progress + fill_ticket - 1 : nullptr;  // (a redo: the chain was waited for the first time round)
            sx.splice_stats = (dbg & 2u) ? g_ck_splice_stats : nullptr;
        }
        // (an inner strip: first its record, then its bridge -- from where the right neighbour's record leaves that strip to this
        // record -- where that neighbour has a record, i.e. is not the pair's last strip, and nothing else is waiting for this
        // wavefront.  ONE call site: the walk with its three recompute shapes is the largest piece of code in the kernel)
        bool walk_ok = true;
#pragma nounroll
        for(int pass = 0; pass < 2; ++pass) {
            if(pass == 1) {
                if(!(inner_strip && sx.bridges && strip + 2 < pd.v_strips)) break;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                sx.mode = kCkBridge;
            }
            const bool r = ck_walk_pair<false, kMulti>(lane, wa, pd, pair, ops, ops_start, ops_len, inner_strip ? nullptr : &failed_strip, sx);
            if(pass == 0) walk_ok = r;
            if(sx.splice_stats != nullptr && lane == 0 && inner_strip) atomicAdd(sx.splice_stats + 2 + pass, 1u);
        }
        if(inner_strip) {
            (void)ck_strip_complete(pd, strip, fill_ticket, lane, progress, handoff_ok);
#ifdef COATI_FILL_TRACE
            COATI_CK_STAMP(1);
            trace_n += 2;
#endif
            continue;
        }
        if(!walk_ok && !multi && band_now != kCkBandOff) {
            // the walk asked for a tile outside the kept band: the same item once more, with everything kept (counted:
            // the word behind the ticket counter, zeroed with it; coati_hip_viterbi_band_stats)
            if(lane == 0) atomicAdd(queue + 1, 1u);
            redo_ticket = ticket;
            continue;
        }
        failed_strip = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(failed_strip)));
        if(!walk_ok && multi && band_pair != kCkBandOff && failed_strip < 64u && !((redo_kept_all >> failed_strip) & 1ull)) {
            // a multi-strip pair: fill THAT strip again with everything kept, then walk again (each strip at most once)
            if(lane == 0) atomicAdd(queue + 1, 1u);
            redo_ticket = ticket;
            redo_strip = failed_strip;
            redo_kept_all |= 1ull << failed_strip;
            continue;
        }
        // NaN = "this pair failed": a producer strip never arrived (spin bound), or the walk lost its way
        if((!handoff_ok || !walk_ok) && lane == 0) scores[pair] = __builtin_nanf("");
        COATI_CK_STAMP(1);  // traceback done
#ifdef COATI_FILL_TRACE
        trace_n += 2;
#endif
    }  // next ticket
}

// ---------------------------------------------------------------------------------------------
// viterbi_ck_stream: ONE persistent launch for a whole coati_hip_viterbi_batch call.  The host keeps
// planning and uploading chunks of pairs while the kernel runs and tells it through `published` (the
// number of work items that are ready; written into HBM by the same in-order upload stream that carried
// the chunk, so the data is there when the number is); the kernel tells the host through a flag in
// page-locked host memory when a chunk's last pair is done -- its results are in host memory by then (round 6: the walks
// store them straight into the caller's page-locked arrays or the slot's staging block, CkStreamChunk::ops_direct ...;
// rounds 3-5, and still where the caller passes no ops array: into the workspace, and the host downloads the chunk's
// results while the kernel works on the next).  No ragged end between chunks, no under-filled ramp-up kernels.
// Visibility: a wavefront that takes a ticket invalidates its vector and scalar caches at system scope
// before it reads anything of the chunk (the slot's addresses held another chunk's data before);
// results are stored write-through at system scope and completed (vmcnt) before the pair is counted.
// Every wait is bounded: a host that stops publishing makes the kernel give up, not hang.
struct CkStreamChunk {
    // every array of a chunk is a 256-byte aligned part of ONE arena (abi.hip): base + 32-bit offsets in units
    // of 256 bytes keep the entry small enough to live in SGPRs while a wavefront works on an item
    uint64_t arena;
    uint32_t off_pairs, off_items, off_a, off_b, off_ck, off_bnd, off_scores, off_ops, off_start, off_len, off_progress;
    uint32_t split_items;  // row part p > 0 of a cut pair waits for the item this many (chunk-local) tickets before its own
    uint32_t* host_flag;   // page-locked host memory: set to chunk_no + 1 when the chunk is complete
    unsigned long long* host_bad;  // page-locked host memory: a code out of range (ck_report_bad), 0 = none
    uint32_t n_pairs, n_items, first_ticket, chunk_no;
    uint32_t done, pad0_;  // pairs finished (device atomics)
    // Results stored by the kernel straight into page-locked HOST memory (device-visible addresses; 0 = into the workspace, the host
    // downloads them): the ops -- the caller's array, or the slot's staging block -- at this chunk's base; the scores / ops offsets /
    // ops lengths of the chunk's pairs; and what the host would add to an ops offset (the chunk's base in the caller's array).
    uint64_t ops_direct, scores_direct, start_direct, len_direct, start_add;
    uint32_t pad_[2];
};
static_assert(sizeof(CkStreamChunk) % 16 == 0, "chunk table entries are copied as a block");
struct CkStreamCtl {
    uint32_t queue, published, closed, error;
    CkStreamChunk chunk[kCkStreamSlots];
};

template <typename T>
__device__ __forceinline__ T sys_load(const T* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <typename T>
__device__ __forceinline__ T dev_load(const T* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The page-locked, device-visible block the host and the kernel talk through.  Host -> kernel: the chunk
// table entries, then ONE 64-bit word {chunks announced, items published} (one store: the pilot never sees
// one half without the other), and "no more chunks".  Kernel -> host: a word per slot, chunk_no + 1 when the
// chunk in it is complete.  Nothing of this goes through hipMemcpy: copies this small are done by a copy
// KERNEL, which could not start while viterbi_ck_stream owns every wavefront slot of the chip.
struct CkStreamHost {
    uint64_t announced;  // (chunks << 32) | items
    uint32_t closed;
    uint32_t n_slots;    // chunk number ci lives in table entry ci % n_slots (the host allocates only the slots a call can use)
    uint32_t pad_[12];
    uint32_t done_flag[16];
    uint64_t t_start, t_done[16];  // device clock (100 MHz) when the pilot started / when a slot's chunk was complete
    unsigned long long bad[16];    // per slot: a sequence code out of range in the chunk (ck_report_bad)
    CkStreamChunk chunk[kCkStreamSlots];
};
static_assert(kCkStreamSlots <= 16 && sizeof(CkStreamChunk) <= 4 * kWave, "the pilot mirrors an entry a word per lane");
static_assert(offsetof(CkStreamHost, t_done) == offsetof(CkStreamHost, done_flag) + 16 * sizeof(uint32_t) + sizeof(uint64_t),
              "the worker that completes a chunk finds t_done[slot] from its done_flag[slot]");

template <bool kSharedTab>
__global__ __launch_bounds__(kCkWaves* kWave, 4) void viterbi_ck_stream(const float* __restrict__ table, GapConsts k,
                                                                          CkStreamCtl* ctl, const CkStreamHost* host,
                                                                          uint32_t* __restrict__ wave_ck, uint64_t wave_slot_dwords,
                                                                          uint32_t* __restrict__ wave_scratch, uint32_t band) {
    __shared__ float tab_all[kSharedTab ? 1 : kCkWaves][kTabRows * kTabStride];
    const int lane_id = threadIdx.x & (kWave - 1);

    float* tab = tab_all[kSharedTab ? 0 : threadIdx.x / kWave];
    uint32_t tab_held = 0xffffffffu;
    if constexpr(kSharedTab) {
        for(int idx = threadIdx.x; idx < kTabFloats; idx += kCkWaves * kWave) {
            const int r = idx / kTabCols, c = idx - r * kTabCols;
            tab[r * kTabStride + c] = table[idx];
        }
        __syncthreads();
        tab_held = 0u;
    }
    if(blockIdx.x == 0 && threadIdx.x / kWave == 0) {
        // PILOT wavefront: the only one that reads the host's block (over PCIe); it mirrors new chunk table
        // entries and the item count into HBM, where the 4 095 workers poll.  (Four thousand wavefronts polling
        // host memory would compete with the uploads for the link.)
        uint32_t last = 0, mirrored = 0;
        const uint32_t n_slots = min(static_cast<uint32_t>(kCkStreamSlots),
                                     max(1u, static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(sys_load(&host->n_slots))))));
        if(lane_id == 0) __hip_atomic_store(const_cast<uint64_t*>(&host->t_start), __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for(uint32_t idle = 0;; ++idle) {
            const uint64_t word = sys_load(&host->announced);
            const uint32_t closed = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(sys_load(&host->closed))));
            uint32_t items = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(word))));
            uint32_t chunks = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(word >> 32))));
            if(closed != 0u) {  // (final once closed is set)
                const uint64_t fin = sys_load(&host->announced);
                items = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(fin))));
                chunks = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(fin >> 32))));
            }
            for(; mirrored != chunks; ++mirrored) {  // entries first ...
                // An entry changes under the eyes of workers that are looking for THEIR chunk (their own entry is complete:
                // `published` covers a ticket only after its entry is in memory -- but the lookup reads every entry).  A
                // worker that read the old first_ticket and the new n_items of a slot took it for its own (first lap: old
                // = 0, so every ticket below n_items matched), indexed the slot's work items with ticket - first_ticket =
                // -1 ... and the GPU faulted (seen once the host announced chunks 50 us apart, round 4).  So n_items is 0
                // while the rest of the entry changes, and is written last; the lookup reads n_items first and reads twice.
                const uint32_t q = mirrored % n_slots;
                constexpr int kCountWord = static_cast<int>(offsetof(CkStreamChunk, n_items) / 4);
                uint32_t* const entry = reinterpret_cast<uint32_t*>(&ctl->chunk[q]);
                if(lane_id == kCountWord) __hip_atomic_store(entry + lane_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                uint32_t v = 0;
                if(lane_id < static_cast<int>(sizeof(CkStreamChunk) / 4)) {
                    v = sys_load(reinterpret_cast<const uint32_t*>(&host->chunk[q]) + lane_id);
                    if(lane_id != kCountWord) __hip_atomic_store(entry + lane_id, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if(lane_id == kCountWord) __hip_atomic_store(entry + lane_id, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if(items != last) {  // ... then the count that makes their tickets valid
                if(lane_id == 0) __hip_atomic_store(&ctl->published, items, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = items;
                idle = 0;
            }
            if(closed != 0u) break;
            if(idle > (1u << 23)) {  // ~20 s without a word from the host: give up, never hang
                if(lane_id == 0) __hip_atomic_store(&ctl->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(64);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if(lane_id == 0) __hip_atomic_store(&ctl->closed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;  // (its workgroup partner works on items as usual)
    }
    const char* tab_bytes = reinterpret_cast<const char*>(tab);
    const uint32_t lds_tab = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(tab));
    const uint32_t wave_id = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x * kCkWaves + threadIdx.x / kWave)));
    uint32_t redo_ticket = 0xffffffffu;
    uint32_t fused_next = 0xffffffffu;  // the second item of a fused two-strip pair (common.hpp: kCkFusedFirst), this wavefront's next item
    bool fused_ok = true, fused_redo = false;  // ... whether its first strip went well, and whether this is the pair's second time round
    for(;;) {
        int lane = lane_id;
        asm volatile("" : "+v"(lane));
        const bool chained = fused_next != 0xffffffffu;  // (the next ticket of the same chunk: published with the first)
        // (as in viterbi_ck; the ticket was published long ago: the wait below returns at once.  A fused pair is done again from its
        // FIRST item, and its second item is a redo too)
        const bool redo = redo_ticket != 0xffffffffu || (chained && fused_redo);
        uint32_t ticket = atomicAdd(&ctl->queue, (lane == 0 && !redo && !chained) ? 1u : 0u);
        ticket = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ticket)));
        if(redo_ticket != 0xffffffffu) ticket = redo_ticket;
        if(chained) ticket = fused_next;
        redo_ticket = 0xffffffffu;
        fused_next = 0xffffffffu;
        // ---- wait until the item is published, or the call is closed (bounded)
        bool mine = false;
        const uint64_t t_wait = __builtin_amdgcn_s_memrealtime();  // (100 MHz)
        for(uint32_t spins = 0;; ++spins) {
            const uint32_t pub = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(dev_load(&ctl->published))));
            if(static_cast<int32_t>(pub - ticket) > 0) {
                mine = true;
                break;
            }
            const uint32_t closed = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(dev_load(&ctl->closed))));
            if(closed != 0u) {  // (the pilot stores the final `published` before it sets `closed`)
                const uint32_t last = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(dev_load(&ctl->published))));
                mine = static_cast<int32_t>(last - ticket) > 0;
                break;
            }
            if(__builtin_amdgcn_s_memrealtime() - t_wait > 3000000000ull) {  // 30 s (the pilot gives up first and closes): never hang
                if(lane == 0) __hip_atomic_store(&ctl->error, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            // thousands of idle wavefronts poll ONE word: back off to a poll every ~10 us, or the memory channel that
            // holds it becomes the bottleneck of the wavefronts that do have work
            __builtin_amdgcn_s_sleep(127);
            if(spins > 2) {
                __builtin_amdgcn_s_sleep(127);
                __builtin_amdgcn_s_sleep(127);
            }
        }
        if(!mine) break;
        // the chunk's arrays were written by the copy engine after this wavefront may have cached the slot's
        // previous contents: drop them (vector L1/L2 at system scope, scalar cache)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        __builtin_amdgcn_s_dcache_inv();
        // ---- which chunk: the slots hold disjoint ticket ranges
        int slot = -1;
        uint32_t candidates = 0;
#pragma unroll
        for(int q = 0; q < kCkStreamSlots; ++q) {
            const uint32_t first = dev_load(&ctl->chunk[q].first_ticket), n = dev_load(&ctl->chunk[q].n_items);
            if(ticket - first < n) candidates |= 1u << q;
        }
        // (another slot's entry may have been changing while it was read -- the pilot above: a candidate counts when a
        // second look, its count FIRST, says the same)
        candidates = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(candidates)));
        while(candidates != 0u) {
            const int q = __builtin_ctz(candidates);
            candidates &= candidates - 1u;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t n = dev_load(&ctl->chunk[q].n_items);
            asm volatile("s_waitcnt vmcnt(0)" : : "v"(n) : "memory");
            const uint32_t first = dev_load(&ctl->chunk[q].first_ticket);
            if(ticket - first < n) slot = q;
        }
        slot = __builtin_amdgcn_readfirstlane(slot);
        if(slot < 0) {  // cannot happen: a published ticket lies in a resident chunk
            if(lane == 0) __hip_atomic_store(&ctl->error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        CkStreamChunk* chp = &ctl->chunk[slot];
        // the entry, wave-uniform (SGPRs): base + offsets
        auto u32 = [](uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); };
        auto u64 = [&](uint64_t v) { return static_cast<uint64_t>(u32(static_cast<uint32_t>(v))) | (static_cast<uint64_t>(u32(static_cast<uint32_t>(v >> 32))) << 32); };
        char* const arena = reinterpret_cast<char*>(u64(chp->arena));
        auto arr = [&](uint32_t off256) { return arena + (static_cast<uint64_t>(u32(off256)) << 8); };
        const uint32_t first_ticket = u32(chp->first_ticket), n_pairs_chunk = u32(chp->n_pairs), chunk_no = u32(chp->chunk_no);
        uint32_t* const host_flag = reinterpret_cast<uint32_t*>(u64(reinterpret_cast<uint64_t>(chp->host_flag)));
        unsigned long long* const host_bad = reinterpret_cast<unsigned long long*>(u64(reinterpret_cast<uint64_t>(chp->host_bad)));
        const PairDesc* __restrict__ ch_pairs = reinterpret_cast<const PairDesc*>(arr(chp->off_pairs));
        const WorkItem* __restrict__ ch_items = reinterpret_cast<const WorkItem*>(arr(chp->off_items));
        const uint8_t* __restrict__ ch_a = reinterpret_cast<const uint8_t*>(arr(chp->off_a));
        const uint8_t* __restrict__ ch_b = reinterpret_cast<const uint8_t*>(arr(chp->off_b));
        uint32_t* __restrict__ ch_ck = reinterpret_cast<uint32_t*>(arr(chp->off_ck));
        float* __restrict__ ch_bnd = reinterpret_cast<float*>(arr(chp->off_bnd));
        float* __restrict__ ch_scores = reinterpret_cast<float*>(arr(chp->off_scores));
        const uint64_t ops_direct = u64(chp->ops_direct);
        uint8_t* __restrict__ ch_ops = ops_direct != 0 ? reinterpret_cast<uint8_t*>(ops_direct) : reinterpret_cast<uint8_t*>(arr(chp->off_ops));
        uint64_t* __restrict__ ch_start = reinterpret_cast<uint64_t*>(arr(chp->off_start));
        uint32_t* __restrict__ ch_len = reinterpret_cast<uint32_t*>(arr(chp->off_len));
        uint32_t* __restrict__ ch_progress = reinterpret_cast<uint32_t*>(arr(chp->off_progress));
        const uint32_t local = ticket - first_ticket;
        const WorkItem item = ch_items[local];
        const uint32_t pair = u32(item.pair), strip_word = u32(item.strip);
        const uint32_t strip = strip_word & 0xffffu, part = strip_word >> 16;  // (row part of a cut pair: the call's last chunks)
        // fused two-strip pairs (every chunk of a streamed call has them marked): the drawer of the first item does both strips
        const bool fused = part == kCkFusedFirst || part == kCkFusedSecond;
        if(fused && part == kCkFusedSecond && !chained && !redo) continue;
        const uint32_t split_items = u32(chp->split_items);
        // (loaded through a pointer the compiler cannot trace to a kernel argument, i.e. with vector loads: made
        // wave-uniform word by word, or every address derived from it -- buffer descriptors included -- counts as
        // divergent)
        PairDesc pd;
        {
            const PairDesc raw = ch_pairs[pair];
            static_assert(sizeof(PairDesc) % 4 == 0, "PairDesc is copied word by word");
            uint32_t words[sizeof(PairDesc) / 4];
            __builtin_memcpy(words, &raw, sizeof raw);
#pragma unroll
            for(size_t q = 0; q < sizeof(PairDesc) / 4; ++q) words[q] = u32(words[q]);
            __builtin_memcpy(&pd, words, sizeof pd);
        }
        // single-strip pairs: the wavefront's own checkpoint slot, shared by all chunks (it holds one item at a time);
        // others: the pair's area in its chunk's workspace
        uint32_t* __restrict__ ckp = pd.flags_off == kCkWaveSlot ? wave_ck + static_cast<uint64_t>(wave_id) * wave_slot_dwords : ch_ck + pd.flags_off;
        uint32_t* wbits = wave_scratch + static_cast<uint64_t>(wave_id) * kCkScratchDwords;
        bool handoff_ok = true;
        if constexpr(!kSharedTab) {
            if(pd.table != tab_held) {
                const float* __restrict__ src = table + static_cast<size_t>(pd.table) * kTabFloats;
                for(int idx = lane; idx < kTabFloats; idx += kWave) {
                    const int r = idx / kTabCols, c = idx - r * kTabCols;
                    tab[r * kTabStride + c] = src[idx];
                }
                tab_held = pd.table;
            }
        }
        const uint8_t* __restrict__ a = ch_a + pd.a_off;
        const uint8_t* __restrict__ b = ch_b + pd.b_off;
        uint32_t kbegin = 0, kend = 0xffffffffu;
        const bool cut = pd.v_parts >= 2 && !redo;
        if(cut) {  // (as in viterbi_ck)
            const uint32_t nlanes = (min(static_cast<uint32_t>(kWave * kW), pd.lb) + kW - 1) / kW;
            ck_part_range(pd.la + nlanes - 1, pd.v_parts, part, kbegin, kend);
            if(part > 0) {  // (no acquire: the lane state and the checkpoints go through the L2s, as in viterbi_ck)
                handoff_ok = wait_progress_relaxed(ch_progress + local - split_items, kbegin);
                if(__hip_atomic_load(ch_progress + local - split_items, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kHandoffPoison) handoff_ok = false;
            }
        }
        // (banded checkpoints: as in viterbi_ck)
        const uint32_t w_item = strip + 1 == pd.v_strips ? pd.v_wlast : pd.v_wmain;
        // (a FUSED two-strip pair keeps a band as the resident kernel's multi-strip pairs do -- both strips are this wavefront's, so
        // a walk that leaves the band has the whole pair filled again with everything kept, from its first item; other multi-strip
        // pairs of a streamed call keep everything)
        const uint32_t band_now = (redo || pd.la == 0 || pd.lb == 0) ? kCkBandOff
                                  : fused                            ? ck_band_half_long(band, pd.la, pd.lb)
                                  : (pd.v_strips == 1 && w_item == 16) ? ck_band_half(band, pd.la, pd.lb)
                                                                       : kCkBandOff;
        if(pd.la > 0 && pd.lb > 0) {
            if(cut)
                handoff_ok = ck_fill_strip<16, false, true, true>(k, pd, pair, strip, local, lane, lds_tab, tab_bytes, a, b, ckp, ch_bnd, ch_scores, ch_progress, kbegin, kend, host_bad, band_now) && handoff_ok;
            else if(w_item == 16)  // (fused, the first strip: no "complete" from in there -- nobody waits for it, nothing is released)
                handoff_ok = ck_fill_strip<16>(k, pd, pair, strip, local, lane, lds_tab, tab_bytes, a, b, ckp, ch_bnd, ch_scores, ch_progress, 0, 0xffffffffu, host_bad, band_now, fused && !redo);
            else if(w_item == 8)
                handoff_ok = ck_fill_strip<8>(k, pd, pair, strip, local, lane, lds_tab, tab_bytes, a, b, ckp, ch_bnd, ch_scores, ch_progress, 0, 0xffffffffu, host_bad);
            else  // (fused, the second strip: its left boundary is this wavefront's own, complete)
                handoff_ok = ck_fill_strip<4>(k, pd, pair, strip, local, lane, lds_tab, tab_bytes, a, b, ckp, ch_bnd, ch_scores, ch_progress, 0, 0xffffffffu, host_bad,
                                              fused ? band_now : kCkBandOff, false, fused && chained);
        }
        if(fused && strip + 1 < pd.v_strips) {
            // the first strip of a fused pair is done: its boundary column is in memory once the stores are acknowledged; the word
            // for the record; the second strip is this wavefront's next item
            publish_progress(ch_progress + local, handoff_ok ? pd.la : kHandoffPoison, lane == kWave - 1);
            fused_next = ticket + 1u;
            fused_ok = handoff_ok;
            fused_redo = redo;
            continue;
        }
        if(fused && chained) handoff_ok = handoff_ok && fused_ok;
        if(cut && part + 1 < ck_parts_count(pd.v_parts)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (no release fence: kThrough)
            publish_progress(ch_progress + local, handoff_ok ? kend : kHandoffPoison, lane == kWave - 1);
            continue;
        }
        if(strip + 1 < pd.v_strips) continue;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if(pd.v_strips > 1 && !(fused && chained)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        float score = 0.0f;
        if(pd.la == 0 || pd.lb == 0) {
            float m, d, in;
            margin_mdi(k, 1u, pd.la, pd.lb, m, d, in);
            (void)terminal_state(k, m, d, in, score);
        }
        const CkWalkArgs wa{k, lds_tab, tab_bytes, a, b, ckp, wbits, false, band_now, 0ull, cut};
        // (results that go straight to host memory: the entry is read again here -- it cannot change while one of its pairs is
        // unfinished -- instead of being carried in SGPRs through the fill)
        const uint64_t start_add = u64(chp->start_add), start_direct = u64(chp->start_direct), len_direct = u64(chp->len_direct);
        PairDesc pdw = pd;
        pdw.ops_off += start_add;  // the walk's positions, and the offset it reports, count from the caller's array's start
        const bool walk_ok = ck_walk_pair<true>(lane, wa, pdw, pair, ch_ops - start_add, start_direct != 0 ? reinterpret_cast<uint64_t*>(start_direct) : ch_start,
                                                len_direct != 0 ? reinterpret_cast<uint32_t*>(len_direct) : ch_len);
        if(!walk_ok && band_now != kCkBandOff) {  // (left the kept band: the same item again -- a fused pair: from its first --, everything kept)
            redo_ticket = (fused && chained) ? ticket - 1u : ticket;
            continue;
        }
        if(lane == 0) {
            // the score (stored plainly by the lane that owned the last column; acknowledged above) goes out
            // write-through like the rest of the pair's results
            if(pd.la > 0 && pd.lb > 0) score = ch_scores[pair];
            if(!handoff_ok || !walk_ok) score = __builtin_nanf("");
            // (the workspace's word too when the score goes straight to the host: the fill's plain store left its line dirty in this
            // XCD's L2, and the slot's next chunk is uploaded around it -- written through here, nothing is left to be evicted later)
            const uint64_t scores_direct = u64(chp->scores_direct);
            __hip_atomic_store(&ch_scores[pair], score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if(scores_direct != 0) __hip_atomic_store(reinterpret_cast<float*>(scores_direct) + pair, score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // the pair's results are in memory -> count it; the last pair of a chunk tells the host
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // A pair that kept its checkpoints in the CHUNK's workspace (several strips, or too long for a wavefront slot) and is not cut
        // into row parts wrote them with plain stores: its last strip's lines may still be dirty in this XCD's L2 (the earlier strips'
        // were written back by the release in front of their "complete").  The host may upload the slot's next chunk the moment the
        // completion word is set -- since round 6 nothing is downloaded in between --, and a line evicted after that would land in
        // it: write them back first.  (Rare pairs: one L2 write-back each.)
        if(pd.flags_off != kCkWaveSlot && !cut) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if(lane == 0) {
            const uint32_t done = __hip_atomic_fetch_add(&chp->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            if(done == n_pairs_chunk) {
                // (done_flag[slot] and t_done[slot] are 128 + 8 * slot bytes apart: CkStreamHost)
                uint64_t* t_done = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(host_flag - slot) + sizeof(uint32_t) * 16 + sizeof(uint64_t)) + slot;
                __hip_atomic_store(t_done, __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(host_flag, chunk_no + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// Debug: the decision byte of every body cell of one pair (coati_hip_debug_viterbi_flags), by
// recomputing EVERY tile from the checkpoints with the traceback's own routine.  One wavefront per
// workgroup, 64 tiles per wavefront and pass.
__global__ __launch_bounds__(kWave) void ck_all_flags(const float* __restrict__ table, GapConsts k,
                                                      const PairDesc* __restrict__ pairs, uint32_t pair,
                                                      const uint8_t* __restrict__ a_cat,
                                                      const uint8_t* __restrict__ b_cat,
                                                      const uint32_t* __restrict__ ck, uint32_t* __restrict__ scratch,
                                                      uint8_t* __restrict__ out) {
    __shared__ float tab[kTabRows * kTabStride];
    const PairDesc pd = pairs[pair];
    const int lane = threadIdx.x;
    const float* __restrict__ src = table + static_cast<size_t>(pd.table) * kTabFloats;
    for(int idx = lane; idx < kTabFloats; idx += kWave) {
        const int r = idx / kTabCols, c = idx - r * kTabCols;
        tab[r * kTabStride + c] = src[idx];
    }
    __syncthreads();
    const uint32_t lds_tab = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(tab));
    const uint8_t* a = a_cat + pd.a_off;
    const uint8_t* b = b_cat + pd.b_off;
    uint32_t* wbits = scratch + static_cast<uint64_t>(blockIdx.x) * kCkScratchDwords;
    const uint32_t bands = ck_bands(pd.la);
    for(uint32_t strip = 0; strip < pd.v_strips; ++strip) {
        const uint32_t col0 = strip * kWave * pd.v_wmain;
        const CkStrip sp = ck_strip_of(pd, ck + pd.flags_off, col0);
        const uint32_t n_tiles = bands * kWave;
        for(uint32_t base = blockIdx.x * kWave; base < n_tiles; base += gridDim.x * kWave) {
            const uint32_t tile = base + lane;
            const int32_t c = static_cast<int32_t>(tile / kWave), t = static_cast<int32_t>(tile % kWave);
            const bool valid = tile < n_tiles;
            if(sp.w == 16)
                ck_recompute<16>(k, pd, sp, valid, t, c, lds_tab, reinterpret_cast<const char*>(tab), a, b, wbits + lane);
            else if(sp.w == 8)
                ck_recompute<8>(k, pd, sp, valid, t, c, lds_tab, reinterpret_cast<const char*>(tab), a, b, wbits + lane);
            else
                ck_recompute<4>(k, pd, sp, valid, t, c, lds_tab, reinterpret_cast<const char*>(tab), a, b, wbits + lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if(valid && t < static_cast<int32_t>(sp.nlanes)) {
                for(uint32_t ks = 0; ks < kCkRows; ++ks) {
                    const int32_t r = c * static_cast<int32_t>(kCkRows) + static_cast<int32_t>(ks) - t;
                    if(r < 0 || r >= static_cast<int32_t>(pd.la)) continue;
                    const uint32_t wa = wbits[(0 * kCkRows + ks) * kWave + lane], wb = wbits[(1 * kCkRows + ks) * kWave + lane],
                                   wc = wbits[(2 * kCkRows + ks) * kWave + lane];
                    for(uint32_t cc = 0; cc < sp.w; ++cc) {
                        const uint32_t bj = sp.col0 + static_cast<uint32_t>(t) * sp.w + cc;
                        if(bj >= pd.lb) break;
                        const uint32_t mm = (wa >> (2u * (sp.w - 1u - cc))) & 3u, dd = (wb >> (2u * (sp.w - 1u - cc))) & 3u,
                                       im = (wc >> (sp.w - 1u - cc)) & 1u;
                        const uint32_t fm = !(mm & 2u) ? 0u : ((mm & 1u) ? 2u : 1u), fd = !(dd & 2u) ? 0u : ((dd & 1u) ? 2u : 1u);
                        out[static_cast<uint64_t>(r) * pd.lb + bj] = static_cast<uint8_t>(fm | (fd << 2) | ((im ^ 1u) << 4));
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the next pass overwrites the scratch
        }
    }
}

// Launch shape of the persistent kernel: `blocks_per_cu` workgroups on each of the 256 CUs (one
// wave per SIMD each), enforced by padding the launch with unused dynamic LDS so that exactly that
// many fit.  All waves start together and draw tickets at once, so a grid with more waves than
// items would scatter the items unevenly over the SIMDs: use no more waves than items.
struct CkShape {
    uint32_t grid;
    size_t dynamic_lds;
};
CkShape ck_launch_shape(uint32_t n_items, bool shared_tab) {
    const uint32_t kCUs = device_cu_count(), kSimds = kCUs * 4;
    const int max_blocks = shared_tab ? 4 : 3;  // wavefronts per SIMD: <= 128 VGPRs -> 4; per-wavefront tables (12.4 KB each): 12 per CU
    const int forced = env_options().fill_blocks_per_cu;
    int best = static_cast<int>(std::min<uint64_t>(max_blocks, (static_cast<uint64_t>(n_items) + kSimds - 1) / kSimds));
    best = std::max(best, 1);
    if(forced >= 1 && forced <= max_blocks) best = forced;
    // LDS footprint per block that admits exactly `best` blocks on a CU's 160 KB: more than
    // 160/(best+1) KB, and `best` of them fit with room for the allocation granule
    const size_t stat = (shared_tab ? 1 : kCkWaves) * kTabRows * kTabStride * sizeof(float);
    static_assert(kCkWaves == 2, "LDS table below is for two wavefronts per workgroup");
    // `best` wavefronts per SIMD = 2 * best workgroups per CU: per-workgroup LDS in (160/(2 best + 1), 160/(2 best)] KB
    constexpr size_t kPerBlock[5] = {0, 72 * 1024, 38 * 1024, 25 * 1024, 19 * 1024};
    const size_t stat_r = (stat + 255) / 256 * 256;
    const size_t dyn = kPerBlock[best] > stat_r ? kPerBlock[best] - stat_r : 0;
    return {kCUs * static_cast<uint32_t>(best) * (4 / kCkWaves), dyn};
}

}  // namespace

#ifdef COATI_FILL_TRACE
extern "C" int coati_hip_debug_trace(unsigned long long* out) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ck_trace), sizeof(g_ck_trace));
    if(e != hipSuccess) return static_cast<int>(e);
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_ck_trace));
    if(e != hipSuccess) return static_cast<int>(e);
    return static_cast<int>(hipMemset(p, 0, sizeof(g_ck_trace)));  // next launch starts clean
}
extern "C" int coati_hip_debug_poll_stats(unsigned long long* out) {
    unsigned long long zero[4] = {0, 0, 0, 0};
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ck_poll), sizeof(g_ck_poll));
    if(e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_ck_poll), zero, sizeof zero);
    return static_cast<int>(e);
}
#endif

uint32_t ck_scratch_waves() { return 256u * 4u * 4u; }  // 4 wavefronts on each of the 1 024 SIMDs
uint64_t ck_scratch_dwords_per_wave() { return kCkScratchDwords; }

// (the default half width of the kept checkpoint band -- COATI_HIP_CK_BAND, else 64 steps: a lane keeps 9 of a 1 kb pair's 67
// bands -- is ck_band_setting() in abi.hip, from the process' options.  Round 5, tools/band_bags.py on one box: 64 against 96
// on the bench's four bags 3.95 / 4.25 / 4.74 / 4.08 ms against 4.01 / 4.30 / 4.76 / 4.19; 48 and 56 are faster still on the
// synthetic bag and up to 0.7 ms slower where pairs are filled twice -- a refill that lands at the end of a launch is a whole pair)

hipError_t launch_viterbi_ck(const BatchDeviceView& v, bool shared_tab, hipStream_t stream) {
    hipError_t e = zero_queue_and_progress(v, v.n_items, stream);  // ticket counter + polled words: zero every launch
    if(e != hipSuccess) return e;
    // strip boundaries of multi-strip pairs are self-validating values (ck_chunk<W, true>): every launch starts from the
    // NaN pattern (64 pairs of 8 kb: 60 MB, ~15 us)
    if(v.multi_strip != 0 && v.bnd_bytes != 0) {
        e = hipMemsetAsync(v.bnd, 0xff, v.bnd_bytes, stream);
        if(e != hipSuccess) return e;
    }
    const CkShape shape = ck_launch_shape(v.n_items, shared_tab);
    // timing experiments only (COATI_HIP_CK_DEBUG): bit 0 = fill only, no traceback
    uint32_t dbg = env_options().ck_debug & 15u;
    // the spliced traceback of multi-strip pairs (ck_walk_pair): records while the launch is at most two rounds of wavefronts
    // (a speculative walk is ~4 rounds of recompute that the strip's wavefront spends before it draws its next item), bridges
    // -- whose wavefront WAITS for its neighbour's record -- only when every item has a wavefront of its own.
    // COATI_HIP_CK_SPLICE = 0 / 1 / miss / nobridge forces the level.
    if(v.multi_strip != 0) {
        const uint32_t waves = shape.grid * kCkWaves;
        const int forced = env_options().ck_splice;
        const uint32_t level = forced >= 0 ? static_cast<uint32_t>(forced) : (v.n_items <= 2u * waves ? 1u : 0u);
        dbg |= (level & 3u) << 4;
        if(v.n_items <= waves) dbg |= 1u << 6;
    }
    // (the debug export of every cell's decisions decodes every tile: that batch keeps every checkpoint)
    const uint32_t band = v.ck_band;
    // (the second instantiation where a launch has multi-strip pairs)
    const bool spliced = v.multi_strip != 0;
    auto launch = [&](auto kernel) {
        if(shape.dynamic_lds > 0) {
            const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shape.dynamic_lds));
            if(ea != hipSuccess) return ea;
        }
        hipLaunchKernelGGL(kernel, dim3(shape.grid), dim3(kCkWaves * kWave), shape.dynamic_lds, stream, v.table,
                           v.k, v.pairs, v.items, v.n_items, v.queue, v.progress, v.a_cat, v.b_cat, v.flags, v.bnd, v.scores,
                           v.ops, v.ops_start, v.ops_len, v.wscratch, v.ck_slot_dwords, v.ck_split_items, dbg, band);
        return hipSuccess;
    };
    e = shared_tab ? (spliced ? launch(viterbi_ck<true, true>) : launch(viterbi_ck<true, false>))
                   : (spliced ? launch(viterbi_ck<false, true>) : launch(viterbi_ck<false, false>));
    if(e != hipSuccess) return e;
    if(dbg & 2u) {
        unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        e = hipStreamSynchronize(stream);
        if(e == hipSuccess) e = hipMemcpyFromSymbol(st, HIP_SYMBOL(g_ck_stats), sizeof st);
        if(e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_ck_stats), zero, sizeof zero);
        if(e != hipSuccess) return e;
        std::fprintf(stderr, "viterbi_ck: %llu pairs, %.2f rounds/pair, %.1f valid tiles/round, %.1f walker iterations/pair, %llu pairs filled twice (band %u); "
                     "%llu row-part hand-overs waited %.1f us on average, %llu of them more than 10 us (%.1f wavefront-ms in all)\n", st[3],
                     st[3] ? double(st[0]) / st[3] : 0.0, st[0] ? double(st[1]) / st[0] : 0.0, st[3] ? double(st[2]) / st[3] : 0.0, st[4], band, st[5],
                     st[5] ? double(st[6]) / st[5] / 100.0 : 0.0, st[7], double(st[6]) / 1e5);
        unsigned int sp[4] = {0, 0, 0, 0}, sp0[4] = {0, 0, 0, 0};
        e = hipMemcpyFromSymbol(sp, HIP_SYMBOL(g_ck_splice_stats), sizeof sp);
        if(e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_ck_splice_stats), sp0, sizeof sp0);
        if(e != hipSuccess) return e;
        if(sp[2] != 0) std::fprintf(stderr, "viterbi_ck splice: %u speculative walks, %u bridges walked; the true walks took %u strips by a record's list, %u by a bridge\n", sp[2], sp[3], sp[0], sp[1]);
    }
    return hipGetLastError();
}

uint64_t ck_stream_ctl_bytes() { return sizeof(CkStreamCtl); }
uint64_t ck_stream_error_offset() { return offsetof(CkStreamCtl, error); }
uint64_t ck_stream_host_bytes() { return sizeof(CkStreamHost); }
void* ck_stream_host_entry(void* host, int slot) { return &static_cast<CkStreamHost*>(host)->chunk[slot]; }
volatile uint32_t* ck_stream_host_done_flag(void* host, int slot) { return &static_cast<CkStreamHost*>(host)->done_flag[slot]; }
void ck_stream_host_announce(void* host, uint32_t chunks, uint32_t items) {
    __atomic_store_n(&static_cast<CkStreamHost*>(host)->announced, (static_cast<uint64_t>(chunks) << 32) | items, __ATOMIC_RELEASE);
}
double ck_stream_host_done_ms(void* host, int slot) {
    const CkStreamHost* h = static_cast<const CkStreamHost*>(host);
    return static_cast<double>(static_cast<int64_t>(h->t_done[slot] - h->t_start)) * 1e-5;
}
unsigned long long ck_stream_host_bad(void* host, int slot) { return __atomic_load_n(&static_cast<CkStreamHost*>(host)->bad[slot], __ATOMIC_ACQUIRE); }
void ck_stream_host_set_slots(void* host, uint32_t n_slots) { static_cast<CkStreamHost*>(host)->n_slots = n_slots; }
void ck_stream_host_close(void* host) { __atomic_store_n(&static_cast<CkStreamHost*>(host)->closed, 1u, __ATOMIC_SEQ_CST); }

void ck_stream_fill_chunk(void* host, void* host_dev, int slot, const void* arena, const BatchDeviceView& v, uint32_t n_pairs,
                          uint32_t first_ticket, uint32_t chunk_no, const CkStreamDirect& direct) {
    void* host_entry = ck_stream_host_entry(host, slot);
    uint32_t* host_flag_dev = &static_cast<CkStreamHost*>(host_dev)->done_flag[slot];
    CkStreamChunk c{};
    const char* base = static_cast<const char*>(arena);
    auto off = [&](const void* p) { return static_cast<uint32_t>((static_cast<const char*>(p) - base) >> 8); };
    c.arena = reinterpret_cast<uint64_t>(arena);
    c.off_pairs = off(v.pairs), c.off_items = off(v.items), c.off_a = off(v.a_cat), c.off_b = off(v.b_cat), c.off_ck = off(v.flags);
    c.off_bnd = off(v.bnd), c.off_scores = off(v.scores), c.off_ops = off(v.ops), c.off_start = off(v.ops_start), c.off_len = off(v.ops_len);
    c.ops_direct = reinterpret_cast<uint64_t>(direct.ops), c.scores_direct = reinterpret_cast<uint64_t>(direct.scores);
    c.start_direct = reinterpret_cast<uint64_t>(direct.ops_start), c.len_direct = reinterpret_cast<uint64_t>(direct.ops_len), c.start_add = direct.start_add;
    c.off_progress = off(v.progress);
    c.split_items = v.ck_split_items;
    c.host_flag = host_flag_dev;
    c.host_bad = &static_cast<CkStreamHost*>(host_dev)->bad[slot];
    static_cast<CkStreamHost*>(host)->bad[slot] = 0;
    c.n_pairs = n_pairs, c.n_items = v.n_items, c.first_ticket = first_ticket, c.chunk_no = chunk_no, c.done = 0;
    std::memcpy(host_entry, &c, sizeof c);
}

hipError_t launch_viterbi_ck_stream(const float* table, GapConsts k, bool shared_tab, void* ctl, const void* host_words, uint32_t* wave_ck,
                                    uint64_t wave_slot_dwords, uint32_t* wave_scratch, uint32_t band, hipStream_t stream) {
    const CkShape shape = ck_launch_shape(0xffffffffu, shared_tab);  // the whole chip: the kernel does not know how much is coming
    const void* fn = shared_tab ? reinterpret_cast<const void*>(viterbi_ck_stream<true>) : reinterpret_cast<const void*>(viterbi_ck_stream<false>);
    if(shape.dynamic_lds > 0) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shape.dynamic_lds));
        if(e != hipSuccess) return e;
    }
    if(shared_tab)
        hipLaunchKernelGGL(viterbi_ck_stream<true>, dim3(shape.grid), dim3(kCkWaves * kWave), shape.dynamic_lds, stream, table, k,
                           static_cast<CkStreamCtl*>(ctl), static_cast<const CkStreamHost*>(host_words), wave_ck, wave_slot_dwords, wave_scratch,
                           band);
    else
        hipLaunchKernelGGL(viterbi_ck_stream<false>, dim3(shape.grid), dim3(kCkWaves * kWave), shape.dynamic_lds, stream, table, k,
                           static_cast<CkStreamCtl*>(ctl), static_cast<const CkStreamHost*>(host_words), wave_ck, wave_slot_dwords, wave_scratch,
                           band);
    return hipGetLastError();
}

hipError_t launch_ck_all_flags(const BatchDeviceView& v, uint32_t pair, uint32_t* scratch, uint32_t n_waves, uint8_t* out,
                               hipStream_t stream) {
    hipLaunchKernelGGL(ck_all_flags, dim3(n_waves), dim3(kWave), 0, stream, v.table, v.k, v.pairs, pair, v.a_cat, v.b_cat,
                       v.flags, scratch, out);
    return hipGetLastError();
}

}  // namespace coati_hip_detail
