// viterbi_l1: the hand-scheduled Viterbi kernel for gap_len == 1 (the default and
// by far the common case): persistent wavefronts, one sequence pair at a time
// per wavefront, fill + fused traceback.
//
// What it replaces in the reference (all CPU, one pair per process):
//   forward_impl<tropical, align_pair_work_mem_t>   src/lib/align_pair.cc:62-139
//   traceback<tropical> / max_mdi / max_mi          src/lib/align_pair.cc:210-303
//
// Design (DESIGN.md has the derivations and the measurements):
//   * a lane owns 16 consecutive descendant columns and walks down the ancestor
//     rows, skewed by one row per lane (anti-diagonal wavefront).  All M/D/I
//     state lives in registers; no fp32 matrix ever reaches memory.
//   * neighbour hand-off between lanes is a DPP `wave_shr:1` move (no LDS).
//   * the 183x15 substitution table is staged in LDS (row stride 17 floats so
//     that a wave's 64 different rows spread over the 32 banks).
//   * the traceback is NOT an arg-max recorded in the fill.  The reference
//     re-derives each decision from the stored scores of the cell it arrives at
//     (align_pair.cc:275-296), so the kernel evaluates exactly those five
//     comparisons per cell (as the sign of a subtraction, deposited with
//     v_alignbit) and stores 5 bits per cell in coalesced 256-byte rows.
//   * the same wavefront then walks its pair's bits (common.hpp: walk_pair).
//
// fp32 only, adds/max/compares in the reference's evaluation order; built with
// -ffp-contract=off -fno-slp-vectorize.
#include "viterbi_cell.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace coati_hip_detail {
#ifdef COATI_FILL_TRACE
// Debug build only (make trace): per-wave wall-clock stamps (s_memrealtime, 100 MHz) of the
// persistent loop, read back by tools/trace_fill.py through coati_hip_debug_trace.
__device__ unsigned long long g_fill_trace[4096 * 16];
#define COATI_STAMP(slot)                                                                          \
    do {                                                                                           \
        if(lane_id == 0 && trace_n + (slot) < 16)                                                  \
            g_fill_trace[trace_wave * 16 + trace_n + (slot)] = __builtin_amdgcn_s_memrealtime();   \
    } while(0)
#else
#define COATI_STAMP(slot) do { } while(0)
#endif
namespace {

// Read-only per-strip context of one wavefront.
struct StripCtx {
    GapConsts k;
    GapVec kv;  // the same four values in VGPRs, for the cell (viterbi_cell.hpp)
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
template <int W, bool kFirst>
__device__ __forceinline__ void one_step(const StripCtx& cx, LaneState<W>& st, uint32_t& arow, float (&s)[W],
                                         const uint32_t (&boff)[W], uint32_t kbase, uint32_t kk, uint32_t a_chunk,
                                         float bx, float bz) {
    constexpr uint32_t kMA = 16 / W, kMC = 32 / W;  // wavefront steps per A/B dword and per C dword
    const GapConsts& k = cx.k;
    const int lane = cx.lane;
    {
        const uint32_t kstep = kbase + kk;
        if constexpr(kFirst) {
            if(kk == static_cast<uint32_t>(lane)) {
                // This lane starts now: state of the margin row (matrix row 0,
                // align_pair.cc:88-90): M = D = lowest, I = go + ge*float(j-1).
                uint32_t bj0 = cx.col0 + lane * W;
                asm volatile("" : "+v"(bj0));  // compute in place: hoisted, these 32 values get spilled
#pragma unroll
                for(int c = 0; c < W; ++c) {
                    const float im = k.go + k.ge * static_cast<float>(bj0 + c);
                    const float i1 = im + k.gs;
                    st.X[c] = i1 + k.ng;
                    st.Y[c] = i1 + k.go;
                }
                if(!cx.last_strip && lane == kWave - 1) store_through(&cx.bnd_x[0], st.X[W - 1]);
            }
        }
        // ---- hand-off from the left neighbour (full exec)
        const float diag = shift_in(st.xlast_old, read_lane(bx, kk));
        const float zl = shift_in(st.zlast, read_lane(bz, kk));
        const uint32_t arow_next = shift_in(arow, read_lane(a_chunk, kk));
        // ---- the W cells (and the LDS gather for the next step)
        row_l1<W>(cx.kv, st, diag, zl, s, cx.lds_tab + arow_next, boff, std::make_integer_sequence<int, W>{});
        arow = arow_next;
        // ---- decision bits (layout in common.hpp): coalesced 256-byte rows, A and B whenever 32
        // bits are complete (every 16/W steps), C every 32/W steps
        if((kstep & (kMA - 1u)) == kMA - 1u) {
            const uint32_t q = kstep & (kMC - 1u);
            uint32_t* dst = cx.fout + static_cast<uint64_t>(kstep / kMC) * kPairDwords + (q / kMA) * (2 * kWave);
            dst[0] = st.acc[ACC_A];
            dst[kWave] = st.acc[ACC_B];
            if(q == kMC - 1u) cx.fout[static_cast<uint64_t>(kstep / kMC) * kPairDwords + 4 * kWave] = st.acc[ACC_C];
        }
        const int r = static_cast<int>(kstep) - lane;  // body row this lane just did
        if(!cx.last_strip && lane == kWave - 1 && r >= 0 && r < static_cast<int>(cx.la)) {
            store_through(&cx.bnd_x[r + 1], st.X[W - 1]);
            store_through(&cx.bnd_z[r], st.zlast);
        }
    }
}

template <int W, bool kFirst>
__device__ __forceinline__ void run_chunk(const StripCtx& cx, LaneState<W>& st, uint32_t& arow, float (&s)[W],
                                          const uint32_t (&boff)[W], uint32_t kbase, uint32_t a_chunk, float bx,
                                          float bz) {
    const uint32_t kend = min(static_cast<uint32_t>(kWave), cx.nsteps - kbase);
    if constexpr(W == 16 && !kFirst) {
        // The hot loop, two steps per iteration: the new X of a column must not overwrite the old
        // one before the next column has taken it as its diagonal input; with two copies of the
        // body the register allocator ping-pongs X between two register sets instead of copying
        // 16 values per step (20 v_mov per step in the single-step loop).
        uint32_t kk = 0;
        for(; kk + 1 < kend; kk += 2) {
            one_step<W, kFirst>(cx, st, arow, s, boff, kbase, kk, a_chunk, bx, bz);
            one_step<W, kFirst>(cx, st, arow, s, boff, kbase, kk + 1, a_chunk, bx, bz);
        }
        if(kk < kend) one_step<W, kFirst>(cx, st, arow, s, boff, kbase, kk, a_chunk, bx, bz);
    } else {
        for(uint32_t kk = 0; kk < kend; ++kk) one_step<W, kFirst>(cx, st, arow, s, boff, kbase, kk, a_chunk, bx, bz);
    }
}

// One work item: one strip (64*W descendant columns) of one pair, all its rows.  Returns false
// if the left neighbour's boundary column did not arrive within the spin bound.
template <int W>
__device__ __forceinline__ bool fill_strip(const GapConsts& k, const PairDesc& pd, uint32_t pair, uint32_t strip,
                                           uint32_t ticket, int lane, uint32_t lds_tab, const char* tab_bytes,
                                           const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                           uint32_t* __restrict__ flags, float* __restrict__ bnd,
                                           float* __restrict__ scores, uint32_t* __restrict__ progress, bool sentinel) {
    const uint32_t la = pd.la, lb = pd.lb;
    const uint32_t col0 = strip * (kWave * pd.v_wmain);  // every strip before this one has the main width
    const uint32_t ncol = min(static_cast<uint32_t>(kWave * W), lb - col0);
    const uint32_t nlanes = (ncol + W - 1) / W;
    const uint32_t nsteps = la + nlanes - 1;
    const bool last_strip = strip + 1 == pd.v_strips;
    uint32_t* __restrict__ fout = flags + pd.flags_off + strip * strip_dwords(la, pd.v_wmain) + lane;
    // strip-boundary columns, one array per strip boundary: [0, la] = X of the strip's last
    // column (index r = X of body row r-1; index 0 = margin row), [la+1, 2la] = Z of body row r.
    // The strips of a pair run on different wavefronts, pipelined through these arrays.
    const uint64_t bstride = 2 * (static_cast<uint64_t>(la) + 1);
    float* __restrict__ bnd_x = bnd + pd.bnd_off + strip * bstride;  // written by this strip
    float* __restrict__ bnd_z = bnd_x + (la + 1);
    const float* __restrict__ in_x = bnd + pd.bnd_off + (strip - 1) * bstride;  // read by it (strip > 0)
    const float* __restrict__ in_z = in_x + (la + 1);
    bool handoff_ok = true;

    // byte offsets of this lane's W table columns
    uint32_t boff[W];
#pragma unroll
    for(int c = 0; c < W; ++c) {
        const uint32_t bj = col0 + lane * W + c;
        boff[c] = bj < lb ? static_cast<uint32_t>(b[bj]) * 4u : 0u;
    }

    const StripCtx cx{k, gap_vec(k), la, col0, nsteps, pair, lds_tab, lane,
                      static_cast<int>((lb - 1 - col0) / W), static_cast<int>((lb - 1 - col0) % W),
                      last_strip, fout, bnd_x, bnd_z, scores};
    LaneState<W> st;
#pragma unroll
    for(int c = 0; c < W; ++c) st.X[c] = st.Y[c] = 0.0f;
#pragma unroll
    for(int p = 0; p < kAccs; ++p) st.acc[p] = 0u;
    st.xlast_old = 0.0f;
    st.zlast = 0.0f;
    // table-row byte offset of the row this lane processes at the CURRENT step,
    // and the W substitution scores gathered for it one step earlier
    uint32_t arow = lane == 0 ? static_cast<uint32_t>(a[0]) * (kTabStride * 4u) : 0u;
    float s[W];
#pragma unroll
    for(int c = 0; c < W; ++c) s[c] = *reinterpret_cast<const float*>(tab_bytes + arow + boff[c]);

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
            }
        }
        if(strip > 0 && sentinel) {
            // Self-validating boundary: the boundary arrays were filled with the NaN pattern 0xffffffff
            // before the launch, the left neighbour stores its values write-through as it produces them, and this
            // strip simply loads its 64 rows (bypassing the L2) until none of them is the sentinel -- one memory round
            // trip per chunk instead of poll + L2 invalidate + load, and no per-chunk drain on the producer's side.
            uint32_t xb = 0, zb = 0;
            for(uint32_t spins = 0;; ++spins) {
                if(crow < la) {
                    xb = __hip_atomic_load(reinterpret_cast<const uint32_t*>(in_x) + crow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    zb = __hip_atomic_load(reinterpret_cast<const uint32_t*>(in_z) + crow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const bool valid = crow >= la || (xb != 0xffffffffu && zb != 0xffffffffu);
                if(__builtin_amdgcn_ballot_w64(valid) == ~0ull) break;
                if(spins > (1u << 24)) {
                    handoff_ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if(crow < la) {
                bx = __builtin_bit_cast(float, xb);
                bz = __builtin_bit_cast(float, zb);
            }
        } else if(strip > 0) {
            // rows kbase .. kbase+63 of the left neighbour's last column must be published
            handoff_ok = handoff_ok && wait_progress(progress + ticket - 1, min(la, kbase + kWave));
            if(__hip_atomic_load(progress + ticket - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kHandoffPoison) handoff_ok = false;
            if(crow < la) {
                bx = in_x[crow];
                bz = in_z[crow];
            }
        }
        // Consume the chunk loads HERE (one wait per 64 steps), not inside the step loop.
        asm volatile("" : "+v"(a_chunk), "+v"(bx), "+v"(bz));
        if(kbase == 0)
            run_chunk<W, true>(cx, st, arow, s, boff, kbase, a_chunk, bx, bz);
        else
            run_chunk<W, false>(cx, st, arow, s, boff, kbase, a_chunk, bx, bz);
        if(!last_strip && !sentinel) {
            // lane 63 has now finished body rows < kbase + 64 - 63; the final count (la) is
            // published below, after the release of the decision bits
            const uint32_t done = min(kbase + kWave, nsteps);
            if(done > kWave - 1 && done - (kWave - 1) < la) publish_progress(progress + ticket, done - (kWave - 1), lane == kWave - 1);
        }
    }
    // score = max(M,D,I) of the terminal-adjusted last cell (align_pair.cc:130-138,265) = X of the last
    // body cell.  The lane that owns the last column is the last active lane, and its last step is
    // the strip's last step (nsteps = la + nlanes - 1): it holds that X now.  (Looked for inside the
    // step loop, the W-1 selects below were speculated into every step by the compiler.)
    if(last_strip && lane == cx.last_lane) {
        float sc = st.X[0];
#pragma unroll
        for(int c = 1; c < W; ++c) sc = (c == cx.last_c) ? st.X[c] : sc;
        scores[pair] = sc;
    }
    // flush the accumulators of an incomplete last dword, left-aligned (layout in common.hpp)
    {
        constexpr uint32_t kMA = 16 / W, kMC = 32 / W;
        const uint32_t g = nsteps / kMC, q = nsteps & (kMC - 1u);  // q steps of the last group are done
        if constexpr(kMA > 1) {
            const uint32_t ra = nsteps & (kMA - 1u);
            if(ra != 0) {
                uint32_t* dst = fout + static_cast<uint64_t>(g) * kPairDwords + (q / kMA) * (2 * kWave);
                dst[0] = st.acc[ACC_A] << (32u - 2u * W * ra);
                dst[kWave] = st.acc[ACC_B] << (32u - 2u * W * ra);
            }
        }
        if(q != 0) fout[static_cast<uint64_t>(g) * kPairDwords + 4 * kWave] = st.acc[ACC_C] << (32u - W * q);
    }
    if(sentinel && strip > 0) {
        // (the chain "every earlier strip has released its decision bits" still runs through the progress words: this
        // strip says "complete" only after its left neighbour has)
        handoff_ok = handoff_ok && wait_progress(progress + ticket - 1, la);
        if(__hip_atomic_load(progress + ticket - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kHandoffPoison) handoff_ok = false;
    }
    if(!last_strip) {
        // The pair's traceback runs on the wavefront of the LAST strip: release this strip's
        // (plainly stored) decision bits before saying "complete".
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        // (a strip whose own input never arrived publishes the poison value: every later strip of the pair,
        // down to the one that writes the result, then knows)
        publish_progress(progress + ticket, handoff_ok ? la : kHandoffPoison, lane == kWave - 1);
    }
    return handoff_ok;
}

// Viterbi fill for gap_len == 1.  PERSISTENT: the grid is sized to fill every CU
// with the same number of workgroups (host: fill_launch_shape) and each
// wavefront pulls work items from an atomic queue until it is empty.  (With one
// workgroup per 4 pairs the hardware dispatcher packs workgroups unevenly --
// in-kernel clocks showed SIMDs running 2x the waves of others -- and a kernel
// took ~2x the time its work implies.)  `items` lists the pairs longest first.
__global__ __launch_bounds__(kFillWaves* kWave, 3) void viterbi_l1(
    const float* __restrict__ table, GapConsts k, const PairDesc* __restrict__ pairs,
    const WorkItem* __restrict__ items, uint32_t n_items, uint32_t* __restrict__ queue,
    uint32_t* __restrict__ progress, const uint8_t* __restrict__ a_cat, const uint8_t* __restrict__ b_cat,
    uint32_t* __restrict__ flags, float* __restrict__ bnd, float* __restrict__ scores,
    uint8_t* __restrict__ ops, uint64_t* __restrict__ ops_start, uint32_t* __restrict__ ops_len, uint32_t mode) {
    const bool sentinel = (mode & 1u) != 0u;  // strips hand their boundary columns over as self-validating values
    // One substitution table per WAVEFRONT in LDS (row stride 17): the pairs of a batch may use
    // different tables (per-leaf branch lengths); a wavefront reloads its copy when the table of
    // its next item differs from the one it holds (11 KB from L2).
    __shared__ float tab_all[kFillWaves][kTabRows * kTabStride];
    const int lane_id = threadIdx.x & (kWave - 1);
    float* tab = tab_all[threadIdx.x / kWave];
    uint32_t tab_held = 0xffffffffu;
    const char* tab_bytes = reinterpret_cast<const char*>(tab);
    const uint32_t lds_tab = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(tab));  // LDS byte address
#ifdef COATI_FILL_TRACE
    const uint32_t trace_wave = (blockIdx.x * kFillWaves + threadIdx.x / kWave) & 4095u;
    uint32_t trace_n = 1;
    if(lane_id == 0) {
        g_fill_trace[trace_wave * 16] = __builtin_amdgcn_s_memrealtime();
        uint32_t hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        g_fill_trace[trace_wave * 16 + 15] = (static_cast<unsigned long long>(xcc_id) << 32) | hw_id;
    }
#endif
    for(;;) {
        // `lane` is made opaque in every iteration: LLVM otherwise treats `lane == 0` as a
        // loop-invariant condition and may peel/unswitch this loop per lane, after which the
        // wave-level operations inside (readfirstlane, DPP, ballots) no longer see the whole wave.
        int lane = lane_id;
        asm volatile("" : "+v"(lane));
        uint32_t ticket = atomicAdd(queue, lane == 0 ? 1u : 0u);  // every lane takes part; lane 0 draws
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
            tab_held = pd.table;
        }
        if(pd.la > 0 && pd.lb > 0) {  // (without body cells only the margins are walked)
            const uint8_t* __restrict__ a = a_cat + pd.a_off;
            const uint8_t* __restrict__ b = b_cat + pd.b_off;
            const uint32_t w = strip + 1 == pd.v_strips ? pd.v_wlast : pd.v_wmain;
            if(w == 16)
                handoff_ok = fill_strip<16>(k, pd, pair, strip, ticket, lane, lds_tab, tab_bytes, a, b, flags, bnd, scores, progress, sentinel);
            else if(w == 8)
                handoff_ok = fill_strip<8>(k, pd, pair, strip, ticket, lane, lds_tab, tab_bytes, a, b, flags, bnd, scores, progress, sentinel);
            else
                handoff_ok = fill_strip<4>(k, pd, pair, strip, ticket, lane, lds_tab, tab_bytes, a, b, flags, bnd, scores, progress, sentinel);
        }
        COATI_STAMP(0);  // fill of this item done
        if(strip + 1 < pd.v_strips) continue;  // not the last strip of its pair: no traceback here
        // ---- traceback of this pair by the wavefront of its last strip, while the bits are still
        // in L2.  What the wave wrote itself: wait until the stores are acknowledged (nobody read
        // these 128-byte aligned lines before, so L1 is cold).  What other wavefronts wrote (earlier
        // strips): they released before publishing "complete", which this wave polled; acquire.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if(pd.v_strips > 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        viterbi_finish(lane, k, 1u, pd, pair, flags, ops, ops_start, ops_len, scores);
        if(!handoff_ok && lane == 0) scores[pair] = __builtin_nanf("");  // a producer never arrived (spin bound)
        COATI_STAMP(1);  // traceback done
#ifdef COATI_FILL_TRACE
        trace_n += 2;
#endif
    }  // next ticket
}


// Launch shape of the persistent fill kernel: `blocks_per_cu` workgroups on each of the 256
// CUs (one wave per SIMD each), enforced by padding the launch with unused dynamic LDS so that
// exactly that many fit.  Three resident waves per SIMD (the VGPR budget) saturate VALU issue
// (measured cost of a pair relative to a saturated SIMD: 1 wave 1.85, 2 waves 1.10, 3 waves
// 1.03).  All waves start together and draw tickets at once, so a grid with more waves than
// items would scatter the items unevenly over the SIMDs: use no more waves than items.
struct FillShape {
    uint32_t grid;
    size_t dynamic_lds;
};
FillShape fill_launch_shape(uint32_t n_items) {
    const uint32_t kCUs = device_cu_count(), kSimds = kCUs * 4;
    constexpr int kMaxBlocks = 3;  // <= 168 VGPRs -> 3 waves per SIMD
    const int forced = env_options().fill_blocks_per_cu;
    int best = static_cast<int>(std::min<uint64_t>(kMaxBlocks, (static_cast<uint64_t>(n_items) + kSimds - 1) / kSimds));
    best = std::max(best, 1);
    if(forced >= 1 && forced <= kMaxBlocks) best = forced;
    // LDS footprint per block that admits exactly `best` blocks on a CU's 160 KB: more than
    // 160/(best+1) KB, and `best` of them fit with room for the allocation granule.  (A first
    // version used 160 KB/best minus 512 B: the allocator rounds up, only best-1 blocks fitted and
    // the rest of the grid started after the queue was empty -- seen in the trace build.)
    constexpr size_t kStatic = kFillWaves * kTabRows * kTabStride * sizeof(float);  // 49 776 B
    constexpr size_t kPerBlock[4] = {0, 96 * 1024, 72 * 1024, 52 * 1024};
    const size_t dyn = kPerBlock[best] - ((kStatic + 255) / 256) * 256;
    return {kCUs * static_cast<uint32_t>(best), dyn};
}

}  // namespace

#ifdef COATI_FILL_TRACE
extern "C" int coati_hip_debug_trace_l1(unsigned long long* out) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fill_trace), sizeof(g_fill_trace));
    if(e != hipSuccess) return static_cast<int>(e);
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_fill_trace));
    if(e != hipSuccess) return static_cast<int>(e);
    return static_cast<int>(hipMemset(p, 0, sizeof(g_fill_trace)));  // next launch starts clean
}
#endif

hipError_t launch_viterbi_l1(const BatchDeviceView& v, hipStream_t stream) {
    hipError_t e = zero_queue_and_progress(v, v.n_items, stream);  // ticket counter + polled words: zero every launch
    if(e != hipSuccess) return e;
    // strip boundaries: self-validating values (fill_strip) unless COATI_HIP_L1_PROGRESS=1 asks for the progress-word
    // protocol (A/B; 160 kb pair: 87.5 -> 85.2 ms with the 0.35 ms fill of the 800 MB of boundary arrays included)
    const bool sentinel = !env_options().l1_progress;
    if(sentinel && v.bnd_bytes != 0) {
        e = hipMemsetAsync(v.bnd, 0xff, v.bnd_bytes, stream);
        if(e != hipSuccess) return e;
    }
    const FillShape shape = fill_launch_shape(v.n_items);
    if(shape.dynamic_lds > 48 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(viterbi_l1), hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(shape.dynamic_lds));
        if(e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(viterbi_l1, dim3(shape.grid), dim3(kFillWaves * kWave), shape.dynamic_lds, stream, v.table, v.k,
                       v.pairs, v.items, v.n_items, v.queue, v.progress, v.a_cat, v.b_cat, v.flags, v.bnd, v.scores,
                       v.ops, v.ops_start, v.ops_len, sentinel ? 1u : 0u);
    return hipGetLastError();
}

}  // namespace coati_hip_detail
