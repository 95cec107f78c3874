// viterbi_k: Viterbi fill + fused traceback for gap unit lengths L = 2 and 3 (`-k 2`, `-k 3`:
// gaps in whole units, e.g. whole codons), on the LIVE cells only.
//
// What it replaces in the reference:
//   forward_impl<tropical, align_pair_work_mem_t>   src/lib/align_pair.cc:62-139  (look_back = L)
//   traceback<tropical> / max_mdi / max_mi          src/lib/align_pair.cc:210-303
//
// With look_back L a match moves (i,j) -> (i-1,j-1), a deletion (i-L,j), an insertion (i,j-L):
// every move keeps (i - j) mod L.  The path starts at (start,start), and the margins are only
// finite where (i - start) or (j - start) is a multiple of L (align_pair.cc:82-91), so every cell
// with (i - j) mod L != 0 is `lowest` in M, D and I and can never be on a path.  The reference
// fills them anyway; this kernel does not: 1/L of the cells carry the whole computation.
//
// Live body cells are (bi, bj) with bi = p*L + r, bj = q*L + r (block row p, block column q,
// phase r).  Inside a block the diagonal runs through the phases:
//   M_r(p,q) = X_{r-1}(p,q) + s      r > 0          X, YL, ZL as in viterbi_l1.hip, per cell:
//   M_0(p,q) = X_{L-1}(p-1,q-1) + s                 X  = max((M+ng)+ng, D+gs, (I+gs)+ng)
//   D_r(p,q) = YL_r(p-1,q)                          YL = max(((M+ng)+go)+e1, D+eL, ((I+gs)+go)+e1)
//   I_r(p,q) = ZL_r(p,q-1)                          ZL = max((M+go)+e1, I+eL)
// with e1 = ge*float(L-1), eL = ge*float(L) (power(), semiring.hpp:81) -- the reference's
// expressions in the reference's order (align_pair.cc:104-119); the traceback decisions use the
// expressions WITHOUT e1 and with ge (align_pair.cc:275-296), exactly as the reference does.
//
// Same machine shape as viterbi_l1 (persistent wavefronts, atomic ticket queue, a lane owns W
// block columns, one block row of skew per lane, DPP hand-off, strips pipelined through HBM
// boundary columns), written in plain HIP C++.  Decision bits (layout: common.hpp, "compact"):
// per strip and wavefront step S = 2L + (L+1)/2 coalesced 256-byte rows.
#include "common.hpp"

#include <algorithm>
#include <cstdlib>

namespace coati_hip_detail {
namespace {

__device__ __forceinline__ uint32_t push_sign(uint32_t acc, float t) {
    return __builtin_amdgcn_alignbit(acc, __builtin_bit_cast(uint32_t, t), 31);  // acc = acc << 1 | sign(t)
}
__device__ __forceinline__ void store_through(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void publish(uint32_t* word, uint32_t rows, bool leader) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if(leader) __hip_atomic_store(word, rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool wait_rows(const uint32_t* word, uint32_t need) {
    for(uint32_t spins = 0; __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need; ++spins) {
        if(spins > (1u << 26)) return false;
        __builtin_amdgcn_s_sleep(4);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}

template <int L, int W>
struct LaneK {
    float X[W];     // X of phase L-1 of the block row processed last: next block row's phase-0 diagonal
    float Y[L][W];  // YL of every phase of that block row: the D values of the next block row
    float xlast_old;  // X[W-1] of the block row before: the right neighbour's diagonal
    float zlast[L];   // ZL of column W-1, per phase: the right neighbour's I values
};

// One work item: one strip (64*W block columns) of one pair, all its block rows.
template <int L, int W>
__device__ __forceinline__ bool fill_strip_k(const GapConsts& k, const PairDesc& pd, uint32_t pair, uint32_t strip,
                                             uint32_t ticket, int lane, const char* tab_bytes,
                                             const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                             uint32_t* __restrict__ flags, float* __restrict__ bnd,
                                             float* __restrict__ scores, uint32_t* __restrict__ progress) {
    constexpr uint32_t kSlots = compact_slots(L);
    const float e1 = k.ge * static_cast<float>(L - 1), eL = k.ge * static_cast<float>(L);
    const uint32_t rows_b = pd.la / L, cols_b = pd.lb / L;  // block rows / block columns
    const uint32_t q0 = strip * (kWave * pd.v_wmain);
    const uint32_t ncol = min(static_cast<uint32_t>(kWave * W), cols_b - q0);
    const uint32_t nlanes = (ncol + W - 1) / W;
    const uint32_t nsteps = rows_b + nlanes - 1;
    const bool last_strip = strip + 1 == pd.v_strips;
    uint32_t* __restrict__ fout = flags + pd.flags_off + strip * compact_strip_dwords(pd.la, L) + lane;
    // strip boundary, one array per boundary: [0, rows_b] X of the strip's last column (entry p+1 =
    // block row p, entry 0 = margin row), then L arrays [rows_b] of ZL per phase
    const uint64_t bstride = (static_cast<uint64_t>(rows_b) + 1) + static_cast<uint64_t>(L) * rows_b;
    float* __restrict__ bnd_x = bnd + pd.bnd_off + strip * bstride;
    float* __restrict__ bnd_z = bnd_x + (rows_b + 1);
    const float* __restrict__ in_x = bnd + pd.bnd_off + (strip - 1) * bstride;
    const float* __restrict__ in_z = in_x + (rows_b + 1);
    bool ok = true;

    // byte offsets (code * 4) of the lane's W * L descendant columns, L per register
    uint32_t boff[W];
#pragma unroll
    for(int c = 0; c < W; ++c) {
        boff[c] = 0;
#pragma unroll
        for(int r = 0; r < L; ++r) {
            const uint32_t bj = (q0 + lane * W + c) * L + r;
            boff[c] |= (bj < pd.lb ? static_cast<uint32_t>(b[bj]) * 4u : 0u) << (8 * r);
        }
    }
    const int last_lane = static_cast<int>((cols_b - 1 - q0) / W), last_c = static_cast<int>((cols_b - 1 - q0) % W);

    LaneK<L, W> st;
#pragma unroll
    for(int c = 0; c < W; ++c) {
        st.X[c] = 0.0f;
#pragma unroll
        for(int r = 0; r < L; ++r) st.Y[r][c] = 0.0f;
    }
    st.xlast_old = 0.0f;
#pragma unroll
    for(int r = 0; r < L; ++r) st.zlast[r] = 0.0f;
    uint32_t codes = 0;  // the L ancestor codes of the block row this lane processes, 8 bits each

    for(uint32_t kbase = 0; kbase < nsteps; kbase += kWave) {
        // ---- per-64-step chunk: lane l holds what lane 0 needs at step kbase + l
        const uint32_t crow = kbase + lane;  // block row
        uint32_t a_chunk = 0;
        float bx = kLowest, bz[L];
#pragma unroll
        for(int r = 0; r < L; ++r) bz[r] = kLowest;
        if(crow < rows_b) {
#pragma unroll
            for(int r = 0; r < L; ++r) a_chunk |= static_cast<uint32_t>(a[crow * L + r]) << (8 * r);
            if(strip == 0) {
                // diagonal of phase 0: matrix cell (crow*L + L - 1, start), margin column (align_pair.cc:82-86)
                if(crow == 0) {
                    bx = (0.0f + k.ng) + k.ng;
                } else {
                    const float dm = (k.ng + k.go) + k.ge * static_cast<float>(crow * L + L - 2);
                    bx = dm + k.gs;
                }
            }
        }
        if(strip > 0) {
            ok = ok && wait_rows(progress + ticket - 1, min(rows_b, kbase + kWave));
            if(crow < rows_b) {
                bx = in_x[crow];
#pragma unroll
                for(int r = 0; r < L; ++r) bz[r] = in_z[static_cast<uint64_t>(r) * rows_b + crow];
            }
        }
        const uint32_t kend = min(static_cast<uint32_t>(kWave), nsteps - kbase);
        for(uint32_t kk = 0; kk < kend; ++kk) {
            const uint32_t kstep = kbase + kk;
            if(kbase == 0 && kk == static_cast<uint32_t>(lane)) {
                // the lane starts: the block row above is the margin (matrix rows 0..L-1).  Only row
                // `start` = L-1 is finite there: I(start, j) = go + ge*float(j-1) where (j-start) % L == 0
                // (align_pair.cc:88-90), i.e. under phase L-1 of every block column.
#pragma unroll
                for(int c = 0; c < W; ++c) {
                    const uint32_t q = q0 + lane * W + c;
                    const float im = k.go + k.ge * static_cast<float>((q + 2) * L - 2);  // matrix col q*L + 2L - 1
                    const float i1 = im + k.gs;
                    st.X[c] = i1 + k.ng;
#pragma unroll
                    for(int r = 0; r < L - 1; ++r) st.Y[r][c] = kLowest;
                    st.Y[L - 1][c] = (i1 + k.go) + e1;
                }
                if(!last_strip && lane == kWave - 1) store_through(&bnd_x[0], st.X[W - 1]);
            }
            // ---- hand-off from the left neighbour (full exec)
            float dchain = shift_in(st.xlast_old, read_lane(bx, kk));
            float zrow[L];
#pragma unroll
            for(int r = 0; r < L; ++r) zrow[r] = shift_in(st.zlast[r], read_lane(bz[r], kk));
            codes = shift_in(codes, read_lane(a_chunk, kk));
            st.xlast_old = st.X[W - 1];
            uint32_t accA[L], accB[L], accC[(L + 1) / 2];
#pragma unroll
            for(int r = 0; r < L; ++r) accA[r] = accB[r] = 0u;
#pragma unroll
            for(int h = 0; h < (L + 1) / 2; ++h) accC[h] = 0u;
            uint32_t arow[L];
#pragma unroll
            for(int r = 0; r < L; ++r) arow[r] = ((codes >> (8 * r)) & 0xffu) * (kTabStride * 4u);
#pragma unroll
            for(int c = 0; c < W; ++c) {
                float dg = dchain;   // X_{L-1}(p-1, q-1)
                dchain = st.X[c];    // (this column's old value is the next column's diagonal)
#pragma unroll
                for(int r = 0; r < L; ++r) {
                    const float s = *reinterpret_cast<const float*>(tab_bytes + arow[r] + ((boff[c] >> (8 * r)) & 0xffu));
                    const float M = dg + s, D = st.Y[r][c], I = zrow[r];
                    const float m1 = M + k.ng, i1 = I + k.gs;
                    const float x1 = m1 + k.ng, x2 = D + k.gs, x3 = i1 + k.ng;
                    const float X = fmaxf(fmaxf(x1, x2), x3);
                    const float y1 = m1 + k.go, y2 = D + k.ge, y3 = i1 + k.go;
                    const float Yd = fmaxf(fmaxf(y1, y2), y3);
                    const float z1 = M + k.go, z2 = I + k.ge;
                    // decisions (common.hpp): "M / D argument is not the maximum", "z1 > z2"
                    accA[r] = push_sign(push_sign(accA[r], x1 - X), x2 - X);
                    accB[r] = push_sign(push_sign(accB[r], y1 - Yd), y2 - Yd);
                    accC[r / 2] = push_sign(accC[r / 2], z2 - z1);
                    st.Y[r][c] = fmaxf(fmaxf(y1 + e1, D + eL), y3 + e1);
                    zrow[r] = fmaxf(z1 + e1, I + eL);
                    dg = X;  // the next phase's diagonal
                }
                st.X[c] = dg;  // X of phase L-1
            }
#pragma unroll
            for(int r = 0; r < L; ++r) st.zlast[r] = zrow[r];
            // ---- decision bits, left-aligned in their dwords (layout: common.hpp, "compact")
            {
                uint32_t* dst = fout + static_cast<uint64_t>(kstep) * (kSlots * kWave);
#pragma unroll
                for(int r = 0; r < L; ++r) {
                    dst[r * kWave] = accA[r] << (32 - 2 * W);
                    dst[(L + r) * kWave] = accB[r] << (32 - 2 * W);
                }
#pragma unroll
                for(int h = 0; h < (L + 1) / 2; ++h)  // a phase pair interleaves like A/B; a lone last phase is 1 bit per column
                    dst[(2 * L + h) * kWave] = accC[h] << (2 * h + 1 < L ? 32 - 2 * W : 32 - W);
            }
            const int p = static_cast<int>(kstep) - lane;  // block row this lane just did
            if(!last_strip && lane == kWave - 1 && p >= 0 && p < static_cast<int>(rows_b)) {
                store_through(&bnd_x[p + 1], st.X[W - 1]);
#pragma unroll
                for(int r = 0; r < L; ++r) store_through(&bnd_z[static_cast<uint64_t>(r) * rows_b + p], st.zlast[r]);
            }
            if(last_strip && p == static_cast<int>(rows_b) - 1 && lane == last_lane) {
                float sc = st.X[0];
#pragma unroll
                for(int c = 1; c < W; ++c) sc = (c == last_c) ? st.X[c] : sc;
                scores[pair] = sc;  // max(M,D,I) of the terminal-adjusted last cell (align_pair.cc:130-138,265)
            }
        }
        if(!last_strip) {
            const uint32_t done = min(kbase + kWave, nsteps);
            if(done > kWave - 1 && done - (kWave - 1) < rows_b) publish(progress + ticket, done - (kWave - 1), lane == kWave - 1);
        }
    }
    if(!last_strip) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        publish(progress + ticket, rows_b, lane == kWave - 1);
    }
    return ok;
}

// kNarrowOnly: every strip of the batch has the narrow shape (the common case: descendants up to
// 1 152 / 1 024 nt) -- fewer registers, three wavefronts per SIMD.
template <int L, bool kNarrowOnly>
__global__ __launch_bounds__(kFillWaves* kWave, kNarrowOnly ? 3 : 2) void viterbi_k(
    const float* __restrict__ table, GapConsts k, const PairDesc* __restrict__ pairs,
    const WorkItem* __restrict__ items, uint32_t n_items, uint32_t* __restrict__ queue,
    uint32_t* __restrict__ progress, const uint8_t* __restrict__ a_cat, const uint8_t* __restrict__ b_cat,
    uint32_t* __restrict__ flags, float* __restrict__ bnd, float* __restrict__ scores,
    uint8_t* __restrict__ ops, uint64_t* __restrict__ ops_start, uint32_t* __restrict__ ops_len) {
    __shared__ float tab_all[kFillWaves][kTabRows * kTabStride];  // one table per wavefront (see viterbi_l1.hip)
    const int lane_id = threadIdx.x & (kWave - 1);
    float* tab = tab_all[threadIdx.x / kWave];
    uint32_t tab_held = 0xffffffffu;
    const char* tab_bytes = reinterpret_cast<const char*>(tab);
    // strip shapes (block columns per lane): the main one is bounded by registers (L * W values of YL
    // per lane), the narrow one makes a 1 kb pair one strip (334 / 501 block columns)
    constexpr int kWMain = L == 3 ? 12 : 16, kWNarrow = L == 3 ? 6 : 8;
    for(;;) {
        int lane = lane_id;  // opaque per iteration (see viterbi_l1.hip)
        asm volatile("" : "+v"(lane));
        uint32_t ticket = atomicAdd(queue, lane == 0 ? 1u : 0u);
        ticket = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ticket)));
        if(ticket >= n_items) break;
        const WorkItem item = items[ticket];
        const uint32_t pair = item.pair, strip = item.strip;
        const PairDesc pd = pairs[pair];
        bool ok = true;
        if(pd.table != tab_held) {
            const float* __restrict__ src = table + static_cast<size_t>(pd.table) * kTabFloats;
            for(int idx = lane; idx < kTabFloats; idx += kWave) {
                const int r = idx / kTabCols, c = idx - r * kTabCols;
                tab[r * kTabStride + c] = src[idx];
            }
            tab_held = pd.table;
        }
        if(pd.la > 0 && pd.lb > 0) {
            const uint8_t* __restrict__ a = a_cat + pd.a_off;
            const uint8_t* __restrict__ b = b_cat + pd.b_off;
            const uint32_t w = strip + 1 == pd.v_strips ? pd.v_wlast : pd.v_wmain;
            if(!kNarrowOnly && w == kWMain)
                ok = fill_strip_k<L, kWMain>(k, pd, pair, strip, ticket, lane, tab_bytes, a, b, flags, bnd, scores, progress);
            else
                ok = fill_strip_k<L, kWNarrow>(k, pd, pair, strip, ticket, lane, tab_bytes, a, b, flags, bnd, scores, progress);
        }
        if(strip + 1 < pd.v_strips) continue;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if(pd.v_strips > 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        viterbi_finish(lane, k, static_cast<uint32_t>(L), pd, pair, flags, ops, ops_start, ops_len, scores);
        if(!ok && lane == 0) scores[pair] = __builtin_nanf("");
    }
}

}  // namespace

hipError_t launch_viterbi_k(const BatchDeviceView& v, bool narrow_only, hipStream_t stream) {
    if(v.gap_len != 2 && v.gap_len != 3) return hipErrorInvalidValue;
    hipError_t e = zero_queue_and_progress(v, v.n_items, stream);  // ticket counter + polled words: zero every launch
    if(e != hipSuccess) return e;
    // two (three for the narrow-only kernel) workgroups per CU, no more wavefronts than items
    const uint32_t blocks = std::min<uint32_t>(narrow_only ? 768u : 512u, std::max<uint32_t>(1u, (v.n_items + kFillWaves - 1) / kFillWaves));
#define COATI_LAUNCH_K(LL, NN)                                                                                          \
    hipLaunchKernelGGL((viterbi_k<LL, NN>), dim3(blocks), dim3(kFillWaves * kWave), 0, stream, v.table, v.k, v.pairs,   \
                       v.items, v.n_items, v.queue, v.progress, v.a_cat, v.b_cat, v.flags, v.bnd, v.scores, v.ops,      \
                       v.ops_start, v.ops_len)
    if(v.gap_len == 2) {
        if(narrow_only) COATI_LAUNCH_K(2, true); else COATI_LAUNCH_K(2, false);
    } else {
        if(narrow_only) COATI_LAUNCH_K(3, true); else COATI_LAUNCH_K(3, false);
    }
#undef COATI_LAUNCH_K
    return hipGetLastError();
}

}  // namespace coati_hip_detail
