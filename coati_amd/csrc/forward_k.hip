// forward_k: the Forward fill (log semiring) for gap unit lengths L = 2 and 3 on the LIVE cells
// only -- the log-semiring counterpart of viterbi_k.hip (read its header first: cells with
// (i - j) mod L != 0 are `lowest` in M, D and I in both semirings; the sampler never visits them).
//
// What it replaces in the reference:
//   forward -> forward_impl<log, align_pair_work_t>   src/lib/align_pair.cc:62-139,149  (look_back = L)
//
// Live body cells are (p*L + r, q*L + r).  Per cell the reference's candidates in the reference's
// order (align_pair.cc:97-119), e1 = ge*float(L-1), eL = ge*float(L):
//   diagonal  = phase r-1 of the same block, or phase L-1 of block (p-1, q-1) for r = 0
//   up        = phase r of block (p-1, q);   left = phase r of block (p, q-1)
// `plus` is common.hpp: log_plus_exact (bit-exact libm restatements; COATI_HIP_FORWARD_FAST=1 selects the
// hardware exp2/log2 version).
// Output: fp32 M/D/I of every live cell (12 B per live cell = 12/L B per matrix cell; layout
// common.hpp: mdi_index, "compact") and the terminal-adjusted last cell per pair.
#include "common.hpp"

#include <algorithm>

namespace coati_hip_detail {
namespace {

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

template <int L, bool kFast>
__global__ __launch_bounds__(kFillWaves* kWave, kFast ? 3 : 2) void forward_k(
    const float* __restrict__ table, GapConsts k, const PairDesc* __restrict__ pairs,
    const WorkItem* __restrict__ items, uint32_t n_items, uint32_t* __restrict__ queue,
    uint32_t* __restrict__ progress, const uint8_t* __restrict__ a_cat, const uint8_t* __restrict__ b_cat,
    float* __restrict__ bnd, float* __restrict__ mdi, float* __restrict__ final_mdi) {
    constexpr int W = fwd_compact_w(L);
    __shared__ float tab_all[kFillWaves][kTabRows * kTabStride];  // one table per wavefront (see viterbi_l1.hip)
    __shared__ uint64_t exp_tab[32];
    load_exp_table(exp_tab, threadIdx.x);
    __syncthreads();
    auto plus2 = [&](float x, float y) -> float {
        if constexpr(kFast) return log_plus(x, y);
        return log_plus_exact(x, y, exp_tab);
    };
    const int lane_id = threadIdx.x & (kWave - 1);
    float* tab = tab_all[threadIdx.x / kWave];
    uint32_t tab_held = 0xffffffffu;
    const char* tab_bytes = reinterpret_cast<const char*>(tab);
    const float e1 = k.ge * static_cast<float>(L - 1), eL = k.ge * static_cast<float>(L);
    for(;;) {
        int lane = lane_id;  // opaque per iteration (see viterbi_l1.hip)
        asm volatile("" : "+v"(lane));
        uint32_t ticket = atomicAdd(queue, lane == 0 ? 1u : 0u);
        ticket = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ticket)));
        if(ticket >= n_items) break;
        const WorkItem item = items[ticket];
        const uint32_t pair = item.pair, strip = item.strip;
        const PairDesc pd = pairs[pair];
        if(pd.table != tab_held) {
            const float* __restrict__ src = table + static_cast<size_t>(pd.table) * kTabFloats;
            for(int idx = lane; idx < kTabFloats; idx += kWave) {
                const int r = idx / kTabCols, c = idx - r * kTabCols;
                tab[r * kTabStride + c] = src[idx];
            }
            tab_held = pd.table;
        }
        if(pd.la == 0 || pd.lb == 0) {  // no body cells: the last cell is a margin cell
            float m, d, in;
            margin_mdi(k, static_cast<uint32_t>(L), pd.la + L - 1, pd.lb + L - 1, m, d, in);
            if(lane == 0) {
                final_mdi[3 * static_cast<uint64_t>(pair) + 0] = (m + k.ng) + k.ng;
                final_mdi[3 * static_cast<uint64_t>(pair) + 1] = d + k.gs;
                final_mdi[3 * static_cast<uint64_t>(pair) + 2] = (in + k.gs) + k.ng;
            }
            continue;
        }
        const uint8_t* __restrict__ a = a_cat + pd.a_off;
        const uint8_t* __restrict__ b = b_cat + pd.b_off;
        const uint32_t rows_b = pd.la / L, cols_b = pd.lb / L;
        const uint32_t strips = fwd_compact_strips(pd.lb, L);
        const uint32_t q0 = strip * (kWave * W);
        const uint32_t ncol = min(static_cast<uint32_t>(kWave * W), cols_b - q0);
        const uint32_t nlanes = (ncol + W - 1) / W;
        const uint32_t nsteps = rows_b + nlanes - 1;
        const bool last_strip = strip + 1 == strips;
        float* __restrict__ mout = mdi + pd.mdi_off + strip * fwd_compact_strip_floats(pd.la, L) + 3 * lane;
        // strip boundary: per block row (entry p + 1; entry 0 = the margin row) M, D, I of every phase
        // of the strip's last block column
        const uint64_t bstride = 3ull * L * (static_cast<uint64_t>(rows_b) + 1);
        float* __restrict__ bnd_out = bnd + pd.bnd_off + strip * bstride;
        const float* __restrict__ bnd_in = bnd + pd.bnd_off + (strip - 1) * bstride;
        const int last_lane = static_cast<int>((cols_b - 1 - q0) / W), last_c = static_cast<int>((cols_b - 1 - q0) % W);
        bool ok = true;

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
        float P[L][3][W];      // M, D, I of every phase of the block row processed last
        float o[3];            // phase L-1 of column W-1 of the block row before: the right neighbour's diagonal
#pragma unroll
        for(int r = 0; r < L; ++r)
#pragma unroll
            for(int mat = 0; mat < 3; ++mat)
#pragma unroll
                for(int c = 0; c < W; ++c) P[r][mat][c] = kLowest;
        o[0] = o[1] = o[2] = kLowest;
        uint32_t codes = 0;

        for(uint32_t kbase = 0; kbase < nsteps; kbase += kWave) {
            const uint32_t crow = kbase + lane;  // block row lane 0 processes at step kbase + lane
            uint32_t a_chunk = 0;
            float chD[3] = {kLowest, kLowest, kLowest};  // diagonal of phase 0: phase L-1 of (crow - 1, q0 - 1)
            float chLM[L], chLI[L];                       // left cells: M, I of every phase of (crow, q0 - 1)
#pragma unroll
            for(int r = 0; r < L; ++r) chLM[r] = chLI[r] = kLowest;
            if(crow < rows_b) {
#pragma unroll
                for(int r = 0; r < L; ++r) a_chunk |= static_cast<uint32_t>(a[crow * L + r]) << (8 * r);
            }
            if(strip == 0) {
                // matrix column `start` (align_pair.cc:82-86): the diagonal cell of block row crow, phase 0
                // is matrix cell (crow*L + L - 1, start)
                if(crow == 0) chD[0] = 0.0f;
                else if(crow < rows_b) chD[1] = (k.ng + k.go) + k.ge * static_cast<float>(crow * L + L - 2);
            } else {
                ok = ok && wait_rows(progress + ticket - 1, min(rows_b, kbase + kWave));
                if(crow < rows_b) {
                    const float* dgp = bnd_in + 3ull * L * crow + 3 * (L - 1);  // block row crow - 1, phase L-1
                    chD[0] = dgp[0];
                    chD[1] = dgp[1];
                    chD[2] = dgp[2];
                    const float* lfp = bnd_in + 3ull * L * (crow + 1);          // block row crow
#pragma unroll
                    for(int r = 0; r < L; ++r) {
                        chLM[r] = lfp[3 * r];
                        chLI[r] = lfp[3 * r + 2];
                    }
                }
            }
            const uint32_t kend = min(static_cast<uint32_t>(kWave), nsteps - kbase);
            for(uint32_t kk = 0; kk < kend; ++kk) {
                const uint32_t kstep = kbase + kk;
                if(kbase == 0 && kk == static_cast<uint32_t>(lane)) {
                    // the lane starts: the block row above is the margin; only matrix row `start` = L-1 is
                    // finite: I(start, j) = go + ge*float(j-1) under phase L-1 (align_pair.cc:88-90)
#pragma unroll
                    for(int c = 0; c < W; ++c) {
                        const uint32_t q = q0 + lane * W + c;
#pragma unroll
                        for(int r = 0; r < L; ++r) P[r][0][c] = P[r][1][c] = P[r][2][c] = kLowest;
                        P[L - 1][2][c] = k.go + k.ge * static_cast<float>((q + 2) * L - 2);
                    }
                    if(!last_strip && lane == kWave - 1) {
#pragma unroll
                        for(int r = 0; r < L; ++r)
#pragma unroll
                            for(int mat = 0; mat < 3; ++mat) store_through(&bnd_out[3 * r + mat], P[r][mat][W - 1]);
                    }
                }
                // ---- hand-off from the left neighbour (full exec)
                float dg[3], lfM[L], lfI[L];
#pragma unroll
                for(int mat = 0; mat < 3; ++mat) dg[mat] = shift_in(o[mat], read_lane(chD[mat], kk));
#pragma unroll
                for(int r = 0; r < L; ++r) {
                    lfM[r] = shift_in(P[r][0][W - 1], read_lane(chLM[r], kk));
                    lfI[r] = shift_in(P[r][2][W - 1], read_lane(chLI[r], kk));
                }
                codes = shift_in(codes, read_lane(a_chunk, kk));
#pragma unroll
                for(int mat = 0; mat < 3; ++mat) o[mat] = P[L - 1][mat][W - 1];
                uint32_t arow[L];
#pragma unroll
                for(int r = 0; r < L; ++r) arow[r] = ((codes >> (8 * r)) & 0xffu) * (kTabStride * 4u);
                float* dst = mout + static_cast<uint64_t>(kstep) * (3 * L * W * kWave);
#pragma unroll
                for(int c = 0; c < W; ++c) {
                    float dM = dg[0], dD = dg[1], dI = dg[2];  // phase L-1 of block (p-1, q-1)
                    dg[0] = P[L - 1][0][c];                    // (this column's old values: next column's diagonal)
                    dg[1] = P[L - 1][1][c];
                    dg[2] = P[L - 1][2][c];
#pragma unroll
                    for(int r = 0; r < L; ++r) {
                        const float s = *reinterpret_cast<const float*>(tab_bytes + arow[r] + ((boff[c] >> (8 * r)) & 0xffu));
                        const float upM = P[r][0][c], upD = P[r][1][c], upI = P[r][2][c];
                        // align_pair.cc:97-119
                        const float m2m = ((dM + k.ng) + k.ng) + s;
                        const float d2m = (dD + k.gs) + s;
                        const float i2m = ((dI + k.gs) + k.ng) + s;
                        const float m2d = ((upM + k.ng) + k.go) + e1;
                        const float i2d = ((upI + k.gs) + k.go) + e1;
                        const float d2d = upD + eL;
                        const float m2i = (lfM[r] + k.go) + e1;
                        const float i2i = lfI[r] + eL;
                        const float M = plus2(plus2(m2m, d2m), i2m);
                        const float D = plus2(plus2(m2d, d2d), i2d);
                        const float I = plus2(m2i, i2i);
                        P[r][0][c] = M;
                        P[r][1][c] = D;
                        P[r][2][c] = I;
                        lfM[r] = M;
                        lfI[r] = I;
                        dM = M;  // the next phase's diagonal
                        dD = D;
                        dI = I;
                        *reinterpret_cast<Mdi*>(dst + (r * W + c) * (3 * kWave)) = Mdi{M, D, I};
                    }
                }
                const int p = static_cast<int>(kstep) - lane;  // block row this lane just did
                if(!last_strip && lane == kWave - 1 && p >= 0 && p < static_cast<int>(rows_b)) {
                    float* bo = bnd_out + 3ull * L * (p + 1);
#pragma unroll
                    for(int r = 0; r < L; ++r)
#pragma unroll
                        for(int mat = 0; mat < 3; ++mat) store_through(&bo[3 * r + mat], P[r][mat][W - 1]);
                }
                if(last_strip && p == static_cast<int>(rows_b) - 1 && lane == last_lane) {
                    float m = P[L - 1][0][0], d = P[L - 1][1][0], in = P[L - 1][2][0];
#pragma unroll
                    for(int c = 1; c < W; ++c) {
                        m = (c == last_c) ? P[L - 1][0][c] : m;
                        d = (c == last_c) ? P[L - 1][1][c] : d;
                        in = (c == last_c) ? P[L - 1][2][c] : in;
                    }
                    float* f = final_mdi + 3 * static_cast<uint64_t>(pair);  // terminal adjustment (align_pair.cc:130-138)
                    f[0] = (m + k.ng) + k.ng;
                    f[1] = d + k.gs;
                    f[2] = (in + k.gs) + k.ng;
                }
            }
            if(!last_strip) {
                const uint32_t done = min(kbase + kWave, nsteps);
                if(done > kWave - 1) publish(progress + ticket, min(rows_b, done - (kWave - 1)), lane == kWave - 1);
            }
        }
        if(!last_strip) publish(progress + ticket, rows_b, lane == kWave - 1);
        if(!ok && last_strip && lane == 0) final_mdi[3 * static_cast<uint64_t>(pair)] = __builtin_nanf("");
    }
}

}  // namespace

hipError_t launch_forward_k(const BatchDeviceView& v, hipStream_t stream) {
    if(v.gap_len != 2 && v.gap_len != 3) return hipErrorInvalidValue;
    hipError_t e = zero_queue_and_progress(v, v.n_fwd_items, stream);  // ticket counter + polled words: zero every launch
    if(e != hipSuccess) return e;
    // up to three workgroups (12 wavefronts) per CU, no more wavefronts than items
    const uint32_t blocks = std::min<uint32_t>(768u, std::max<uint32_t>(1u, (v.n_fwd_items + kFillWaves - 1) / kFillWaves));
#define COATI_LAUNCH_FK(LL, FF)                                                                                        \
    hipLaunchKernelGGL((forward_k<LL, FF>), dim3(blocks), dim3(kFillWaves * kWave), 0, stream, v.table, v.k, v.pairs,   \
                       v.fwd_items, v.n_fwd_items, v.queue, v.progress, v.a_cat, v.b_cat, v.bnd, v.mdi, v.final_mdi)
    const bool fast = v.fwd_fast != 0;
    if(v.gap_len == 2) {
        if(fast) COATI_LAUNCH_FK(2, true); else COATI_LAUNCH_FK(2, false);
    } else {
        if(fast) COATI_LAUNCH_FK(3, true); else COATI_LAUNCH_FK(3, false);
    }
#undef COATI_LAUNCH_FK
    return hipGetLastError();
}

}  // namespace coati_hip_detail
