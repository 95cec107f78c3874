// dp_generic: the general DP fill -- any gap unit length 1..8 and both semirings.
//
//   viterbi (tropical): forward_impl<tropical, mem>  src/lib/align_pair.cc:62-139,195
//                       + the decision bits and the fused traceback (common.hpp)
//   forward (log):      forward_impl<log, align_pair_work_t>  align_pair.cc:62-139,149
//                       M/D/I of every body cell to HBM (12 B/cell) for sampleback; the
//                       eight edge matrices of align_pair_work_t (align_pair.hpp:94-103)
//                       are NOT stored: they are recomputed on demand from the
//                       neighbours' M/D/I with the fill's own expressions.
//
// Same wavefront shape as viterbi_l1 (one pair per wavefront, a lane owns 16
// columns, one row of skew per lane, DPP hand-off of the diagonal), but written
// in plain C++ in the reference's evaluation order, with the neighbours that are
// gap_len rows up / gap_len columns left kept in LDS:
//   ring[slot][mat][col]   M/D/I of the last gap_len rows of the lane's own columns
//   rowbuf[mat][col + L]   M and I of the newest row of every column; the L entries
//                          in front are the previous strip's last columns
// It is the slower, general path; gap_len 1 Viterbi takes viterbi_l1 instead.
#include "common.hpp"

namespace coati_hip_detail {
namespace {

constexpr int kMaxGapLen = 8;
constexpr int kRowBufCols = kStrip + 16;

// utils.hpp:134-156 (y = -|a-b| <= 0, so only the first two branches are reachable), bit-exact
// (common.hpp: log_plus_exact)
template <bool kLog>
__device__ __forceinline__ float plus(float a, float b, const uint64_t* exp_tab) {
    if constexpr(kLog) return log_plus_exact(a, b, exp_tab);
    return fmaxf(a, b);
}

struct Boundary {  // what lane 0 needs at one wavefront step: its diagonal cell and the L cells to its left
    float dg[3];
};

template <bool kLog>
__global__ __launch_bounds__(kWave) void dp_generic(const float* __restrict__ table, GapConsts k, uint32_t L,
                                                    const PairDesc* __restrict__ pairs,
                                                    const uint32_t* __restrict__ order, uint32_t n_pairs,
                                                    uint32_t* __restrict__ queue, const uint8_t* __restrict__ a_cat,
                                                    const uint8_t* __restrict__ b_cat, uint32_t* __restrict__ flags,
                                                    float* __restrict__ bnd, float* __restrict__ scores,
                                                    uint8_t* __restrict__ ops, uint64_t* __restrict__ ops_start,
                                                    uint32_t* __restrict__ ops_len, float* __restrict__ mdi,
                                                    float* __restrict__ final_mdi) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* tab = lds;                                  // [183][17]
    float* ring = tab + kTabRows * kTabStride;         // [L][3][kStrip]
    float* rowbuf = ring + L * 3 * kStrip;             // [2][kRowBufCols]
    float* chunk = rowbuf + 2 * kRowBufCols;           // [64][3 + 2 * kMaxGapLen]
    constexpr int kChunkStride = 3 + 2 * kMaxGapLen;
    uint32_t tab_held = 0xffffffffu;  // which of the model's tables the LDS copy holds
    __shared__ uint64_t exp_tab[32];
    load_exp_table(exp_tab, threadIdx.x);
    __syncthreads();
    const int lane_id = threadIdx.x;
    const float ext_lm1 = k.ge * static_cast<float>(L - 1), ext_l = k.ge * static_cast<float>(L);  // power(), semiring.hpp:81

    for(;;) {
        // `lane` is made opaque in every iteration: LLVM otherwise treats `lane == 0` as a
        // loop-invariant condition and peels/unswitches this loop per lane, after which the
        // wave-level operations inside (readfirstlane, ballots) no longer see the whole wave
        // (observed: lanes != 0 spinning forever on ticket 0).
        int lane = lane_id;
        asm volatile("" : "+v"(lane));
        uint32_t ticket = atomicAdd(queue, lane == 0 ? 1u : 0u);  // every lane takes part; lane 0 draws
        ticket = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ticket)));
        if(ticket >= n_pairs) break;
        const uint32_t pair = order[ticket];
        const PairDesc pd = pairs[pair];
        const uint32_t la = pd.la, lb = pd.lb;
        if(pd.table != tab_held) {
            __builtin_amdgcn_s_barrier();  // (single wave: the previous pair's reads are done)
            const float* __restrict__ src = table + static_cast<size_t>(pd.table) * kTabFloats;
            for(int idx = lane; idx < kTabFloats; idx += kWave) {
                const int r = idx / kTabCols, c = idx - r * kTabCols;
                tab[r * kTabStride + c] = src[idx];
            }
            tab_held = pd.table;
            __builtin_amdgcn_s_barrier();
        }
        const uint8_t* __restrict__ a = a_cat + pd.a_off;
        const uint8_t* __restrict__ b = b_cat + pd.b_off;
        float last_m = kLowest, last_d = kLowest, last_i = kLowest;  // unadjusted last cell (in its owner lane)

        if(la > 0 && lb > 0) {
            const uint32_t strips = n_strips(lb);
            // strip-boundary buffer: per body row r (index r+1; index 0 = the margin row above the
            // body): the M/D/I of the strip's last column [3] and M, I of its last L columns [2L]
            const uint32_t bstride = 3 + 2 * L;
            float* __restrict__ bbuf = bnd + pd.bnd_off;
            for(uint32_t strip = 0; strip < strips; ++strip) {
                const uint32_t col0 = strip * kStrip;
                const uint32_t ncol = min(static_cast<uint32_t>(kStrip), lb - col0);
                const uint32_t nlanes = (ncol + kW - 1) / kW;
                const uint32_t nsteps = la + nlanes - 1;
                const bool last_strip = strip + 1 == strips;
                const int last_lane = static_cast<int>(((lb - 1) & (kStrip - 1)) / kW), last_c = (lb - 1) & (kW - 1);
                uint32_t* __restrict__ fout = flags + pd.flags_off + strip * strip_dwords(la) + lane;
                float* __restrict__ mout = mdi + pd.mdi_off + strip * strip_mdi_floats(la) + 3 * lane;

                uint32_t bcode[kW];
#pragma unroll
                for(int c = 0; c < kW; ++c) {
                    const uint32_t bj = col0 + lane * kW + c;
                    bcode[c] = bj < lb ? static_cast<uint32_t>(b[bj]) : 0u;
                }
                float Mp[kW], Dp[kW], Ip[kW];  // the row above (diagonal inputs)
#pragma unroll
                for(int c = 0; c < kW; ++c) Mp[c] = Dp[c] = Ip[c] = kLowest;
                float old15[3] = {kLowest, kLowest, kLowest};  // column 15 of the row before the newest
                uint32_t acc[kAccs] = {0u, 0u, 0u};
                uint32_t slot = 0;  // ring slot of the row this lane processes next (= row mod L)

                for(uint32_t kbase = 0; kbase < nsteps; kbase += kWave) {
                    // ---- chunk: what lane 0 needs at steps kbase .. kbase+63 (body rows of the same index)
                    __builtin_amdgcn_s_barrier();  // (single wave: orders the LDS chunk rewrite after its last reads)
                    {
                        const uint32_t row = kbase + lane;
                        float* cb = chunk + lane * kChunkStride;
                        if(strip == 0) {
                            // column `start` of the matrix: cell (row + L - 1, L - 1); everything left of it is lowest
                            float m, d, in;
                            margin_mdi(k, L, row + L - 1, L - 1, m, d, in);
                            cb[0] = m;
                            cb[1] = d;
                            cb[2] = in;
                            for(uint32_t q = 0; q < 2 * L; ++q) cb[3 + q] = kLowest;
                        } else if(row < la) {
                            const float* src = bbuf + static_cast<uint64_t>(row) * bstride;  // entry `row` = body row row-1
                            cb[0] = src[0];
                            cb[1] = src[1];
                            cb[2] = src[2];
                            const float* cur = bbuf + static_cast<uint64_t>(row + 1) * bstride;  // body row `row`
                            for(uint32_t q = 0; q < 2 * L; ++q) cb[3 + q] = cur[3 + q];
                        }
                    }
                    __builtin_amdgcn_s_barrier();
                    uint32_t a_row = 0;
                    const uint32_t kend = min(static_cast<uint32_t>(kWave), nsteps - kbase);
                    for(uint32_t kk = 0; kk < kend; ++kk) {
                        const uint32_t kstep = kbase + kk;
                        const int r = static_cast<int>(kstep) - lane;  // body row of this lane at this step
                        if(r == 0) {
                            // the lane starts: rows -L..-1 are matrix rows 0..L-1 (align_pair.cc:82-91)
                            for(uint32_t q = 0; q < L; ++q) {
#pragma unroll
                                for(int c = 0; c < kW; ++c) {
                                    const uint32_t col = lane * kW + c;
                                    float m, d, in;
                                    margin_mdi(k, L, q, col0 + col + L, m, d, in);  // matrix row q, matrix col bj+L
                                    ring[(q * 3 + 0) * kStrip + col] = m;
                                    ring[(q * 3 + 1) * kStrip + col] = d;
                                    ring[(q * 3 + 2) * kStrip + col] = in;
                                    if(q == L - 1) {
                                        Mp[c] = m;
                                        Dp[c] = d;
                                        Ip[c] = in;
                                    }
                                }
                            }
                            old15[0] = Mp[kW - 1];
                            old15[1] = Dp[kW - 1];
                            old15[2] = Ip[kW - 1];
                            slot = 0;
                            if(!last_strip && lane == kWave - 1) {  // entry 0 of the boundary: the margin row
                                bbuf[0] = Mp[kW - 1];
                                bbuf[1] = Dp[kW - 1];
                                bbuf[2] = Ip[kW - 1];
                            }
                        }
                        // ---- hand-off: the diagonal cell of column 0 comes from the left neighbour (DPP)
                        const float* cb = chunk + kk * kChunkStride;
                        float dgM = shift_in(old15[0], cb[0]);
                        float dgD = shift_in(old15[1], cb[1]);
                        float dgI = shift_in(old15[2], cb[2]);
                        if(lane == 0)
                            for(uint32_t q = 0; q < L; ++q) {  // lane 0's left neighbours of this row
                                rowbuf[q] = cb[3 + 2 * q];
                                rowbuf[kRowBufCols + q] = cb[3 + 2 * q + 1];
                            }
                        old15[0] = Mp[kW - 1];
                        old15[1] = Dp[kW - 1];
                        old15[2] = Ip[kW - 1];
                        const uint32_t rr = r < 0 ? 0u : (static_cast<uint32_t>(r) < la ? static_cast<uint32_t>(r) : la - 1);
                        a_row = static_cast<uint32_t>(a[rr]) * kTabStride;
#pragma unroll
                        for(int c = 0; c < kW; ++c) {
                            const uint32_t col = lane * kW + c;
                            const float s = tab[a_row + bcode[c]];
                            const float upM = ring[(slot * 3 + 0) * kStrip + col], upD = ring[(slot * 3 + 1) * kStrip + col],
                                        upI = ring[(slot * 3 + 2) * kStrip + col];
                            const float lfM = rowbuf[col], lfI = rowbuf[kRowBufCols + col];
                            // align_pair.cc:97-124
                            const float m2m = ((dgM + k.ng) + k.ng) + s;
                            const float d2m = (dgD + k.gs) + s;
                            const float i2m = ((dgI + k.gs) + k.ng) + s;
                            const float m2d = ((upM + k.ng) + k.go) + ext_lm1;
                            const float i2d = ((upI + k.gs) + k.go) + ext_lm1;
                            const float d2d = upD + ext_l;
                            const float m2i = (lfM + k.go) + ext_lm1;
                            const float i2i = lfI + ext_l;
                            const float M = plus<kLog>(plus<kLog>(m2m, d2m, exp_tab), i2m, exp_tab);
                            const float D = plus<kLog>(plus<kLog>(m2d, d2d, exp_tab), i2d, exp_tab);
                            const float I = plus<kLog>(m2i, i2i, exp_tab);
                            dgM = Mp[c];
                            dgD = Dp[c];
                            dgI = Ip[c];
                            Mp[c] = M;
                            Dp[c] = D;
                            Ip[c] = I;
                            ring[(slot * 3 + 0) * kStrip + col] = M;
                            ring[(slot * 3 + 1) * kStrip + col] = D;
                            ring[(slot * 3 + 2) * kStrip + col] = I;
                            rowbuf[col + L] = M;
                            rowbuf[kRowBufCols + col + L] = I;
                            if constexpr(kLog) {
                                *reinterpret_cast<Mdi*>(mout + (static_cast<uint64_t>(kstep) * kW + c) * (3 * kWave)) = Mdi{M, D, I};
                            } else {
                                // the five decisions of align_pair.cc:275-296 on this cell (common.hpp layout)
                                const float x1 = (M + k.ng) + k.ng, x2 = D + k.gs, x3 = (I + k.gs) + k.ng;
                                const float y1 = (M + k.ng) + k.go, y2 = D + k.ge, y3 = (I + k.gs) + k.go;
                                // max_mdi = arg-max, ties M over D over I: "M is not the max", "D is not the max"
                                const float xx = fmaxf(fmaxf(x1, x2), x3), yy = fmaxf(fmaxf(y1, y2), y3);
                                acc[ACC_A] = (acc[ACC_A] << 2) | (x1 < xx ? 2u : 0u) | (x2 < xx ? 1u : 0u);
                                acc[ACC_B] = (acc[ACC_B] << 2) | (y1 < yy ? 2u : 0u) | (y2 < yy ? 1u : 0u);
                                acc[ACC_C] = (acc[ACC_C] << 1) | ((M + k.go) > (I + k.ge) ? 1u : 0u);
                            }
                        }
                        slot = slot + 1 == L ? 0u : slot + 1;
                        if constexpr(!kLog) {
                            uint32_t* dst = fout + static_cast<uint64_t>(kstep >> 1) * kPairDwords + (kstep & 1u) * (2 * kWave);
                            dst[0] = acc[ACC_A];
                            dst[kWave] = acc[ACC_B];
                            if(kstep & 1u) dst[2 * kWave] = acc[ACC_C];
                        }
                        if(!last_strip && lane == kWave - 1 && r >= 0 && r < static_cast<int>(la)) {
                            float* dst = bbuf + static_cast<uint64_t>(r + 1) * bstride;
                            dst[0] = Mp[kW - 1];
                            dst[1] = Dp[kW - 1];
                            dst[2] = Ip[kW - 1];
                            for(uint32_t q = 0; q < L; ++q) {  // M, I of the strip's last L columns of row r
                                dst[3 + 2 * q] = rowbuf[kStrip + q];
                                dst[3 + 2 * q + 1] = rowbuf[kRowBufCols + kStrip + q];
                            }
                        }
                        if(last_strip && r == static_cast<int>(la) - 1 && lane == last_lane) {
#pragma unroll
                            for(int c = 0; c < kW; ++c)
                                if(c == last_c) {
                                    last_m = Mp[c];
                                    last_d = Dp[c];
                                    last_i = Ip[c];
                                }
                        }
                    }
                }
                if constexpr(!kLog)
                    if(nsteps & 1u) fout[static_cast<uint64_t>(nsteps >> 1) * kPairDwords + 4 * kWave] = acc[ACC_C] << 16;
            }
        }
        // ---- terminal state (align_pair.cc:130-138) from the unadjusted last cell
        int owner = 0;
        if(la > 0 && lb > 0) {
            owner = static_cast<int>(((lb - 1) & (kStrip - 1)) / kW);
        } else {
            margin_mdi(k, L, la + L - 1, lb + L - 1, last_m, last_d, last_i);
        }
        last_m = read_lane(last_m, owner);
        last_d = read_lane(last_d, owner);
        last_i = read_lane(last_i, owner);
        if constexpr(kLog) {
            if(lane == 0) {
                final_mdi[3 * static_cast<uint64_t>(pair) + 0] = (last_m + k.ng) + k.ng;
                final_mdi[3 * static_cast<uint64_t>(pair) + 1] = last_d + k.gs;
                final_mdi[3 * static_cast<uint64_t>(pair) + 2] = (last_i + k.gs) + k.ng;
            }
        } else {
            float score;
            const int start_state = terminal_state(k, last_m, last_d, last_i, score);
            if(lane == 0) scores[pair] = score;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the walk reads this wave's own stores
            walk_pair(lane, k, L, pd, pair, start_state, flags, ops, ops_start, ops_len);
        }
    }
}

size_t generic_lds_bytes(uint32_t L) {
    return sizeof(float) * (kTabRows * kTabStride + static_cast<size_t>(L) * 3 * kStrip + 2 * kRowBufCols +
                            kWave * (3 + 2 * kMaxGapLen));
}

}  // namespace

hipError_t launch_dp_generic(const BatchDeviceView& v, bool forward, hipStream_t stream) {
    if(v.gap_len < 1 || v.gap_len > kMaxGapLen) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(v.queue, 0, sizeof(uint32_t), stream);
    if(e != hipSuccess) return e;
    const size_t lds = generic_lds_bytes(v.gap_len);
    const void* fn = forward ? reinterpret_cast<const void*>(dp_generic<true>) : reinterpret_cast<const void*>(dp_generic<false>);
    if(lds > 48 * 1024) {
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if(e != hipSuccess) return e;
    }
    const uint32_t per_cu = static_cast<uint32_t>(std::min<size_t>(8, (160 * 1024) / lds));
    const uint32_t grid = std::min<uint32_t>(v.n_pairs, device_cu_count() * std::max(1u, per_cu));
    if(forward)
        hipLaunchKernelGGL(dp_generic<true>, dim3(grid), dim3(kWave), lds, stream, v.table, v.k, v.gap_len, v.pairs, v.order,
                           v.n_pairs, v.queue, v.a_cat, v.b_cat, v.flags, v.bnd, v.scores, v.ops, v.ops_start, v.ops_len,
                           v.mdi, v.final_mdi);
    else
        hipLaunchKernelGGL(dp_generic<false>, dim3(grid), dim3(kWave), lds, stream, v.table, v.k, v.gap_len, v.pairs, v.order,
                           v.n_pairs, v.queue, v.a_cat, v.b_cat, v.flags, v.bnd, v.scores, v.ops, v.ops_start, v.ops_len,
                           v.mdi, v.final_mdi);
    return hipGetLastError();
}

}  // namespace coati_hip_detail
