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
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
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
constexpr int kPlanes = 5;                   // decision bit-planes per cell
constexpr int kPairDwords = kPlanes * kWave; // dwords one pair of wavefront steps stores (320)
constexpr int kFillWaves = 4;                // sequence pairs per workgroup
constexpr float kLowest = -FLT_MAX;          // semiring zero(), semiring.hpp:83,113

// Decision bit-planes.  Bit = 1 means:
//   P_M1: D beats M after a match move     (max_mdi first test,  align_pair.cc:213-216)
//   P_M2: I beats max(M,D) after a match   (max_mdi second test, align_pair.cc:217-219)
//   P_D1, P_D2: the same two tests after a deletion move (align_pair.cc:285-287)
//   P_IM: M beats I after an insertion move (max_mi, align_pair.cc:230-232; tie -> I)
enum : int { P_M1 = 0, P_M2 = 1, P_D1 = 2, P_D2 = 3, P_IM = 4 };

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

// HBM layout of the decision bits of one strip (1024 columns) of one pair:
//   dword[(k >> 1) * 320 + plane * 64 + lane], k = wavefront step = body_row + lane
//   bits 31..16 = step k even, bits 15..0 = step k odd; inside a half, the cell
//   of the lane's column c (0..15) is bit 15 - c.
// A wavefront writes five fully coalesced 256-byte rows every two steps:
// 5 bits per DP cell.
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
// acc = (acc << 1) | signbit(d): one v_alignbit_b32
__device__ __forceinline__ uint32_t push_sign(uint32_t acc, float d) {
    return __builtin_amdgcn_alignbit(acc, __builtin_bit_cast(uint32_t, d), 31);
}

// Register state of one lane: its 16 columns of the row it processed last.
struct LaneState {
    float X[kW];  // max((M+ng)+ng, D+gs, (I+gs)+ng): feeds M of the next diagonal cell
    float Y[kW];  // max((M+ng)+go, D+ge, (I+gs)+go): the D value of the cell below (gap_len 1)
    float xlast_old;  // X[15] of the row before: the right neighbour's diagonal input
    float zlast;      // max(M+go, I+ge) of column 15: the right neighbour's I value
    uint32_t acc[kPlanes];  // decision bits, shifted in cell by cell
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
template <int C>
__device__ __forceinline__ void cell_l1(const GapConsts& k, LaneState& st, float& diag, float& zl, float s) {
    const float M = diag + s;
    const float D = st.Y[C];
    const float I = zl;
    diag = st.X[C];
    const float m1 = M + k.ng;
    const float x1 = m1 + k.ng;
    const float y1 = m1 + k.go;
    const float z1 = M + k.go;
    const float x2 = D + k.gs;
    const float y2 = D + k.ge;
    const float i1 = I + k.gs;
    const float x3 = i1 + k.ng;
    const float y3 = i1 + k.go;
    const float z2 = I + k.ge;
    const float xm = fmaxf(x1, x2);
    const float ym = fmaxf(y1, y2);
    st.X[C] = fmaxf(xm, x3);
    st.Y[C] = fmaxf(ym, y3);
    zl = fmaxf(z1, z2);
    st.acc[P_M1] = push_sign(st.acc[P_M1], x1 - x2);  // x2 > x1
    st.acc[P_M2] = push_sign(st.acc[P_M2], xm - x3);  // x3 > max(x1,x2)
    st.acc[P_D1] = push_sign(st.acc[P_D1], y1 - y2);
    st.acc[P_D2] = push_sign(st.acc[P_D2], ym - y3);
    st.acc[P_IM] = push_sign(st.acc[P_IM], z2 - z1);  // z1 > z2
}

template <int... C>
__device__ __forceinline__ void row_l1(const GapConsts& k, LaneState& st, float diag, float zl,
                                       const float (&s)[kW], std::integer_sequence<int, C...>) {
    st.xlast_old = st.X[kW - 1];
    (cell_l1<C>(k, st, diag, zl, s[C]), ...);
    st.zlast = zl;
}

// Viterbi fill for gap_len == 1.  grid = ceil(n_pairs / 4) workgroups of 4
// waves; wave w of block b owns pair 4b+w.
__global__ __launch_bounds__(kFillWaves* kWave, 4) void viterbi_fill_l1(
    const float* __restrict__ table, GapConsts k, const PairDesc* __restrict__ pairs,
    uint32_t n_pairs, const uint8_t* __restrict__ a_cat, const uint8_t* __restrict__ b_cat,
    uint32_t* __restrict__ flags, float* __restrict__ bnd, float* __restrict__ scores) {
    __shared__ float tab[kTabRows * kTabStride];
    for(int idx = threadIdx.x; idx < kTabRows * kTabCols; idx += blockDim.x) {
        const int r = idx / kTabCols, c = idx - r * kTabCols;
        tab[r * kTabStride + c] = table[idx];
    }
    __syncthreads();

    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t pair = blockIdx.x * kFillWaves + (threadIdx.x >> 6);
    if(pair >= n_pairs) return;
    const PairDesc pd = pairs[pair];
    const uint32_t la = pd.la, lb = pd.lb;
    if(la == 0 || lb == 0) return;  // no body cells: the walker handles the margins
    const uint8_t* __restrict__ a = a_cat + pd.a_off;
    const uint8_t* __restrict__ b = b_cat + pd.b_off;
    const char* tab_bytes = reinterpret_cast<const char*>(tab);

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

        LaneState st;
#pragma unroll
        for(int c = 0; c < kW; ++c) st.X[c] = st.Y[c] = 0.0f;
#pragma unroll
        for(int p = 0; p < kPlanes; ++p) st.acc[p] = 0u;
        st.xlast_old = 0.0f;
        st.zlast = 0.0f;
        uint32_t arow = 0;

        for(uint32_t kbase = 0; kbase < nsteps; kbase += kWave) {
            // ---- per-64-step chunk: lane l fetches what lane 0 will need at step kbase+l
            const uint32_t crow = kbase + lane;
            uint32_t a_chunk = 0;
            float bx = kLowest, bz = kLowest;
            if(crow < la) {
                a_chunk = static_cast<uint32_t>(a[crow]) * (kTabStride * 4u);
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
            const uint32_t kend = min(static_cast<uint32_t>(kWave), nsteps - kbase);
            for(uint32_t kk = 0; kk < kend; ++kk) {
                const uint32_t kstep = kbase + kk;
                if(kbase == 0 && kk == static_cast<uint32_t>(lane)) {
                    // This lane starts now: state of the margin row (matrix row 0,
                    // align_pair.cc:88-90): M = D = lowest, I = go + ge*float(j-1).
#pragma unroll
                    for(int c = 0; c < kW; ++c) {
                        const uint32_t bj = col0 + lane * kW + c;
                        const float im = k.go + k.ge * static_cast<float>(bj);
                        const float i1 = im + k.gs;
                        st.X[c] = i1 + k.ng;
                        st.Y[c] = i1 + k.go;
                    }
                    if(!last_strip && lane == kWave - 1) bnd_x[0] = st.X[kW - 1];
                }
                // ---- hand-off from the left neighbour (full exec)
                const float diag = shift_in(st.xlast_old, read_lane(bx, kk));
                const float zl = shift_in(st.zlast, read_lane(bz, kk));
                arow = shift_in(arow, read_lane(a_chunk, kk));
                // ---- substitution scores of this lane's 16 cells
                float s[kW];
#pragma unroll
                for(int c = 0; c < kW; ++c)
                    s[c] = *reinterpret_cast<const float*>(tab_bytes + arow + boff[c]);
                // ---- the 16 cells
                row_l1(k, st, diag, zl, s, std::make_integer_sequence<int, kW>{});
                // ---- every second step: five coalesced 256-byte rows of decision bits
                if(kstep & 1u) {
                    uint32_t* dst = fout + static_cast<uint64_t>(kstep >> 1) * kPairDwords;
#pragma unroll
                    for(int p = 0; p < kPlanes; ++p) dst[p * kWave] = st.acc[p];
                }
                const int r = static_cast<int>(kstep) - lane;  // body row this lane just did
                if(!last_strip && lane == kWave - 1 && r >= 0 && r < static_cast<int>(la)) {
                    bnd_x[r + 1] = st.X[kW - 1];
                    bnd_z[r] = st.zlast;
                }
                if(last_strip && r == static_cast<int>(la) - 1 &&
                   lane == static_cast<int>(((lb - 1) & (kStrip - 1)) / kW)) {
                    // score = max(M,D,I) of the terminal-adjusted last cell
                    // (align_pair.cc:130-138,265) = X of the last body cell.
                    const int cl = (lb - 1) & (kW - 1);
                    float sc = st.X[0];
#pragma unroll
                    for(int c = 1; c < kW; ++c) sc = (c == cl) ? st.X[c] : sc;
                    scores[pair] = sc;
                }
            }
        }
        if(nsteps & 1u) {  // the last (even) step has no odd partner: flush it to the high half
            uint32_t* dst = fout + static_cast<uint64_t>(nsteps >> 1) * kPairDwords;
#pragma unroll
            for(int p = 0; p < kPlanes; ++p) dst[p * kWave] = st.acc[p] << 16;
        }
    }
}

// ---------------------------------------------------------------------------
// Traceback walker (traceback<tropical>, align_pair.cc:249-303, gap_len 1):
// one lane per pair.  Emits ops right-to-left into the pair's slot so that they
// end up in left-to-right order at [ops_start, ops_start + ops_len).
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t plane_bit(const uint32_t* __restrict__ flags, uint64_t base,
                                              uint32_t la, uint32_t bi, uint32_t bj, int plane) {
    const uint32_t strip = bj / kStrip, t = (bj % kStrip) / kW, c = bj % kW;
    const uint32_t kstep = bi + t;
    const uint64_t d = base + strip * strip_dwords(la) + static_cast<uint64_t>(kstep >> 1) * kPairDwords +
                       plane * kWave + t;
    const uint32_t bit = ((kstep & 1u) ? 0u : 16u) + (kW - 1 - c);
    return (flags[d] >> bit) & 1u;
}

__device__ __forceinline__ int state_after(const uint32_t* __restrict__ flags, uint64_t base,
                                           uint32_t la, uint32_t bi, uint32_t bj, int moved) {
    if(moved == COATI_HIP_OP_MATCH) {
        if(plane_bit(flags, base, la, bi, bj, P_M2)) return COATI_HIP_OP_INS;
        return plane_bit(flags, base, la, bi, bj, P_M1) ? COATI_HIP_OP_DEL : COATI_HIP_OP_MATCH;
    }
    if(moved == COATI_HIP_OP_DEL) {
        if(plane_bit(flags, base, la, bi, bj, P_D2)) return COATI_HIP_OP_INS;
        return plane_bit(flags, base, la, bi, bj, P_D1) ? COATI_HIP_OP_DEL : COATI_HIP_OP_MATCH;
    }
    return plane_bit(flags, base, la, bi, bj, P_IM) ? COATI_HIP_OP_MATCH : COATI_HIP_OP_INS;
}

__global__ __launch_bounds__(64) void viterbi_walk_l1(
    GapConsts k, const PairDesc* __restrict__ pairs, uint32_t n_pairs,
    const uint32_t* __restrict__ flags, uint8_t* __restrict__ ops, uint64_t* __restrict__ ops_start,
    uint32_t* __restrict__ ops_len, float* __restrict__ scores) {
    const uint32_t pair = blockIdx.x * blockDim.x + threadIdx.x;
    if(pair >= n_pairs) return;
    const PairDesc pd = pairs[pair];
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
        scores[pair] = fmaxf(fmaxf(tm, td), ti);
        st = la > 0 ? COATI_HIP_OP_DEL : COATI_HIP_OP_INS;
    } else {
        // max_mdi of the terminal-adjusted last cell == its "after match" decision
        st = state_after(flags, pd.flags_off, la, la - 1, lb - 1, COATI_HIP_OP_MATCH);
    }
    while(i > 0 || j > 0) {
        ops[--pos] = static_cast<uint8_t>(st);
        if(st == COATI_HIP_OP_MATCH) {
            --i;
            --j;
        } else if(st == COATI_HIP_OP_DEL) {
            --i;
        } else {
            --j;
        }
        if(i == 0 && j == 0) break;
        if(j == 0) {
            st = COATI_HIP_OP_DEL;  // column 0: only D is finite (align_pair.cc:82-86)
        } else if(i == 0) {
            st = COATI_HIP_OP_INS;  // row 0: only I is finite (align_pair.cc:88-90)
        } else {
            st = state_after(flags, pd.flags_off, la, i - 1, j - 1, st);
        }
    }
    ops_start[pair] = pos;
    ops_len[pair] = static_cast<uint32_t>(pd.ops_off + la + lb - pos);
}

// Debug: decode one pair's bit-planes into the oracle's byte-per-cell encoding.
__global__ void decode_flags(const PairDesc* __restrict__ pairs, uint32_t pair,
                             const uint32_t* __restrict__ flags, uint8_t* __restrict__ out) {
    const PairDesc pd = pairs[pair];
    const uint64_t n = static_cast<uint64_t>(pd.la) * pd.lb;
    for(uint64_t idx = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; idx < n;
        idx += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint32_t bi = idx / pd.lb, bj = idx % pd.lb;
        const uint32_t m1 = plane_bit(flags, pd.flags_off, pd.la, bi, bj, P_M1);
        const uint32_t m2 = plane_bit(flags, pd.flags_off, pd.la, bi, bj, P_M2);
        const uint32_t d1 = plane_bit(flags, pd.flags_off, pd.la, bi, bj, P_D1);
        const uint32_t d2 = plane_bit(flags, pd.flags_off, pd.la, bi, bj, P_D2);
        const uint32_t im = plane_bit(flags, pd.flags_off, pd.la, bi, bj, P_IM);
        const uint32_t fm = m2 ? 2u : m1, fd = d2 ? 2u : d1;
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
    uint8_t *d_a = nullptr, *d_b = nullptr, *d_ops = nullptr;
    uint32_t* d_flags = nullptr;
    float *d_bnd = nullptr, *d_scores = nullptr;
    uint64_t* d_ops_start = nullptr;
    uint32_t* d_ops_len = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
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
    void* ptrs[] = {b->d_desc, b->d_a,      b->d_b,         b->d_ops,    b->d_flags,
                    b->d_bnd,  b->d_scores, b->d_ops_start, b->d_ops_len};
    for(void* p : ptrs)
        if(p != nullptr) (void)hipFree(p);
    for(hipEvent_t e : b->ev)
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
    if(n_pairs > 0)
        B_TRY(hipMemcpy(b->d_desc, b->desc.data(), n_pairs * sizeof(PairDesc), hipMemcpyHostToDevice));
    if(a_total > 0) B_TRY(hipMemcpy(b->d_a, a_cat + a_off[0], a_total, hipMemcpyHostToDevice));
    if(b_total > 0) B_TRY(hipMemcpy(b->d_b, b_cat + b_off[0], b_total, hipMemcpyHostToDevice));
    for(auto& e : b->ev) B_TRY(hipEventCreate(&e));
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
    HIP_TRY(hipEventRecord(b->ev[0], m->stream));
    if(n > 0) {
        const uint32_t grid = (n + kFillWaves - 1) / kFillWaves;
        hipLaunchKernelGGL(viterbi_fill_l1, dim3(grid), dim3(kFillWaves * kWave), 0, m->stream,
                           m->d_table, m->k, b->d_desc, n, b->d_a, b->d_b, b->d_flags, b->d_bnd,
                           b->d_scores);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(b->ev[1], m->stream));
    if(n > 0) {
        hipLaunchKernelGGL(viterbi_walk_l1, dim3((n + 63) / 64), dim3(64), 0, m->stream, m->k, b->d_desc, n,
                           b->d_flags, b->d_ops, b->d_ops_start, b->d_ops_len, b->d_scores);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(b->ev[2], m->stream));
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

int coati_hip_viterbi_last_timing(coati_hip_batch_t* b, float* fill_ms, float* walk_ms) {
    if(b == nullptr) return fail(COATI_HIP_EINVAL, "last_timing: batch is NULL");
    if(!b->launched) return fail(COATI_HIP_ESTATE, "last_timing: nothing was launched");
    int rc = coati_hip_batch_sync(b);
    if(rc != COATI_HIP_OK) return rc;
    float f = 0.f, w = 0.f;
    HIP_TRY(hipEventElapsedTime(&f, b->ev[0], b->ev[1]));
    HIP_TRY(hipEventElapsedTime(&w, b->ev[1], b->ev[2]));
    if(fill_ms != nullptr) *fill_ms = f;
    if(walk_ms != nullptr) *walk_ms = w;
    return COATI_HIP_OK;
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
