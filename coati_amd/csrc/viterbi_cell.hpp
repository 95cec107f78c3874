// viterbi_cell.hpp -- the gap_len-1 Viterbi cell shared by viterbi_l1.hip (fill that deposits five
// decision bits per cell) and viterbi_ck.hip (its tile recompute): lane state, the hand-scheduled
// 25-instruction cell, the strip-boundary hand-off between wavefronts.
#ifndef COATI_HIP_VITERBI_CELL_HPP
#define COATI_HIP_VITERBI_CELL_HPP

#include "common.hpp"

#include <utility>

namespace coati_hip_detail {
namespace {

// Register state of one lane: its W columns (W = 16, 8 or 4) of the row it processed last.
template <int W>
struct LaneState {
    float X[W];  // max((M+ng)+ng, D+gs, (I+gs)+ng): feeds M of the next diagonal cell
    float Y[W];  // max((M+ng)+go, D+ge, (I+gs)+go): the D value of the cell below (gap_len 1)
    float xlast_old;  // X[W-1] of the row before: the right neighbour's diagonal input
    float zlast;      // max(M+go, I+ge) of column W-1: the right neighbour's I value
    uint32_t acc[kAccs];    // decision bits, shifted in cell by cell
};

// One DP cell for gap_len == 1 (align_pair.cc:97-124 with look_back = 1, where
// power(gap_extend, 0) is -0.0f and adding it is the identity).  Because fp32
// addition is monotone, max(x1+s, x2+s, x3+s) == max(x1,x2,x3)+s bit for bit,
// so M = X(diagonal cell) + s.
//
// The five decisions are max_mdi / max_mi (align_pair.cc:210-232) on the
// expressions of align_pair.cc:275-296.  max_mdi(x1,x2,x3) is the arg-max with ties
// M over D over I, so two bits suffice: "x1 is not the maximum" and "x2 is not the
// maximum" (M if the first is clear, else D if the second is clear, else I).  With
// X = max3(x1,x2,x3) each of them is the sign bit of x - X: exact, because fp32
// subtraction of two finite numbers is zero only when they are equal (gradual
// underflow is on; x - x = +0.0f).  Likewise z1 > z2 is the sign of z2 - z1.  On
// gfx950 v_sub_f32 issues at twice the rate of v_cmp_f32 and the bit is
// deposited with a single v_alignbit_b32 (measured: tools/ubench).
// One DP cell = ONE asm block of 25 VALU instructions with a fixed order and a
// hand register allocation.  Why not leave it to the compiler (all measured or
// observed):
//  * on gfx950 v_add/v_sub_f32 and v_add_u32 issue every 2 cycles, v_max_f32, v_max3_f32 and
//    v_alignbit_b32 have a 4-cycle initiation interval, and a 4-cycle op that ALTERNATES with
//    2-cycle ops costs ~9 cycles per pair instead of 6 (tools/ubench: "mix add,max 1:1").  The
//    eight slow ops of the cell therefore sit in two runs (max, max3, max3, [alignbit,] alignbit
//    and alignbit x 3) between two runs of adds/subs; a register-only replay of this order runs at
//    31.2 ns per 64-lane cell at 3 waves/SIMD against 33.0 for the earlier alternating order
//    (25.0 against 31.4 at 2 waves/SIMD), the kernel gained 3.3 %.  Every consumer is >= 3
//    instructions behind its producer.
//  * hipcc batches the X/Y maxes of a whole row (32 back-to-back v_max), and once
//    values are opaque adds canonicalising v_max around every fmaxf (IEEE mode);
//    between adjacent dependent inline-asm statements the hazard recognizer
//    inserts s_nop.  One block per cell has none of that.
// No instruction here has a software-visible hazard (no trans ops, no DPP or
// readlane consumer inside).  The deposit of the cell's last decision (D2) is
// carried in `pend` into the next cell; the LDS address of this column's score
// for the NEXT wavefront step is computed here, the ds_read is issued by the
// compiler right after the block (so that it also places the s_waitcnt).
#define COATI_CELL_FAST_A                                                                   \
    "v_add_f32 %[t0], %[diag], %[s]\n\t"      /* F  M  = diag + s                        */ \
    "v_add_f32 %[t1], %[ge], %[zl]\n\t"       /* F  z2 = I + ge                          */ \
    "v_add_f32 %[t2], %[gs], %[zl]\n\t"       /* F  i1 = I + gs                          */ \
    "v_add_f32 %[t3], %[go], %[t0]\n\t"       /* F  z1 = M + go                          */ \
    "v_add_f32 %[t0], %[ng], %[t0]\n\t"       /* F  m1 = M + ng                          */ \
    "v_add_f32 %[t5], %[gs], %[y]\n\t"        /* F  x2 = D + gs                          */ \
    "v_add_f32 %[t8], %[ge], %[y]\n\t"        /* F  y2 = D + ge                          */ \
    "v_add_f32 %[t4], %[ng], %[t0]\n\t"       /* F  x1 = m1 + ng                         */ \
    "v_add_f32 %[t6], %[ng], %[t2]\n\t"       /* F  x3 = i1 + ng                         */ \
    "v_add_f32 %[t7], %[go], %[t0]\n\t"       /* F  y1 = m1 + go                         */ \
    "v_add_f32 %[t9], %[go], %[t2]\n\t"       /* F  y3 = i1 + go                         */ \
    "v_sub_f32 %[t10], %[t1], %[t3]\n\t"      /* F  z2 - z1  (sign: z1 > z2)             */ \
    "v_add_u32 %[addr], %[lds], %[boff]\n\t"  /* F  LDS address of next step's score     */ \
    "v_max_f32 %[zl], %[t3], %[t1]\n\t"       /* S  Z  = max(z1,z2) -> I of next column  */ \
    "v_max3_f32 %[x], %[t4], %[t5], %[t6]\n\t" /* S  X  = max(x1,x2,x3)                  */ \
    "v_max3_f32 %[y], %[t7], %[t8], %[t9]\n\t" /* S  Y  = max(y1,y2,y3)                  */
#define COATI_CELL_PEND                                                                     \
    "v_alignbit_b32 %[aB], %[aB], %[pend], 31\n\t" /* S  D2 of the previous cell         */
#define COATI_CELL_TAIL                                                                     \
    "v_alignbit_b32 %[aC], %[aC], %[t10], 31\n\t" /* S  IM                               */ \
    "v_sub_f32 %[t4], %[t4], %[x]\n\t"        /* F  x1 - X   (sign: x1 is not the max)   */ \
    "v_sub_f32 %[t5], %[t5], %[x]\n\t"        /* F  x2 - X   (sign: x2 is not the max)   */ \
    "v_sub_f32 %[t7], %[t7], %[y]\n\t"        /* F  y1 - Y   (sign: y1 is not the max)   */ \
    "v_sub_f32 %[pend], %[t8], %[y]\n\t"      /* F  y2 - Y, carried into the next cell   */ \
    "v_alignbit_b32 %[aA], %[aA], %[t4], 31\n\t" /* S  M1                                */ \
    "v_alignbit_b32 %[aA], %[aA], %[t5], 31\n\t" /* S  M2                                */ \
    "v_alignbit_b32 %[aB], %[aB], %[t7], 31"    /* S  D1                                   */

template <int C, int W>
__device__ __forceinline__ void cell_l1(const GapVec& k, LaneState<W>& st, float& diag, float& zl, float& pend,
                                        float& s, uint32_t lds_next_row, uint32_t boff) {
    float x_new, t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10;
    uint32_t addr;
#define COATI_CELL_OPERANDS                                                                              \
    : [x] "=&v"(x_new), [y] "+v"(st.Y[C]), [zl] "+v"(zl), [pend] "+v"(pend), [aA] "+v"(st.acc[ACC_A]),   \
      [aB] "+v"(st.acc[ACC_B]), [aC] "+v"(st.acc[ACC_C]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), \
      [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7), [t8] "=&v"(t8),     \
      [t9] "=&v"(t9), [t10] "=&v"(t10), [addr] "=&v"(addr)                                                \
    : [diag] "v"(diag), [s] "v"(s), [lds] "v"(lds_next_row), [boff] "v"(boff), [ng] "v"(k.ng),          \
      [gs] "v"(k.gs), [go] "v"(k.go), [ge] "v"(k.ge)
    if constexpr(C > 0) {
        asm volatile(COATI_CELL_FAST_A COATI_CELL_PEND COATI_CELL_TAIL COATI_CELL_OPERANDS);
    } else {
        asm volatile(COATI_CELL_FAST_A COATI_CELL_TAIL COATI_CELL_OPERANDS);
    }
#undef COATI_CELL_OPERANDS
    diag = st.X[C];  // the next column's diagonal input is this column's previous-row X
    st.X[C] = x_new;
    // s was consumed by the block's first instruction: reuse it for the next step's score
    s = *reinterpret_cast<const __attribute__((address_space(3))) float*>(addr);
}

template <int W, int... C>
__device__ __forceinline__ void row_l1(const GapVec& k, LaneState<W>& st, float diag, float zl,
                                       float (&s)[W], uint32_t lds_next_row, const uint32_t (&boff)[W],
                                       std::integer_sequence<int, C...>) {
    st.xlast_old = st.X[W - 1];
    float pend = 0.0f;
    (cell_l1<C, W>(k, st, diag, zl, pend, s[C], lds_next_row, boff[C]), ...);
    asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(st.acc[ACC_B]) : "v"(pend));  // D2 of the last column
    st.zlast = zl;
}


// Raw buffer descriptor in four SGPRs for the hand-written blocks (viterbi_lp.hip, viterbi_ck.hip): base, no stride,
// `bytes` valid (the VGPR offset is range-checked against it: reads beyond return 0, writes beyond are dropped), 32-bit
// data format.  A lane whose offset register holds kDropOffset stores nothing.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
constexpr uint32_t kDropOffset = 0x80000000u;
__device__ __forceinline__ u32x4_t raw_rsrc(const void* p, uint64_t bytes = 0x7ffffff0ull) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    u32x4_t r;
    r.x = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(a)));
    r.y = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(a >> 32))) & 0xffffu;
    r.z = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(bytes < 0x7ffffff0ull ? bytes : 0x7ffffff0ull)));
    r.w = 0x00020000u;
    return r;
}

// Strip-boundary hand-off between wavefronts (cdna_hip_programming.md Guideline 16, recipe R1):
// the payload is stored write-through (agent-scope relaxed atomic store = `sc1`), the storing
// wave drains (s_waitcnt vmcnt(0)) and ONE lane publishes a progress word; the consumer polls that
// word relaxed and then executes ONE agent-scope acquire before its plain loads.
__device__ __forceinline__ void store_through(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void publish_progress(uint32_t* word, uint32_t rows, bool leader) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if(leader) __hip_atomic_store(word, rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// what a strip publishes instead of its row count when its own left neighbour timed out
constexpr uint32_t kHandoffPoison = 0xffffffffu;
// false if the producer did not get there within the spin bound (never hang the GPU)
__device__ __forceinline__ bool wait_progress(const uint32_t* word, uint32_t need) {
    for(uint32_t spins = 0; __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need; ++spins) {
        if(spins > (1u << 26)) return false;
        __builtin_amdgcn_s_sleep(4);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}

// the same without the acquire: for a consumer that reads what it waited for PAST its L2 (agent-scope loads), viterbi_ck's
// row parts
__device__ __forceinline__ bool wait_progress_relaxed(const uint32_t* word, uint32_t need) {
    for(uint32_t spins = 0; __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need; ++spins) {
        if(spins > (1u << 26)) return false;
        __builtin_amdgcn_s_sleep(4);
    }
    return true;
}

}  // namespace
}  // namespace coati_hip_detail
#endif
