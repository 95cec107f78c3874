#!/usr/bin/env python3
"""Generate tools/ubench/cell_pk.hip: issue-rate test of two codings of the DP cell with a
fully hand-allocated register file (one asm block = ITER wavefront steps of 16 cells):
  base: the 27-instruction scalar-fp32 cell of viterbi_l1.hip;
  pk:   10 adds + 4 subs folded into 7 v_pk_add_f32 (VGPR-pair constants) -> 20 instructions.
No memory traffic; the numbers say what the VALU can issue, nothing else.
"""
from pathlib import Path

NG, GS, GO, GE = "s4", "s5", "s6", "s7"          # base: SGPR constants
# pk: constant pairs in VGPRs (an SGPR-pair operand of v_pk_add_f32 computed wrong values on gfx950)
C1, C2, C3 = "v[2:3]", "v[4:5]", "v[6:7]"        # [gs,ge] [go,ng] [ng,go]
ZL = 8          # v8 = zl (pair v[8:9])
MP = 10         # v10 = M (pair v[10:11])
IP = 12         # [i1, z2]
QP = 14         # [z1, m1]
A, B, Cc, MX, D1, D2 = 16, 18, 20, 22, 24, 26
DZ, DIAG, ADDR, LDS = 28, 29, 30, 31
AA, AB, AC = 32, 33, 34
YP = 40         # pair c: v[40+2c] = Y[c], v[41+2c] = XA[c]
XB = 72         # v[72+c]  = XB[c]
S = 88          # v[88+c]  = s[c]
BOFF = 104      # v[104+c]


def pair(n):
    return f"v[{n}:{n+1}]"


def cell_pk(c, parity):
    xin = f"v{YP+2*c+1}" if parity == 0 else f"v{XB+c}"
    xout = f"v{XB+c}" if parity == 0 else f"v{YP+2*c+1}"
    y = YP + 2 * c
    L = []
    L.append(f"v_add_f32 v{MP}, v{DIAG}, v{S+c}")
    L.append(f"v_pk_add_f32 {pair(IP)}, {pair(ZL)}, {C1} op_sel_hi:[0,1]")       # [zl+gs, zl+ge]
    L.append(f"v_pk_add_f32 {pair(QP)}, {pair(MP)}, {C2} op_sel_hi:[0,1]")       # [M+go, M+ng]
    L.append(f"v_mov_b32 v{DIAG}, {xin}" if False else f"v_max_f32 v{ZL}, v{QP}, v{IP+1}")
    L.append(f"v_pk_add_f32 {pair(A)}, {pair(QP)}, {C3} op_sel:[1,0] op_sel_hi:[1,1]")  # [m1+ng, m1+go]
    L.append(f"v_sub_f32 v{DZ}, v{IP+1}, v{QP}")
    L.append(f"v_pk_add_f32 {pair(B)}, {pair(y)}, {C1} op_sel_hi:[0,1]")         # [D+gs, D+ge]
    L.append(f"v_alignbit_b32 v{AC}, v{AC}, v{DZ}, 31")
    L.append(f"v_pk_add_f32 {pair(Cc)}, {pair(IP)}, {C3} op_sel_hi:[0,1]")       # [i1+ng, i1+go]
    L.append(f"v_max_f32 v{MX}, v{A}, v{B}")
    L.append(f"v_pk_add_f32 {pair(D1)}, {pair(A)}, {pair(B)} neg_lo:[0,1] neg_hi:[0,1]")   # [x1-x2, y1-y2]
    L.append(f"v_max_f32 v{MX+1}, v{A+1}, v{B+1}")
    L.append(f"v_add_u32 v{ADDR}, v{LDS}, v{BOFF+c}")
    L.append(f"v_alignbit_b32 v{AA}, v{AA}, v{D1}, 31")
    L.append(f"v_pk_add_f32 {pair(D2)}, {pair(MX)}, {pair(Cc)} neg_lo:[0,1] neg_hi:[0,1]") # [xm-x3, ym-y3]
    L.append(f"v_mov_b32 v{DIAG}, {xin}")     # next column's diagonal input (register renaming in the real kernel)
    L.append(f"v_max_f32 {xout}, v{MX}, v{Cc}")
    L.append(f"v_alignbit_b32 v{AB}, v{AB}, v{D1+1}, 31")
    L.append(f"v_max_f32 v{y}, v{MX+1}, v{Cc+1}")
    L.append(f"v_alignbit_b32 v{AA}, v{AA}, v{D2}, 31")
    L.append(f"v_alignbit_b32 v{AB}, v{AB}, v{D2+1}, 31")
    L.append(f"v_add_f32 v{S+c}, v{S+c}, v{ADDR}" if False else f"v_xor_b32 v{LDS}, v{LDS}, v{ADDR}")
    return L


def cell_base(c, parity):
    xin = f"v{YP+2*c+1}" if parity == 0 else f"v{XB+c}"
    xout = f"v{XB+c}" if parity == 0 else f"v{YP+2*c+1}"
    y = f"v{YP+2*c}"
    t0, t1, t2, t3, t4, t5, pend, zl = "v10", "v11", "v12", "v13", "v14", "v15", "v16", f"v{ZL}"
    L = [f"v_add_f32 {t0}, v{DIAG}, v{S+c}", f"v_add_f32 {t1}, {GE}, {zl}",
         f"v_alignbit_b32 v{AB}, v{AB}, {pend}, 31",
         f"v_add_f32 {t2}, {GS}, {zl}", f"v_add_f32 {t3}, {GO}, {t0}", f"v_add_f32 {t0}, {NG}, {t0}",
         f"v_max_f32 {zl}, {t3}, {t1}", f"v_add_f32 {t4}, {NG}, {t0}", f"v_add_f32 {t5}, {GS}, {y}",
         f"v_sub_f32 {t1}, {t1}, {t3}", f"v_max_f32 {t3}, {t4}, {t5}", f"v_add_f32 {pend}, {NG}, {t2}",
         f"v_alignbit_b32 v{AC}, v{AC}, {t1}, 31", f"v_sub_f32 {t1}, {t4}, {t5}",
         f"v_mov_b32 v{DIAG}, {xin}",
         f"v_max_f32 {xout}, {t3}, {pend}", f"v_sub_f32 {t4}, {t3}, {pend}", f"v_add_f32 {t5}, {GO}, {t0}",
         f"v_alignbit_b32 v{AA}, v{AA}, {t1}, 31", f"v_add_f32 {t1}, {GE}, {y}", f"v_add_f32 {t3}, {GO}, {t2}",
         f"v_max_f32 {t0}, {t5}, {t1}", f"v_sub_f32 {t2}, {t5}, {t1}", f"v_alignbit_b32 v{AA}, v{AA}, {t4}, 31",
         f"v_add_u32 v{ADDR}, v{LDS}, v{BOFF+c}", f"v_max_f32 {y}, {t0}, {t3}", f"v_sub_f32 {pend}, {t0}, {t3}",
         f"v_alignbit_b32 v{AB}, v{AB}, {t2}, 31", f"v_xor_b32 v{LDS}, v{LDS}, v{ADDR}"]
    return L


def cell_max3(c, parity):
    """25 VALU: X and Y as v_max3_f32; decisions as sign(x1 - X), sign(x2 - X) (zero iff equal)."""
    xin = f"v{YP+2*c+1}" if parity == 0 else f"v{XB+c}"
    xout = f"v{XB+c}" if parity == 0 else f"v{YP+2*c+1}"
    y = f"v{YP+2*c}"
    t0, t1, t2, t3, t4, t5, t6, pend, zl = "v10", "v11", "v12", "v13", "v14", "v15", "v17", "v16", f"v{ZL}"
    L = [f"v_add_f32 {t0}, v{DIAG}, v{S+c}", f"v_add_f32 {t1}, {GE}, {zl}",
         f"v_alignbit_b32 v{AB}, v{AB}, {pend}, 31",
         f"v_add_f32 {t2}, {GS}, {zl}", f"v_add_f32 {t3}, {GO}, {t0}", f"v_add_f32 {t0}, {NG}, {t0}",
         f"v_max_f32 {zl}, {t3}, {t1}", f"v_add_f32 {t4}, {NG}, {t0}", f"v_add_f32 {t5}, {GS}, {y}",
         f"v_add_f32 {t6}, {NG}, {t2}", f"v_sub_f32 {t1}, {t1}, {t3}",
         f"v_mov_b32 v{DIAG}, {xin}",
         f"v_max3_f32 {xout}, {t4}, {t5}, {t6}", f"v_add_f32 {t3}, {GO}, {t0}", f"v_add_f32 {t6}, {GE}, {y}",
         f"v_alignbit_b32 v{AC}, v{AC}, {t1}, 31", f"v_add_f32 {t2}, {GO}, {t2}",
         f"v_sub_f32 {t4}, {t4}, {xout}", f"v_sub_f32 {t5}, {t5}, {xout}",
         f"v_max3_f32 {y}, {t3}, {t6}, {t2}", f"v_add_u32 v{ADDR}, v{LDS}, v{BOFF+c}",
         f"v_alignbit_b32 v{AA}, v{AA}, {t4}, 31", f"v_sub_f32 {t3}, {t3}, {y}",
         f"v_alignbit_b32 v{AA}, v{AA}, {t5}, 31", f"v_sub_f32 {pend}, {t6}, {y}",
         f"v_alignbit_b32 v{AB}, v{AB}, {t3}, 31", f"v_xor_b32 v{LDS}, v{LDS}, v{ADDR}"]
    return L


def cell_clustered(c, parity):
    """The 25 instructions of cell_max3 with the 4-cycle ops (max, max3, alignbit) in two clusters
    instead of interleaved with the 2-cycle ops (three more temporaries)."""
    xin = f"v{YP+2*c+1}" if parity == 0 else f"v{XB+c}"
    xout = f"v{XB+c}" if parity == 0 else f"v{YP+2*c+1}"
    y = f"v{YP+2*c}"
    t0, t1, t2, t3, t4, t5, t6, pend, zl = "v10", "v11", "v12", "v13", "v14", "v15", "v17", "v16", f"v{ZL}"
    t7, t8, t9, t10 = "v18", "v19", "v20", "v21"
    L = [f"v_add_f32 {t0}, v{DIAG}, v{S+c}", f"v_add_f32 {t1}, {GE}, {zl}", f"v_add_f32 {t2}, {GS}, {zl}",
         f"v_add_f32 {t3}, {GO}, {t0}", f"v_add_f32 {t0}, {NG}, {t0}", f"v_add_f32 {t5}, {GS}, {y}",
         f"v_add_f32 {t8}, {GE}, {y}", f"v_add_f32 {t4}, {NG}, {t0}", f"v_add_f32 {t6}, {NG}, {t2}",
         f"v_add_f32 {t7}, {GO}, {t0}", f"v_add_f32 {t9}, {GO}, {t2}", f"v_sub_f32 {t10}, {t1}, {t3}",
         f"v_mov_b32 v{DIAG}, {xin}", f"v_add_u32 v{ADDR}, v{LDS}, v{BOFF+c}",
         # slow cluster A
         f"v_max_f32 {zl}, {t3}, {t1}", f"v_max3_f32 {xout}, {t4}, {t5}, {t6}", f"v_max3_f32 {y}, {t7}, {t8}, {t9}",
         f"v_alignbit_b32 v{AB}, v{AB}, {pend}, 31", f"v_alignbit_b32 v{AC}, v{AC}, {t10}, 31",
         # fast cluster B
         f"v_sub_f32 {t4}, {t4}, {xout}", f"v_sub_f32 {t5}, {t5}, {xout}", f"v_sub_f32 {t7}, {t7}, {y}",
         f"v_sub_f32 {pend}, {t8}, {y}", f"v_xor_b32 v{LDS}, v{LDS}, v{ADDR}",
         # slow cluster B
         f"v_alignbit_b32 v{AA}, v{AA}, {t4}, 31", f"v_alignbit_b32 v{AA}, v{AA}, {t5}, 31",
         f"v_alignbit_b32 v{AB}, v{AB}, {t7}, 31"]
    return L


def _clustered_variant(c, parity, variant):
    xin = f"v{YP+2*c+1}" if parity == 0 else f"v{XB+c}"
    xout = f"v{XB+c}" if parity == 0 else f"v{YP+2*c+1}"
    y = f"v{YP+2*c}"
    t0, t1, t2, t3, t4, t5, t6, pend, zl = "v10", "v11", "v12", "v13", "v14", "v15", "v17", "v16", f"v{ZL}"
    t7, t8, t9, t10 = "v18", "v19", "v20", "v21"
    fast_a = [f"v_add_f32 {t0}, v{DIAG}, v{S+c}", f"v_add_f32 {t1}, {GE}, {zl}", f"v_add_f32 {t2}, {GS}, {zl}",
              f"v_add_f32 {t3}, {GO}, {t0}", f"v_add_f32 {t0}, {NG}, {t0}", f"v_add_f32 {t5}, {GS}, {y}",
              f"v_add_f32 {t8}, {GE}, {y}", f"v_add_f32 {t4}, {NG}, {t0}", f"v_add_f32 {t6}, {NG}, {t2}",
              f"v_add_f32 {t7}, {GO}, {t0}", f"v_add_f32 {t9}, {GO}, {t2}", f"v_sub_f32 {t10}, {t1}, {t3}",
              f"v_mov_b32 v{DIAG}, {xin}", f"v_add_u32 v{ADDR}, v{LDS}, v{BOFF+c}"]
    maxes = [f"v_max_f32 {zl}, {t3}, {t1}", f"v_max3_f32 {xout}, {t4}, {t5}, {t6}", f"v_max3_f32 {y}, {t7}, {t8}, {t9}"]
    a_pend, a_c = f"v_alignbit_b32 v{AB}, v{AB}, {pend}, 31", f"v_alignbit_b32 v{AC}, v{AC}, {t10}, 31"
    subs = [f"v_sub_f32 {t4}, {t4}, {xout}", f"v_sub_f32 {t5}, {t5}, {xout}", f"v_sub_f32 {t7}, {t7}, {y}",
            f"v_sub_f32 {pend}, {t8}, {y}", f"v_xor_b32 v{LDS}, v{LDS}, v{ADDR}"]
    a_m1, a_m2, a_d1 = (f"v_alignbit_b32 v{AA}, v{AA}, {t4}, 31", f"v_alignbit_b32 v{AA}, v{AA}, {t5}, 31",
                        f"v_alignbit_b32 v{AB}, v{AB}, {t7}, 31")
    if variant == 2:   # as `clustered`, the last run reordered so that aA is not written back to back
        return fast_a + maxes + [a_pend, a_c] + subs + [a_m1, a_d1, a_m2]
    if variant == 4:   # three maxes | four subs | all five deposits
        return fast_a + maxes + subs + [a_pend, a_c, a_m1, a_d1, a_m2]
    raise ValueError(variant)


def cell_clustered2(c, parity):
    return _clustered_variant(c, parity, 2)


def cell_clustered4(c, parity):
    return _clustered_variant(c, parity, 4)


def cell_cmp(c, parity):
    """22 VALU: the five decisions as v_cmp into SGPR pairs (lane masks), stored by the scalar unit."""
    xin = f"v{YP+2*c+1}" if parity == 0 else f"v{XB+c}"
    xout = f"v{XB+c}" if parity == 0 else f"v{YP+2*c+1}"
    y = f"v{YP+2*c}"
    t0, t1, t2, t3, t4, t5, x3, zl = "v10", "v11", "v12", "v13", "v14", "v15", "v16", f"v{ZL}"
    m = 40 + 12 * (c & 1)        # mask registers s[m..m+9], two sets alternate (x4 stores need 4-aligned tuples)
    pm = 40 + 12 * ((c + 1) & 1)  # the previous cell's set: stored while this cell computes
    off = (c * 40) % 640
    L = [f"v_add_f32 {t0}, v{DIAG}, v{S+c}", f"v_add_f32 {t1}, {GE}, {zl}",
         f"v_add_f32 {t2}, {GS}, {zl}", f"v_add_f32 {t3}, {GO}, {t0}", f"v_add_f32 {t0}, {NG}, {t0}",
         f"s_store_dwordx4 s[{pm}:{pm+3}], s[16:17], {hex(off)}",
         f"v_max_f32 {zl}, {t3}, {t1}", f"v_add_f32 {t4}, {NG}, {t0}", f"v_add_f32 {t5}, {GS}, {y}",
         f"v_cmp_gt_f32_e64 s[{m}:{m+1}], {t3}, {t1}",
         f"v_add_f32 {x3}, {NG}, {t2}", f"v_max_f32 {t3}, {t4}, {t5}",
         f"s_store_dwordx4 s[{pm+4}:{pm+7}], s[16:17], {hex(off+16)}",
         f"v_cmp_gt_f32_e64 s[{m+2}:{m+3}], {t5}, {t4}",
         f"v_mov_b32 v{DIAG}, {xin}",
         f"v_add_f32 {t1}, {GE}, {y}", f"v_max_f32 {xout}, {t3}, {x3}",
         f"v_add_f32 {t5}, {GO}, {t0}", f"v_cmp_gt_f32_e64 s[{m+4}:{m+5}], {x3}, {t3}",
         f"s_store_dwordx2 s[{pm+8}:{pm+9}], s[16:17], {hex(off+32)}",
         f"v_add_f32 {t4}, {GO}, {t2}", f"v_max_f32 {t0}, {t5}, {t1}",
         f"v_cmp_gt_f32_e64 s[{m+6}:{m+7}], {t1}, {t5}",
         f"v_add_u32 v{ADDR}, v{LDS}, v{BOFF+c}", f"v_max_f32 {y}, {t0}, {t4}",
         f"v_cmp_gt_f32_e64 s[{m+8}:{m+9}], {t4}, {t0}", f"v_xor_b32 v{LDS}, v{LDS}, v{ADDR}"]
    if c == 15:
        L.append("s_add_u32 s16, s16, 640")
        L.append("s_addc_u32 s17, s17, 0")
    return L


def kernel(name, cell):
    body = []
    for parity in (0, 1):
        for c in range(16):
            body += cell(c, parity)
    n_instr = len([x for x in body if x.startswith("v_")]) // 32
    asm = "\\n\\t\"\n        \"".join(body)
    clobbers = ", ".join(f'"v{i}"' for i in range(2, 120))
    init = "".join('"v_mov_b32 v%d, %%[seed]\\n\\t"' % i for i in range(40, 120))
    return n_instr, f'''
__global__ __launch_bounds__(256) void {name}(float* out, float seed, float ng, float gs, float go, float ge, int iters, char* scratch) {{
    float r;
    const unsigned long long sp = reinterpret_cast<unsigned long long>(scratch) +
        static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) / 64)) * (1280ull * 400ull);
    const unsigned sp_lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(sp)), sp_hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(sp >> 32));
    asm volatile(
        "s_mov_b32 s16, %[splo]\\n\\t s_mov_b32 s17, %[sphi]\\n\\t"
        "s_mov_b32 s4, %[ng]\\n\\t s_mov_b32 s5, %[gs]\\n\\t s_mov_b32 s6, %[go]\\n\\t s_mov_b32 s7, %[ge]\\n\\t"
        "v_mov_b32 v2, s5\\n\\t v_mov_b32 v3, s7\\n\\t v_mov_b32 v4, s6\\n\\t v_mov_b32 v5, s4\\n\\t v_mov_b32 v6, s4\\n\\t v_mov_b32 v7, s6\\n\\t"
        "v_mov_b32 v8, %[seed]\\n\\t v_mov_b32 v9, 0\\n\\t v_mov_b32 v11, 0\\n\\t v_mov_b32 v16, 0\\n\\t v_mov_b32 v29, %[seed]\\n\\t v_mov_b32 v31, 0\\n\\t"
        "v_mov_b32 v32, 0\\n\\t v_mov_b32 v33, 0\\n\\t v_mov_b32 v34, 0\\n\\t"
        {init}
        "s_mov_b32 s8, %[iters]\\n\\t"
        "1:\\n\\t"
        "{asm}\\n\\t"
        "s_sub_u32 s8, s8, 1\\n\\t s_cmp_lg_u32 s8, 0\\n\\t s_cbranch_scc1 1b\\n\\t"
        "s_dcache_wb\\n\\t s_waitcnt lgkmcnt(0)\\n\\t"
        "v_add_f32 %[r], v8, v29\\n\\t v_add_f32 %[r], %[r], v40\\n\\t v_add_f32 %[r], %[r], v41\\n\\t v_add_f32 %[r], %[r], v72\\n\\t"
        "v_xor_b32 %[r], %[r], v32\\n\\t v_xor_b32 %[r], %[r], v33\\n\\t v_xor_b32 %[r], %[r], v34\\n\\t v_xor_b32 %[r], %[r], v31"
        : [r] "=&v"(r)
        : [seed] "v"(seed + threadIdx.x), [ng] "s"(ng), [gs] "s"(gs), [go] "s"(go), [ge] "s"(ge), [iters] "s"(iters), [splo] "s"(sp_lo), [sphi] "s"(sp_hi)
        : "s4", "s5", "s6", "s7", "s8", "s16", "s17", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "scc", "memory", {clobbers});
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}}
'''


nb, kb = kernel("cell_base", cell_base)
npk, kp = kernel("cell_pk", cell_pk)
ncm, kc = kernel("cell_cmp", cell_cmp)
nm3, km3 = kernel("cell_max3", cell_max3)
ncl, kcl = kernel("cell_clustered", cell_clustered)
ncl2, kcl2 = kernel("cell_clustered2", cell_clustered2)
ncl4, kcl4 = kernel("cell_clustered4", cell_clustered4)
src = f'''// GENERATED by gen_cell_pk.py -- do not edit.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do {{ hipError_t e = (x); if (e != hipSuccess) {{ printf("%s: %s\\n", #x, hipGetErrorString(e)); return 1; }} }} while (0)
{kb}
{kp}
{kc}
{km3}
{kcl}
{kcl2}
{kcl4}
template <typename K> int run(const char* name, K kern, int n_instr) {{
    float* d_out; CHECK(hipMalloc(&d_out, sizeof(float) * 256 * 256 * 8));
    char* d_scratch; CHECK(hipMalloc(&d_scratch, 256ull * 4 * 4 * 1280 * 400));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 200;  // x2 steps x16 cells
    for (int wps : {{1, 2, 3, 4}}) {{
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {{
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(256 * wps), dim3(256), 0, 0, d_out, 1.0f, -0.001f, -1.79f, -6.9f, -0.18f, iters, d_scratch);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }}
        const double cells = 32.0 * iters * wps;  // per SIMD
        printf("%-10s %2d instr/cell  waves/SIMD %d: %.3f ms -> %.2f ns per cell per SIMD, %.2f ns/instr\\n", name, n_instr, wps, best,
               best * 1e6 / cells, best * 1e6 / cells / n_instr);
    }}
    CHECK(hipFree(d_out)); CHECK(hipFree(d_scratch));
    return 0;
}}
int main() {{
    if (run("base", cell_base, {nb})) return 1;
    if (run("pk", cell_pk, {npk})) return 1;
    if (run("cmp+sst", cell_cmp, {ncm})) return 1;
    if (run("max3", cell_max3, {nm3})) return 1;
    if (run("clustered", cell_clustered, {ncl})) return 1;
    if (run("clustered2", cell_clustered2, {ncl2})) return 1;
    if (run("clustered4", cell_clustered4, {ncl4})) return 1;
    return 0;
}}
'''
Path(__file__).with_name("cell_pk.hip").write_text(src)
print("base", nb, "pk", npk)
