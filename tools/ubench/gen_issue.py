#!/usr/bin/env python3
"""Generate tools/ubench/issue_model.hip (round 3): what does the gfx950 vector unit charge for the
instructions of the lean DP cell, counted in SHADER CYCLES (s_memtime around the loop, per wavefront)
with the clock the chip actually held (s_memtime / s_memrealtime), not in wall time at an assumed 2.4 GHz.

Questions:
  * v_max_f32 / v_max3_f32 alone cost ~4 cycles, v_add_f32 ~2.  In a 3:1 add:max mix the max is hidden.
    Does v_max3_f32 hide the same way?  How many fast instructions does a slow one need after it?
  * integer maxima (v_max_i32, v_min_u32, v_max3_i32, v_min3_u32): same price?
  * SGPR vs VGPR constant operand; v_add_f32 with a DPP source (folds the hand-off mov into the first add).
  * the 15-instruction cell as the kernel has it, and re-codings: max3 -> two v_max_f32 (17 instructions),
    different spacings.
Every kernel: one asm block, hand-allocated registers, `iters` loop iterations of the listed body.
"""
from pathlib import Path
import sys

NG, GS, GO, GE = "s4", "s5", "s6", "s7"
ZL, DIAG, ADDR, LDS = 8, 29, 30, 31
YP, XB, S, BOFF = 40, 72, 88, 104


def _regs(c, parity):
    xin = f"v{YP+2*c+1}" if parity == 0 else f"v{XB+c}"
    xout = f"v{XB+c}" if parity == 0 else f"v{YP+2*c+1}"
    return xin, xout, f"v{YP+2*c}"


def cells(fn):
    body = []
    for parity in (0, 1):
        for c in range(16):
            body += fn(c, parity)
    return body


def cell_lean_kernel(c, parity):
    """The order viterbi_ck.hip has (COATI_CELL_LEAN), five temporaries."""
    xin, xout, y = _regs(c, parity)
    diag = f"v{DIAG}" if c == 0 else _regs(c - 1, parity)[0]
    t0, t1, t2, t3, t4, zl, s = "v10", "v11", "v12", "v13", "v14", f"v{ZL}", f"v{S+c}"
    return [f"v_add_f32 {t0}, {diag}, {s}", f"v_add_f32 {t1}, {GE}, {zl}", f"v_add_f32 {t2}, {GS}, {zl}",
            f"v_add_f32 {t3}, {GO}, {t0}", f"v_add_f32 {t0}, {NG}, {t0}", f"v_add_u32 {s}, v{LDS}, v7",
            f"v_add_f32 {t4}, {NG}, {t2}", f"v_max_f32 {zl}, {t3}, {t1}", f"v_add_f32 {t1}, {GS}, {y}",
            f"v_add_f32 {t3}, {NG}, {t0}", f"v_add_f32 {t2}, {GO}, {t2}", f"v_max3_f32 {xout}, {t3}, {t1}, {t4}",
            f"v_add_f32 {t1}, {GE}, {y}", f"v_add_f32 {t0}, {GO}, {t0}", f"v_max3_f32 {y}, {t0}, {t1}, {t2}"]


def cell_17(c, parity, order=0, vconst=False, imax=False):
    """max3 -> two v_max_f32: 12 fast + 5 slow, a slow one never directly after a slow one.
    order 0: F F M F F M ... ; order 1: the five maxima as late as their inputs allow, single fast between."""
    xin, xout, y = _regs(c, parity)
    diag = f"v{DIAG}" if c == 0 else _regs(c - 1, parity)[0]
    t0, t1, t2, t3, t4, t5, t6, zl, s = "v10", "v11", "v12", "v13", "v14", "v15", "v17", f"v{ZL}", f"v{S+c}"
    ng, gs, go, ge = (("v2", "v3", "v4", "v5") if vconst else (NG, GS, GO, GE))
    mx = "v_min_u32" if imax else "v_max_f32"
    if order == 0:
        return [f"v_add_f32 {t0}, {diag}, {s}",          # M
                f"v_add_f32 {t1}, {ge}, {zl}",            # z2
                f"v_add_f32 {t2}, {gs}, {zl}",            # i1
                f"v_add_f32 {t3}, {go}, {t0}",            # z1
                f"v_add_f32 {t0}, {ng}, {t0}",            # m1
                f"{mx} {zl}, {t3}, {t1}",                 # Z
                f"v_add_f32 {t1}, {gs}, {y}",             # x2
                f"v_add_f32 {t3}, {ng}, {t0}",            # x1
                f"v_add_f32 {t4}, {ng}, {t2}",            # x3
                f"{mx} {t5}, {t3}, {t1}",                 # max(x1,x2)
                f"v_add_f32 {t1}, {ge}, {y}",             # y2
                f"v_add_f32 {t3}, {go}, {t0}",            # y1
                f"{mx} {xout}, {t5}, {t4}",               # X
                f"v_add_f32 {t2}, {go}, {t2}",            # y3
                f"{mx} {t6}, {t3}, {t1}",                 # max(y1,y2)
                f"v_add_u32 {s}, v{LDS}, v7",
                f"{mx} {y}, {t6}, {t2}"]                  # Y
    # order 1: every maximum followed by exactly two fast ops where possible
    return [f"v_add_f32 {t0}, {diag}, {s}", f"v_add_f32 {t1}, {ge}, {zl}", f"v_add_f32 {t3}, {go}, {t0}",
            f"v_add_f32 {t2}, {gs}, {zl}",
            f"{mx} {zl}, {t3}, {t1}",
            f"v_add_f32 {t0}, {ng}, {t0}", f"v_add_f32 {t1}, {gs}, {y}", f"v_add_f32 {t3}, {ng}, {t0}",
            f"{mx} {t5}, {t3}, {t1}",
            f"v_add_f32 {t4}, {ng}, {t2}", f"v_add_f32 {t1}, {ge}, {y}",
            f"{mx} {xout}, {t5}, {t4}",
            f"v_add_f32 {t3}, {go}, {t0}", f"v_add_f32 {t2}, {go}, {t2}",
            f"{mx} {t6}, {t3}, {t1}",
            f"v_add_u32 {s}, v{LDS}, v7",
            f"{mx} {y}, {t6}, {t2}"]


def cell_15v(c, parity):
    """the kernel's 15 instructions with the four constants in VGPRs instead of SGPRs"""
    L = cell_lean_kernel(c, parity)
    for sreg, vreg in ((NG, "v2"), (GS, "v3"), (GO, "v4"), (GE, "v5")):
        L = [x.replace(f" {sreg},", f" {vreg},") for x in L]
    return L


def cell_15i(c, parity):
    """15 instructions with integer maxima on the bit patterns (v_min_u32 / v_min3_u32)"""
    return [x.replace("v_max3_f32", "v_min3_u32").replace("v_max_f32", "v_min_u32") for x in cell_lean_kernel(c, parity)]


def cell_15_wide(c, parity):
    """max3 kept, but three or more fast ops after every slow one (needs 8 temporaries: the slow ops of this
    cell issue among the adds of the NEXT cell's head -- software pipelined by a third of a cell)."""
    xin, xout, y = _regs(c, parity)
    diag = f"v{DIAG}" if c == 0 else _regs(c - 1, parity)[0]
    t0, t1, t2, t3, t4, t5, t6, t7, zl, s = "v10", "v11", "v12", "v13", "v14", "v15", "v17", "v18", f"v{ZL}", f"v{S+c}"
    return [f"v_add_f32 {t0}, {diag}, {s}", f"v_add_f32 {t1}, {GE}, {zl}", f"v_add_f32 {t2}, {GS}, {zl}",
            f"v_add_f32 {t3}, {GO}, {t0}",
            f"v_max_f32 {zl}, {t3}, {t1}",
            f"v_add_f32 {t0}, {NG}, {t0}", f"v_add_f32 {t4}, {NG}, {t2}", f"v_add_f32 {t5}, {GS}, {y}",
            f"v_add_f32 {t6}, {NG}, {t0}",
            f"v_max3_f32 {xout}, {t6}, {t5}, {t4}",
            f"v_add_f32 {t7}, {GE}, {y}", f"v_add_f32 {t2}, {GO}, {t2}", f"v_add_f32 {t0}, {GO}, {t0}",
            f"v_add_u32 {s}, v{LDS}, v7",
            f"v_max3_f32 {y}, {t0}, {t7}, {t2}"]


def cell19_parts(c, parity):
    """viterbi_lp's cell (gen_viterbi_lp.py): packed adds; returns (front, back): the recurrence and the decision bits"""
    xin, xout, y = _regs(c, parity)
    P = f"v[{YP + 2 * c}:{YP + 2 * c + 1}]"   # [Y:Xin] here (register roles do not matter for timing)
    front = ["v_pk_add_f32 v[10:11], v[28:29], v[2:3] op_sel:[0,0] op_sel_hi:[0,1]",
             "v_pk_add_f32 v[12:13], v[8:9], v[4:5] op_sel:[0,0] op_sel_hi:[0,1]",
             "v_pk_add_f32 v[14:15], v[10:11], v[2:3] op_sel:[1,1] op_sel_hi:[1,0]",
             f"v_pk_add_f32 v[16:17], {P}, v[4:5] op_sel:[1,1] op_sel_hi:[1,0]",
             "v_pk_add_f32 v[18:19], v[12:13], v[2:3] op_sel:[1,1] op_sel_hi:[1,0]",
             f"v_add_f32 v28, v{YP + 2 * c}, v{S + c}",
             "v_max_f32 v8, v10, v12",
             "v_sub_f32 v20, v12, v10",
             f"v_max3_f32 v{YP + 2 * c}, v14, v16, v18",
             f"v_max3_f32 v{YP + 2 * c + 1}, v15, v17, v19",
             f"v_add_u32 v{S + c}, v{LDS}, v7"]
    back = ["v_alignbit_b32 v24, v24, v20, 31",
            f"v_pk_add_f32 v[14:15], v[14:15], {P} neg_lo:[0,1] neg_hi:[0,1]",
            f"v_pk_add_f32 v[16:17], v[16:17], {P} neg_lo:[0,1] neg_hi:[0,1]",
            "v_alignbit_b32 v25, v25, v14, 31", "v_alignbit_b32 v25, v25, v16, 31",
            "v_alignbit_b32 v26, v26, v15, 31", "v_alignbit_b32 v26, v26, v17, 31"]
    return front, back


def cell19(c, parity):
    f, b = cell19_parts(c, parity)
    return f + b


def cell19_interleaved(c, parity):
    """the decision-bit instructions between the adds (a different temporary set would be needed in the kernel; here
    only the issue pattern matters)"""
    f, b = cell19_parts(c, parity)
    out = []
    for i in range(max(len(f), len(b))):
        if i < len(f):
            out.append(f[i])
        if i < len(b):
            out.append(b[i])
    return out


def cell19_fastsplit(c, parity):
    """the three fast-class instructions spread evenly among the slow ones"""
    f, b = cell19_parts(c, parity)
    seq = f + b
    fast = [x for x in seq if x.split()[0] in ("v_add_f32", "v_sub_f32", "v_add_u32")]
    slow = [x for x in seq if x not in fast]
    out, k = [], 0
    for i, x in enumerate(slow):
        out.append(x)
        if (i + 1) % 5 == 0 and k < len(fast):
            out.append(fast[k])
            k += 1
    return out + fast[k:]


def mix(fast_n, slow, slow_n=1, fast="v_add_f32", const="v2"):
    """4 independent chains (v10..v13); per chain: fast_n fast ops then slow_n slow ops, chains interleaved
    instruction by instruction so that consecutive instructions are independent"""
    body = []
    seq = [fast] * fast_n + [slow] * slow_n
    for rep in range(8):
        for op in seq:
            for r in ("v10", "v11", "v12", "v13"):
                body.append(fmt(op, r, const))
    return body


def mix_serial(fast_n, slow, const="v2"):
    """one chain per group: fast_n fast ops then the slow one, each instruction depending on the one before"""
    body = []
    for rep in range(8):
        for r in ("v10", "v11", "v12", "v13"):
            for op in ["v_add_f32"] * fast_n + [slow]:
                body.append(fmt(op, r, const))
    return body


def fmt(op, r, const):
    if op in ("v_max3_f32", "v_max3_i32", "v_min3_u32", "v_fma_f32", "v_med3_f32"):
        return f"{op} {r}, {const}, {r}, v14"
    if op == "v_add_f32_dpp":
        return f"v_add_f32_dpp {r}, {r}, v14 wave_shr:1 row_mask:0xf bank_mask:0xf"
    if op == "v_add_f32_dpp_row":
        return f"v_add_f32_dpp {r}, {r}, v14 row_shr:1 row_mask:0xf bank_mask:0xf"
    if op == "v_mov_b32_dpp":
        return f"v_mov_b32_dpp {r}, v14 wave_shr:1 row_mask:0xf bank_mask:0xf"
    if op == "v_pk_add_f32":
        return f"v_pk_add_f32 v[{20 + 2 * int(r[1:]) - 20}:{21 + 2 * int(r[1:]) - 20}], v[{20 + 2 * int(r[1:]) - 20}:{21 + 2 * int(r[1:]) - 20}], v[2:3]"
    if op == "v_pk_add_f32_sel":  # (x, x) + (c0, c1), then sign differences with neg: the modifiers the DP cell would use
        i = 2 * int(r[1:])
        return f"v_pk_add_f32 v[{i}:{i + 1}], v[{i}:{i + 1}], v[2:3] op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]"
    if op in ("v_pk_fma_f32", "v_pk_mul_f32"):
        i = 2 * int(r[1:])
        return f"{op} v[{i}:{i + 1}], v[{i}:{i + 1}], v[2:3]" + (f", v[{i + 8}:{i + 9}]" if op == "v_pk_fma_f32" else "")
    if op == "v_pk_mov_b32":
        i = 2 * int(r[1:])
        return f"v_pk_mov_b32 v[{i}:{i + 1}], v[{i + 8}:{i + 9}], v[{i + 8}:{i + 9}] op_sel:[0,1]"
    if op == "v_pk_max_f32":
        return f"v_pk_max_f32 v[{2 * int(r[1:])}:{2 * int(r[1:]) + 1}], v[{2 * int(r[1:])}:{2 * int(r[1:]) + 1}], v[2:3]"
    return f"{op} {r}, {const}, {r}"


VARIANTS = {}


def V(name, body):
    VARIANTS[name] = body


for op in ("v_add_f32", "v_max_f32", "v_max3_f32", "v_min_u32", "v_max_i32", "v_min3_u32", "v_max3_i32", "v_fma_f32",
           "v_mul_f32", "v_add_f32_dpp", "v_add_f32_dpp_row", "v_mov_b32_dpp", "v_med3_f32"):
    V("pure " + op, mix(1, op, 0, fast=op))
V("fma x1 : add x1", mix(1, "v_fma_f32"))
V("pk_fma x1 : add x2", mix(2, "v_pk_fma_f32"))
for op in ("v_pk_add_f32", "v_pk_add_f32_sel", "v_pk_mov_b32", "v_pk_fma_f32", "v_pk_mul_f32"):
    V("pure " + op, mix(1, op, 0, fast=op))
V("add x3 : pk_add", mix(3, "v_pk_add_f32"))
V("pk_add x3 : max3", mix(3, "v_max3_f32", fast="v_pk_add_f32"))
V("pure v_add_f32 sgpr", mix(1, "v_add_f32", 0, const="s4"))
V("pure v_max_f32 sgpr", mix(1, "v_max_f32", 0, fast="v_max_f32", const="s4"))
for n in (1, 2, 3, 4):
    V(f"add x{n} : max", mix(n, "v_max_f32"))
    V(f"add x{n} : max3", mix(n, "v_max3_f32"))
V("add x1 : min_u32", mix(1, "v_min_u32"))
V("add x2 : min3_u32", mix(2, "v_min3_u32"))
V("add x3 : min3_u32", mix(3, "v_min3_u32"))
V("add x2 : max x2", mix(2, "v_max_f32", 2))
V("add x4 : max x2", mix(4, "v_max_f32", 2))
V("add x3 : fma", mix(3, "v_fma_f32"))
V("add x3 : mov_dpp", mix(3, "v_mov_b32_dpp"))
V("serial add x3 : max", mix_serial(3, "v_max_f32"))
V("serial add x3 : max3", mix_serial(3, "v_max3_f32"))
V("cell15 kernel", cells(cell_lean_kernel))
V("cell15 vgpr consts", cells(cell_15v))
V("cell15 int maxima", cells(cell_15i))
V("cell15 wide", cells(cell_15_wide))
V("cell19 lp", cells(cell19))
V("cell19 lp interleaved", cells(cell19_interleaved))
V("cell19 lp fast spread", cells(cell19_fastsplit))
V("cell17 order0", cells(lambda c, p: cell_17(c, p, 0)))
V("cell17 order1", cells(lambda c, p: cell_17(c, p, 1)))
V("cell17 order1 vgpr", cells(lambda c, p: cell_17(c, p, 1, vconst=True)))
V("cell17 order1 int", cells(lambda c, p: cell_17(c, p, 1, imax=True)))


def bank_mix(op, regs, const="v2", dsts=None):
    body = []
    for rep in range(8):
        for i, r in enumerate(regs):
            d = dsts[i] if dsts else r
            if op == "v_max3_f32":
                body.append(f"v_max3_f32 {d}, {const}, {r}, {r.replace('v1', 'v5') if False else r}")
            else:
                body.append(f"{op} {d}, {const}, {r}")
    return body


# v2 is bank 2 (register number mod 4)
V("bank add: src same bank", bank_mix("v_add_f32", ["v10", "v14", "v18", "v22"]))
V("bank add: src other bank", bank_mix("v_add_f32", ["v11", "v15", "v19", "v23"]))
V("bank add: dst=const bank", bank_mix("v_add_f32", ["v11", "v15", "v19", "v23"], dsts=["v26", "v30", "v34", "v38"]))
V("bank add: dst 3rd bank", bank_mix("v_add_f32", ["v11", "v15", "v19", "v23"], dsts=["v24", "v28", "v32", "v36"]))
V("bank max: src same bank", bank_mix("v_max_f32", ["v10", "v14", "v18", "v22"]))
V("bank max: src other bank", bank_mix("v_max_f32", ["v11", "v15", "v19", "v23"]))
V("bank max3: all same", ["v_max3_f32 v10, v2, v14, v18", "v_max3_f32 v22, v6, v26, v30"] * 16)
V("bank max3: all differ", ["v_max3_f32 v10, v2, v11, v12", "v_max3_f32 v22, v6, v27, v28"] * 16)
V("bank max3: two same", ["v_max3_f32 v10, v2, v14, v11", "v_max3_f32 v22, v6, v26, v27"] * 16)
V("add x3 other bank : max3 differ", (["v_add_f32 v11, v2, v11", "v_add_f32 v15, v2, v15", "v_add_f32 v19, v2, v19", "v_max3_f32 v20, v2, v11, v16"]) * 8)


def kernel(idx, body):
    n_valu = len([x for x in body if x.startswith("v_")])
    asm = "\\n\\t\"\n        \"".join(body)
    clobbers = ", ".join(f'"v{i}"' for i in range(2, 104))
    init = "".join('"v_mov_b32 v%d, %%[seed]\\n\\t"' % i for i in range(8, 104))
    return n_valu, f'''
__global__ __launch_bounds__(256) void k{idx}(unsigned long long* stamps, float* out, float seed, float ng, float gs, float go, float ge, int iters) {{
    float r;
    unsigned t0l, t0h, t1l, t1h, r0l, r0h, r1l, r1h;
    asm volatile(
        "s_mov_b32 s4, %[ng]\\n\\t s_mov_b32 s5, %[gs]\\n\\t s_mov_b32 s6, %[go]\\n\\t s_mov_b32 s7, %[ge]\\n\\t"
        "v_mov_b32 v2, s4\\n\\t v_mov_b32 v3, s5\\n\\t v_mov_b32 v4, s6\\n\\t v_mov_b32 v5, s7\\n\\t v_mov_b32 v6, s4\\n\\t v_mov_b32 v7, s6\\n\\t"
        {init}
        "v_mov_b32 v31, 0\\n\\t"
        "s_mov_b32 s8, %[iters]\\n\\t"
        "s_memtime s[20:21]\\n\\t s_memrealtime s[24:25]\\n\\t s_waitcnt lgkmcnt(0)\\n\\t"
        "1:\\n\\t"
        "{asm}\\n\\t"
        "s_sub_u32 s8, s8, 1\\n\\t s_cmp_lg_u32 s8, 0\\n\\t s_cbranch_scc1 1b\\n\\t"
        "s_memtime s[22:23]\\n\\t s_memrealtime s[26:27]\\n\\t s_waitcnt lgkmcnt(0)\\n\\t"
        "v_mov_b32 %[t0l], s20\\n\\t v_mov_b32 %[t0h], s21\\n\\t v_mov_b32 %[t1l], s22\\n\\t v_mov_b32 %[t1h], s23\\n\\t"
        "v_mov_b32 %[r0l], s24\\n\\t v_mov_b32 %[r0h], s25\\n\\t v_mov_b32 %[r1l], s26\\n\\t v_mov_b32 %[r1h], s27\\n\\t"
        "v_add_f32 %[r], v8, v29\\n\\t v_add_f32 %[r], %[r], v40\\n\\t v_add_f32 %[r], %[r], v41\\n\\t v_add_f32 %[r], %[r], v72\\n\\t"
        "v_add_f32 %[r], %[r], v10\\n\\t v_add_f32 %[r], %[r], v11\\n\\t v_add_f32 %[r], %[r], v12\\n\\t v_add_f32 %[r], %[r], v13"
        : [r] "=&v"(r), [t0l] "=&v"(t0l), [t0h] "=&v"(t0h), [t1l] "=&v"(t1l), [t1h] "=&v"(t1h), [r0l] "=&v"(r0l), [r0h] "=&v"(r0h), [r1l] "=&v"(r1l), [r1h] "=&v"(r1h)
        : [seed] "v"(seed + threadIdx.x), [ng] "s"(ng), [gs] "s"(gs), [go] "s"(go), [ge] "s"(ge), [iters] "s"(iters)
        : "s4", "s5", "s6", "s7", "s8", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc", "vcc", "memory", {clobbers});
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if((threadIdx.x & 63) == 0) {{
        const unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
        stamps[2 * w] = ((static_cast<unsigned long long>(t1h) << 32) | t1l) - ((static_cast<unsigned long long>(t0h) << 32) | t0l);
        stamps[2 * w + 1] = ((static_cast<unsigned long long>(r1h) << 32) | r1l) - ((static_cast<unsigned long long>(r0h) << 32) | r0l);
    }}
}}
'''


names = list(VARIANTS)
if len(sys.argv) > 1:
    names = [n for n in names if any(a in n for a in sys.argv[1:])]
kernels, runs = [], []
for i, n in enumerate(names):
    cnt, src_k = kernel(i, VARIANTS[n])
    kernels.append(src_k)
    runs.append(f'    if (run("{n}", k{i}, {cnt})) return 1;')
src = f'''// GENERATED by gen_issue.py -- do not edit, do not commit.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHECK(x) do {{ hipError_t e = (x); if (e != hipSuccess) {{ printf("%s: %s\\n", #x, hipGetErrorString(e)); return 1; }} }} while (0)
{"".join(kernels)}
template <typename K> int run(const char* name, K kern, int n_valu) {{
    static float* d_out = nullptr; static unsigned long long* d_st = nullptr;
    if(!d_out) {{ CHECK(hipMalloc(&d_out, sizeof(float) * 256 * 256 * 8)); CHECK(hipMalloc(&d_st, 16 * 1024 * 8)); }}
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-26s %3d VALU/iter |", name, n_valu);
    for (int wps : {{1, 2, 4}}) {{
        // ~1.5 M vector instructions per wavefront per launch; warm the clock with 40 launches of the same kernel
        const int iters = std::max(1, 1500000 / n_valu);
        for (int rep = 0; rep < 40; ++rep)
            hipLaunchKernelGGL(kern, dim3(256 * wps), dim3(256), 0, 0, d_st, d_out, 1.0f, -0.001f, -1.79f, -6.9f, -0.18f, iters);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(256 * wps), dim3(256), 0, 0, d_st, d_out, 1.0f, -0.001f, -1.79f, -6.9f, -0.18f, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> st(2 * 1024 * wps);
        CHECK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> cyc, ghz;
        for (int w = 0; w < 1024 * wps; ++w) {{ cyc.push_back(double(st[2 * w])); ghz.push_back(double(st[2 * w]) / double(st[2 * w + 1]) * 0.1); }}
        std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
        const double instr = double(n_valu) * iters;
        printf("  w%d: %5.2f cyc/instr/SIMD, %.2f GHz, wall %.2f @2.4 |", wps, cyc[cyc.size() / 2] / instr / wps, ghz[ghz.size() / 2],
               ms * 1e-3 * 2.4e9 / (instr * wps));
    }}
    printf("\\n");
    return 0;
}}
int main() {{
{chr(10).join(runs)}
    return 0;
}}
'''
Path(__file__).with_name("issue_model.hip").write_text(src)
print("generated", len(names), "kernels")
