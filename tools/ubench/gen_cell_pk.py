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


def _lean_regs(c, parity):
    xin = f"v{YP+2*c+1}" if parity == 0 else f"v{XB+c}"
    xout = f"v{XB+c}" if parity == 0 else f"v{YP+2*c+1}"
    return xin, xout, f"v{YP+2*c}"


def cell_lean(c, parity):
    """15 VALU: the fill without decision bits (checkpoint design), slow ops clustered at the end.
    The diagonal input is read from the ping-pong register directly (no v_mov)."""
    xin, xout, y = _lean_regs(c, parity)
    diag = f"v{DIAG}" if c == 0 else _lean_regs(c - 1, parity)[0]
    t0, t1, t2, t3, t4, t5, t6, zl = "v10", "v11", "v12", "v13", "v14", "v15", "v17", f"v{ZL}"
    t7, t8, t9 = "v18", "v19", "v20"
    return [f"v_add_f32 {t0}, {diag}, v{S+c}", f"v_add_f32 {t1}, {GE}, {zl}", f"v_add_f32 {t2}, {GS}, {zl}",
            f"v_add_f32 {t3}, {GO}, {t0}", f"v_add_f32 {t0}, {NG}, {t0}", f"v_add_f32 {t5}, {GS}, {y}",
            f"v_add_f32 {t8}, {GE}, {y}", f"v_add_f32 {t4}, {NG}, {t0}", f"v_add_f32 {t6}, {NG}, {t2}",
            f"v_add_f32 {t7}, {GO}, {t0}", f"v_add_f32 {t9}, {GO}, {t2}", f"v_add_u32 v{ADDR}, v{LDS}, v{BOFF+c}",
            f"v_max_f32 {zl}, {t3}, {t1}", f"v_max3_f32 {xout}, {t4}, {t5}, {t6}", f"v_max3_f32 {y}, {t7}, {t8}, {t9}"]


def cell_lean_bc(c, parity):
    """cell_lean with the sources of each v_max3 / v_max in ONE VGPR bank (register number mod 4): does a
    bank clash cost issue cycles on this chip?  (The compiler allocates the kernel's temporaries; ~1.4 of the
    multi-source instructions per cell clash there.)"""
    xin, xout, y = _lean_regs(c, parity)
    diag = f"v{DIAG}" if c == 0 else _lean_regs(c - 1, parity)[0]
    t0, t2, zl = "v10", "v12", f"v{ZL}"
    t3, t1 = "v13", "v25"             # bank 1, 1
    t4, t5, t6 = "v14", "v22", "v26"  # bank 2, 2, 2
    t7, t8, t9 = "v19", "v23", "v27"  # bank 3, 3, 3
    return [f"v_add_f32 {t0}, {diag}, v{S+c}", f"v_add_f32 {t1}, {GE}, {zl}", f"v_add_f32 {t2}, {GS}, {zl}",
            f"v_add_f32 {t3}, {GO}, {t0}", f"v_add_f32 {t0}, {NG}, {t0}", f"v_add_f32 {t5}, {GS}, {y}",
            f"v_add_f32 {t8}, {GE}, {y}", f"v_add_f32 {t4}, {NG}, {t0}", f"v_add_f32 {t6}, {NG}, {t2}",
            f"v_add_f32 {t7}, {GO}, {t0}", f"v_add_f32 {t9}, {GO}, {t2}", f"v_add_u32 v{ADDR}, v{LDS}, v{BOFF+c}",
            f"v_max_f32 {zl}, {t3}, {t1}", f"v_max3_f32 {xout}, {t4}, {t5}, {t6}", f"v_max3_f32 {y}, {t7}, {t8}, {t9}"]


def cell_lean31(c, parity):
    """The same 15 instructions with one slow op after every 3+ fast ones."""
    xin, xout, y = _lean_regs(c, parity)
    diag = f"v{DIAG}" if c == 0 else _lean_regs(c - 1, parity)[0]
    t0, t1, t2, t3, t4, t5, t6, zl = "v10", "v11", "v12", "v13", "v14", "v15", "v17", f"v{ZL}"
    t7, t8, t9 = "v18", "v19", "v20"
    return [f"v_add_f32 {t0}, {diag}, v{S+c}", f"v_add_f32 {t1}, {GE}, {zl}", f"v_add_f32 {t2}, {GS}, {zl}",
            f"v_add_f32 {t3}, {GO}, {t0}", f"v_add_f32 {t0}, {NG}, {t0}", f"v_add_f32 {t5}, {GS}, {y}",
            f"v_max_f32 {zl}, {t3}, {t1}",
            f"v_add_f32 {t8}, {GE}, {y}", f"v_add_f32 {t4}, {NG}, {t0}", f"v_add_f32 {t6}, {NG}, {t2}",
            f"v_add_f32 {t7}, {GO}, {t0}",
            f"v_max3_f32 {xout}, {t4}, {t5}, {t6}",
            f"v_add_f32 {t9}, {GO}, {t2}", f"v_add_u32 v{ADDR}, v{LDS}, v{BOFF+c}",
            f"v_max3_f32 {y}, {t7}, {t8}, {t9}"]


def cell_lean2(c, parity):
    """Two columns interleaved is not possible (zl chains); instead: slow ops of the PREVIOUS cell are
    delayed into this cell's adds (software pipelining by one cell), 3:1."""
    xin, xout, y = _lean_regs(c, parity)
    diag = f"v{DIAG}" if c == 0 else _lean_regs(c - 1, parity)[0]
    # two temp sets alternate by column so that the delayed max3 of cell c-1 can still read its inputs
    sets = ([10, 11, 12, 13, 14, 15, 17, 18, 19, 20], [22, 23, 24, 25, 26, 27, 35, 36, 37, 38])
    t0, t1, t2, t3, t4, t5, t6, t7, t8, t9 = (f"v{r}" for r in sets[c & 1])
    p4, p5, p6, p7, p8, p9 = (f"v{r}" for r in sets[(c + 1) & 1][4:])
    zl = f"v{ZL}"
    pxout, py = (_lean_regs(c - 1, parity)[1], _lean_regs(c - 1, parity)[2]) if c > 0 else (_lean_regs(15, 1 - parity)[1], _lean_regs(15, 1 - parity)[2])
    return [f"v_add_f32 {t0}, {diag}, v{S+c}", f"v_add_f32 {t1}, {GE}, {zl}", f"v_add_f32 {t2}, {GS}, {zl}",
            f"v_max3_f32 {pxout}, {p4}, {p5}, {p6}",
            f"v_add_f32 {t3}, {GO}, {t0}", f"v_add_f32 {t0}, {NG}, {t0}", f"v_add_f32 {t5}, {GS}, {y}",
            f"v_max3_f32 {py}, {p7}, {p8}, {p9}",
            f"v_add_f32 {t8}, {GE}, {y}", f"v_add_f32 {t4}, {NG}, {t0}", f"v_add_f32 {t6}, {NG}, {t2}",
            f"v_max_f32 {zl}, {t3}, {t1}",
            f"v_add_f32 {t7}, {GO}, {t0}", f"v_add_f32 {t9}, {GO}, {t2}", f"v_add_u32 v{ADDR}, v{LDS}, v{BOFF+c}"]


def _step_overhead():
    """What a wavefront step adds to its 16 cells in the checkpoint design: the hand-off of X/Z/row code
    (readlane + mov + DPP shift each), one 8-byte store per lane of the received (diag, zl), half a
    16-byte row-checkpoint store (8 of them per 16 steps), the scalar bookkeeping."""
    L = []
    for src, dst in ((9, 29), (8, 8), (39, 39)):
        L += [f"v_readlane_b32 s20, v{src}, 5", "v_mov_b32 v21, s20", f"v_mov_b32_dpp v21, v{src} wave_shr:1 row_mask:0xf bank_mask:0xf",
              f"v_mov_b32 v{dst}, v21"]
    L += ["global_store_dwordx2 v3, v[8:9], s[16:17]", "s_add_u32 s16, s16, 1024", "s_addc_u32 s17, s17, 0"]
    return L


def cell_lean_st(c, parity):
    L = cell_lean(c, parity)
    if c == 15:
        L += _step_overhead()
        L.append(f"global_store_dwordx4 v4, v[{40 + 8 * parity}:{43 + 8 * parity}], s[16:17] offset:512")
    return L


def cell_lean31_st(c, parity):
    L = cell_lean31(c, parity)
    if c == 15:
        L += _step_overhead()
        L.append(f"global_store_dwordx4 v4, v[{40 + 8 * parity}:{43 + 8 * parity}], s[16:17] offset:512")
    return L


def cell_clustered_st(c, parity):
    """The current 25-instruction cell with the current per-step overhead (2.5 row stores)."""
    L = cell_clustered(c, parity)
    if c == 15:
        L += _step_overhead()[:-3]
        L += ["global_store_dword v2, v32, s[16:17]", "global_store_dword v2, v33, s[16:17] offset:256"]
        if parity:
            L.append("global_store_dword v2, v34, s[16:17] offset:512")
        L += ["s_add_u32 s16, s16, 768", "s_addc_u32 s17, s17, 0"]
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
        "v_mbcnt_lo_u32_b32 v2, -1, 0\\n\\t v_mbcnt_hi_u32_b32 v2, -1, v2\\n\\t v_lshlrev_b32 v3, 3, v2\\n\\t v_lshlrev_b32 v4, 4, v2\\n\\t v_lshlrev_b32 v2, 2, v2\\n\\t"
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


import sys

ALL = {"base": cell_base, "pk": cell_pk, "cmp+sst": cell_cmp, "max3": cell_max3, "clustered": cell_clustered,
       "clustered2": cell_clustered2, "clustered4": cell_clustered4, "clustered_st": cell_clustered_st,
       "lean": cell_lean, "lean_bc": cell_lean_bc, "lean31": cell_lean31, "lean2": cell_lean2, "lean_st": cell_lean_st,
       "lean31_st": cell_lean31_st}
# default: the round-2 question (how fast can the fill go without decision bits); `all` adds the round-1 codings
# (pk needs v2..v7 as constants: run it alone, `gen_cell_pk.py pk`, the store variants overwrite them)
names = sys.argv[1:] or ["clustered", "clustered_st", "lean", "lean31", "lean2", "lean_st", "lean31_st"]
if names == ["all"]:
    names = [n for n in ALL if n != "pk"]
kernels, runs = [], []
for n in names:
    fn = "cell_" + n.replace("+", "_")
    cnt, src_k = kernel(fn, ALL[n])
    kernels.append(src_k)
    runs.append(f'    if (run("{n}", {fn}, {cnt})) return 1;')
src = f'''// GENERATED by gen_cell_pk.py -- do not edit, do not commit.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do {{ hipError_t e = (x); if (e != hipSuccess) {{ printf("%s: %s\\n", #x, hipGetErrorString(e)); return 1; }} }} while (0)
{"".join(kernels)}
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
        printf("%-12s %2d VALU/cell  waves/SIMD %d: %.3f ms -> %.2f ns per 64-lane cell per SIMD = %.0f GCUPS on 1024 SIMDs\\n", name, n_instr, wps, best,
               best * 1e6 / cells, 65536.0 / (best * 1e6 / cells));
    }}
    CHECK(hipFree(d_out)); CHECK(hipFree(d_scratch));
    return 0;
}}
int main() {{
{chr(10).join(runs)}
    return 0;
}}
'''
Path(__file__).with_name("cell_pk.hip").write_text(src)
print("generated", names)
