#!/usr/bin/env python3
"""Generate coati_amd/csrc/viterbi_lp_block.inc: the hand-allocated gfx950 instruction text of
viterbi_lp.hip's 16-step blocks (4 columns per lane), as two string-literal macros.

Why generated text and not C++: a wavefront that is ALONE on its SIMD (a single long pair cut into
strips: fewer strips than SIMDs) issues one instruction per ~4.3 cycles whatever the instruction
(profiles/r03/ubench_issue_model.txt), so the only thing that counts is the NUMBER of
instructions per step -- and there v_pk_add_f32 does two of the cell's additions for the price of
one (profiles/r03/ubench_issue_model.txt, "pure v_pk_add_f32": 5.1 cycles alone, like v_add_f32).
The packed form needs its operands in even-aligned register PAIRS and the maxima read the halves
of those pairs: inline-asm operands cannot name half of a 64-bit operand, so the block names its
registers itself (COATI_LP_* pins them in viterbi_lp.hip).  Three shapes: 4, 3 and 2 columns per lane.

3 columns per lane (round 6: 834 strips for a 160 kb pair where 4 columns make 626 and 2 columns 1 251 -- more than the
1 024 SIMDs) do not divide the 32-bit decision words of the 4- and 2-column layout, so that shape keeps its decision bits
PER COLUMN: nine accumulators (A, B, C of each column) in pinned consecutive registers, one group per 16-step block,
stored LANE-major with three instructions -- [A0 A1 A2] and [B0 B1 B2] as dwordx3, the three 16-bit C words packed into a
dwordx2 (common.hpp: "3 columns per lane"; a vector memory instruction costs a lone wavefront ~4 ordinary ones, and the
walk along a diagonal finds a lane's three columns in one cache line).  Same five bits per cell, same deposits.

The strip's right boundary leaves as ONE store per step: [Z : X] of lane 63's last column (v28 and a copy of X in v29, which
is free once column 0 has read the Z handed in) into the interleaved boundary array of viterbi_lp.hip, and a chunk of 16
rows arrives with one dwordx2 load per lane.

One step of one lane (its 4 columns of one row), in the reference's evaluation order
(src/lib/align_pair.cc:97-124, gap_len 1) -- the same fp32 operations as viterbi_cell.hpp's
25-instruction cell, 19 instructions per cell:
    M  = diag + s                         v_add_f32 (cell 0: DPP from the left lane)
    [z1:m1] = M + [go:ng]                 v_pk_add_f32
    [z2:i1] = I + [ge:gs]                 v_pk_add_f32
    [x1:y1] = m1 + [ng:go]                v_pk_add_f32
    [x2:y2] = D + [gs:ge]                 v_pk_add_f32
    [x3:y3] = i1 + [ng:go]                v_pk_add_f32
    Z = max(z1, z2); X = max3(x1, x2, x3); Y = max3(y1, y2, y3)
    decision bits: signs of z2-z1, [x1-X : y1-Y], [x2-X : y2-Y] (v_sub_f32, 2 x v_pk_add_f32 with neg),
    deposited with five v_alignbit_b32 -- the same five bits in the same accumulators and order as
    viterbi_cell.hpp, so the traceback (common.hpp) reads the same layout.
Hand-off between lanes: DPP wave_shr:1; lane 0 takes the strip's left boundary from lane j of the
16-row chunk registers with a row_shl:j DPP confined to lanes 0-3 (the wave_shr that follows
overwrites lanes 1-3).  The diagonal input is the left lane's X of its last column from BEFORE its
previous step: cell 3 ping-pongs its [X:Y] pair between two register pairs instead of copying.

usage: python coati_amd/csrc/gen_viterbi_lp.py [out]   (default: viterbi_lp_block.inc next to itself)
"""
from pathlib import Path

# ---- physical VGPRs (pairs must be even-aligned)
CA, CB = 2, 4                  # [go:ng], [ge:gs]
PBASE = 8                      # [X:Y] of columns 0..W-2 at v[8+2c]; the last column ping-pongs between the next two pairs
ZL = 28                        # v28 = Z carried along the row, v29 = Z handed in by the left lane
M = 30                         # v30 (v31 unused)
T1, T2, T3, T4, T5 = 32, 34, 36, 38, 40
TS, ADDR = 42, 43
SA, SB = 44, 48                # the two score sets (v44.., v48..): a step reads one and gathers the next step's into the other
PINNED_CLOBBERS = [29, 30, 31] + list(range(32, 44))
# 3 columns per lane: A of columns 0..2 in v52..v54, B in v56..v58, C in v62, v63, v61; v60 = C0 | C1 << 16 at the block's end,
# so that v[52:54], v[56:58], v[60:61] are what the three stores send (tuples even-aligned: gfx90a+)
ACC3_A, ACC3_B, ACC3_C, ACC3_PACK = [52, 53, 54], [56, 57, 58], [62, 63, 61], 60


def pair(r):
    return f"v[{r}:{r + 1}]"


def step(W, j, first, bnd, pairtab):
    """step j (0..15) of a block of W columns per lane; first: lanes take their margin-row state at step == lane;
    bnd: lane 63 publishes the strip's right boundary (dropped by its offset register elsewhere);
    pairtab: the scores of two adjacent columns come from the 16-entry pair table (A/C/G/T descendants) with ONE
    ds_read_b64 -- an LDS instruction costs a lone wavefront ~18 cycles of issue, four times a vector instruction"""
    L = []
    colacc = W == 3  # decision bits per column (aa0.., ab0.., ac0..) instead of three accumulators shared by the columns
    P = [PBASE + 2 * c for c in range(W - 1)]
    PA, PB = PBASE + 2 * (W - 1), PBASE + 2 * W
    even = j % 2 == 0
    src3 = PB if even else PA    # the left lane's X of its last column before its previous step
    cur3 = PA if even else PB    # the last column's state entering this step
    new3 = PB if even else PA    # ... and leaving it
    ar_src, ar_dst = ("%[ara]", "%[arb]") if even else ("%[arb]", "%[ara]")
    sel = f"row_shl:{j}" if j else "quad_perm:[0,1,2,3]"
    lane0 = f"{sel} row_mask:0x1 bank_mask:0x1"
    shr = "wave_shr:1 row_mask:0xf bank_mask:0xf"
    # scores: this step reads the set the step before gathered, and gathers the next step's into the other set -- all
    # W gathers right after the table row is known, a whole step ahead of their use
    cur = (lambda c: f"v{SA + c}") if even else (lambda c: f"v{SB + c}")
    nxt = (lambda c: f"v{SB + c}") if even else (lambda c: f"v{SA + c}")
    L.append("s_waitcnt lgkmcnt(0)")
    if first:
        L.append(f"v_cmp_eq_u32_e32 vcc, {j}, %[lrel]")
        for c in range(W):
            r = P[c] if c < W - 1 else cur3
            L.append(f"v_cndmask_b32_e32 v{r}, v{r}, %[mx{c}], vcc")
            L.append(f"v_cndmask_b32_e32 v{r + 1}, v{r + 1}, %[my{c}], vcc")
    # hand-off: M of column 0, the I input, the table row of the NEXT step
    L.append(f"v_mov_b32_dpp {ar_dst}, %[ach] {lane0}")
    L.append(f"v_mov_b32_dpp {ar_dst}, {ar_src} {shr}")
    L.append(f"v_add_f32_dpp v{M}, %[bx], {cur(0)} {lane0}")
    L.append(f"v_add_f32_dpp v{M}, v{src3}, {cur(0)} {shr}")
    L.append(f"v_mov_b32_dpp v{ZL + 1}, %[bz] {lane0}")
    L.append(f"v_mov_b32_dpp v{ZL + 1}, v{ZL} {shr}")
    if pairtab:
        for h in range(W // 2):  # (pair table rows are twice as long: 2 x the row offset)
            L.append(f"v_lshl_add_u32 v{ADDR}, {ar_dst}, 1, %[blp{h}]")
            L.append(f"ds_read_b64 v[{nxt(2 * h)[1:]}:{int(nxt(2 * h)[1:]) + 1}], v{ADDR}")
        if W % 2:  # the odd column out: a single gather from the plain table
            L.append(f"v_add_u32 v{ADDR}, {ar_dst}, %[bl{W - 1}]")
            L.append(f"ds_read_b32 {nxt(W - 1)}, v{ADDR}")
    else:
        for c in range(W):
            L.append(f"v_add_u32 v{ADDR}, {ar_dst}, %[bl{c}]")
            L.append(f"ds_read_b32 {nxt(c)}, v{ADDR}")
    for c in range(W):
        aa, ab, ac = (f"v{ACC3_A[c]}", f"v{ACC3_B[c]}", f"v{ACC3_C[c]}") if colacc else ("%[aa]", "%[ab]", "%[ac]")
        rd = P[c] if c < W - 1 else cur3
        wr = P[c] if c < W - 1 else new3
        zsel = "op_sel:[1,0] op_sel_hi:[1,1]" if c == 0 else "op_sel:[0,0] op_sel_hi:[0,1]"
        L += [
            f"v_pk_add_f32 {pair(T1)}, {pair(M)}, {pair(CA)} op_sel:[0,0] op_sel_hi:[0,1]",   # [z1:m1] = M + [go:ng]
            f"v_pk_add_f32 {pair(T2)}, {pair(ZL)}, {pair(CB)} {zsel}",                       # [z2:i1] = I + [ge:gs]
            f"v_pk_add_f32 {pair(T3)}, {pair(T1)}, {pair(CA)} op_sel:[1,1] op_sel_hi:[1,0]",  # [x1:y1] = m1 + [ng:go]
            f"v_pk_add_f32 {pair(T4)}, {pair(rd)}, {pair(CB)} op_sel:[1,1] op_sel_hi:[1,0]",  # [x2:y2] = D + [gs:ge]
            f"v_pk_add_f32 {pair(T5)}, {pair(T2)}, {pair(CA)} op_sel:[1,1] op_sel_hi:[1,0]",  # [x3:y3] = i1 + [ng:go]
        ]
        if c < W - 1:
            L.append(f"v_add_f32 v{M}, v{rd}, {cur(c + 1)}")                                  # M of the next column
        L += [
            f"v_max_f32 v{ZL}, v{T1}, v{T2}",                                                 # Z
            f"v_sub_f32 v{TS}, v{T2}, v{T1}",                                                 # z2 - z1
            f"v_max3_f32 v{wr}, v{T3}, v{T4}, v{T5}",                                         # X
            f"v_max3_f32 v{wr + 1}, v{T3 + 1}, v{T4 + 1}, v{T5 + 1}",                         # Y
            f"v_alignbit_b32 {ac}, {ac}, v{TS}, 31",                                          # IM
            f"v_pk_add_f32 {pair(T3)}, {pair(T3)}, {pair(wr)} neg_lo:[0,1] neg_hi:[0,1]",     # [x1-X : y1-Y]
            f"v_pk_add_f32 {pair(T4)}, {pair(T4)}, {pair(wr)} neg_lo:[0,1] neg_hi:[0,1]",     # [x2-X : y2-Y]
            f"v_alignbit_b32 {aa}, {aa}, v{T3}, 31",                                          # M1
            f"v_alignbit_b32 {aa}, {aa}, v{T4}, 31",                                          # M2
            f"v_alignbit_b32 {ab}, {ab}, v{T3 + 1}, 31",                                      # D1
            f"v_alignbit_b32 {ab}, {ab}, v{T4 + 1}, 31",                                      # D2
        ]
    if bnd:  # [Z : X] of the row lane 63 just finished, at float index 1 + 2 * row of the interleaved boundary array
        L.append(f"v_mov_b32 v{ZL + 1}, v{new3}")
        L.append(f"buffer_store_dwordx2 {pair(ZL)}, %[offx], %[rs_out], %[so_out] offen offset:{8 * j + 4} sc1")
    if colacc:
        # the group of this block, lane-major (common.hpp): [A0 A1 A2] at lane * 12, [B0 B1 B2] at 768 + lane * 12,
        # [C0 | C1 << 16, C2] at 1536 + lane * 8 -- 2 048 bytes per group.  s_nop 1: a store of more than 64 bits reads its data
        # registers late, and the next block's deposits write them
        if j == 15:
            L.append(f"v_perm_b32 v{ACC3_PACK}, v{ACC3_C[1]}, v{ACC3_C[0]}, %[sel_lo16]")
            L.append(f"buffer_store_dwordx3 v[{ACC3_A[0]}:{ACC3_A[2]}], %[offb], %[rs_bits], %[so_bits] offen")
            L.append(f"buffer_store_dwordx3 v[{ACC3_B[0]}:{ACC3_B[2]}], %[offb], %[rs_bits], %[so_bits] offen offset:768")
            L.append(f"buffer_store_dwordx2 v[{ACC3_PACK}:{ACC3_PACK + 1}], %[offb2], %[rs_bits], %[so_bits] offen offset:1536")
            L.append("s_nop 1")
        return L
    # decision rows (layout: common.hpp): an A/B dword holds 16/W steps, a group of 32/W steps is 1280 bytes
    ma, mc = 16 // W, 32 // W
    group, q = j // mc, j % mc
    if j % ma == ma - 1:
        base = group * 1280 + (q // ma) * 512
        L.append(f"buffer_store_dword %[aa], %[offb], %[rs_bits], %[so_bits] offen offset:{base}")
        L.append(f"buffer_store_dword %[ab], %[offb], %[rs_bits], %[so_bits] offen offset:{base + 256}")
    if q == mc - 1:
        L.append(f"buffer_store_dword %[ac], %[offb], %[rs_bits], %[so_bits] offen offset:{group * 1280 + 1024}")
    return L


# Step of the block before which the NEXT chunk's loads are issued: as late as the load latency (~0.6 us) allows, so
# that a strip follows its left neighbour more closely (160 kb pair, 4 columns: at step 0 43.3 ms, 8: 42.5, 12: 42.4,
# 14: 43.1, 15: 45.2 -- the block then ends waiting for them; 2-column steps are shorter: 8).
LOAD_AT = {4: 12, 3: 10, 2: 8}


def block(W, first, pairtab):
    bnd = not first
    loads = ["buffer_load_dwordx2 %[nxz], %[vin_x], %[rs_in], 0 offen sc1",
             "buffer_load_ubyte %[na], %[vin_a], %[rs_a], 0 offen"]
    L = []
    vmem_after = 0
    for j in range(16):
        if j == LOAD_AT[W]:
            L += loads
            vmem_after = 0
        s = step(W, j, first, bnd, pairtab)
        vmem_after += sum(1 for x in s if x.startswith("buffer_"))
        L += s
    # the counter retires in issue order: everything up to the chunk loads is done when at most the operations issued
    # after them are outstanding
    L.append(f"s_waitcnt vmcnt({vmem_after}) lgkmcnt(0)")
    return L, vmem_after


def emit(name, lines):
    out = [f"#define {name} \\"]
    for i, x in enumerate(lines):
        end = "" if i + 1 == len(lines) else " \\"
        out.append(f'    "{x}\\n\\t"{end}')
    return "\n".join(out) + "\n"


def main():
    import os
    if "COATI_LP_LOAD_AT" in os.environ:  # (experiment)
        for w in LOAD_AT:
            LOAD_AT[w] = int(os.environ["COATI_LP_LOAD_AT"])
    text = "// GENERATED by gen_viterbi_lp.py -- do not edit (see that script for what the text does)\n"
    for W in (2, 3, 4):
        for pairtab in (False, True):
            first, n_first = block(W, True, pairtab)
            main_, n_main = block(W, False, pairtab)
            tag = f"{W}P" if pairtab else f"{W}"
            text += (f"// {W} columns per lane{', pair table' if pairtab else ''}: instructions per 16-step block: first {len(first)}, main {len(main_)}\n"
                     + emit(f"COATI_LP{tag}_BLOCK_FIRST_ASM", first) + emit(f"COATI_LP{tag}_BLOCK_MAIN_ASM", main_))
            print(f"W={W} pairtab={pairtab}: first block {len(first)} instructions ({n_first} stores), main block {len(main_)} ({n_main} stores)")
    clob = ", ".join(f'"v{r}"' for r in PINNED_CLOBBERS)
    text += f"#define COATI_LP_SCRATCH_CLOBBERS {clob}\n"
    import sys
    out = Path(sys.argv[1]) if len(sys.argv) > 1 else Path(__file__).with_name("viterbi_lp_block.inc")
    out.write_text(text)


if __name__ == "__main__":
    main()
