#!/usr/bin/env python3
"""Generate coati_amd/csrc/viterbi_ck_block.inc: 16 wavefront steps of viterbi_ck's lean fill (16 columns per
lane, single-strip pairs, every lane already started) as ONE block of instruction text.

Why: the C++ step is 256 instructions of cells (16 x (15 + the LDS gather)) and 41 of everything else -- two
exec-masked tests (has this lane started? does lane 63 publish a boundary?), v_readlane + DPP for the three handed
values, loop counters, address arithmetic, waits.  In the block a step is 256 + 9: the lane-0 feed is a row_shl:j DPP
from the 16-row chunk registers (gen_viterbi_lp.py has the trick), the stores take immediate offsets, the X of the
last column alternates between two registers so that the diagonal hand-off needs no copy, and M of column c+1 is
formed in cell c before X of column c is overwritten, so X stays in place (the C++ loop ping-pongs all 16).
The cell is COATI_CELL_LEAN of viterbi_ck.hip, instruction for instruction.

Checkpoints as in viterbi_ck.hip: before the block's first step the lane state (X, Y: eight 16-byte stores, the row
checkpoint of this band), per step what the lane received (diagonal X, left Z: one 8-byte store); a lane outside the
kept band has its offset registers out of range and stores nothing.

RESULT (round 3, measured, NOT kept in the product): bit-exact; but at four wavefronts per SIMD the instruction count
is not what bounds the fill -- +4 % at 40 000 pairs, +1..3 % at 1 000-6 000, level at 10 000 (A/B in one process, box to
box +-2.5 %).  And the block pins 83 vector registers: inside the persistent kernel (another ~45 live values at the
128-register cap of four wavefronts per SIMD) the allocator reloads from scratch between blocks, each reload behind a
wait for the block's 24 stores; as a non-inlined function the lane state crosses the call through memory, +27 % HBM
traffic per launch (7.8 against 6.1 GB), or the whole item's fill moves into the function and its C++ chunks spill.
History: git log -- coati_amd/csrc/gen_viterbi_ck.py (commit 515ea18 has the integrated version).

usage: python tools/experiments/gen_viterbi_ck_block.py   (writes viterbi_ck_block.inc next to itself)
"""
from pathlib import Path

W = 16
XB = 8        # X[c] = v[8 + c]   (quads for the 16-byte stores)
YB = 28       # Y[c] = v[28 + c]
DIAG = 44     # v44 = diagonal input, v45 = Z input of column 0: the pair the per-step checkpoint stores
PINNED_CLOBBERS = [44, 45]


def step(j):
    L = []
    even = j % 2 == 0
    xsrc = "%[xb]" if even else f"v{XB + 15}"   # the left lane's X of column 15 before its previous step
    xnew = "%[xb]" if even else f"v{XB + 15}"   # where this step's X of column 15 goes
    ar_src, ar_dst = ("%[ara]", "%[arb]") if even else ("%[arb]", "%[ara]")
    sel = f"row_shl:{j}" if j else "quad_perm:[0,1,2,3]"
    lane0 = f"{sel} row_mask:0x1 bank_mask:0x1"
    shr = "wave_shr:1 row_mask:0xf bank_mask:0xf"
    L += ["s_waitcnt lgkmcnt(0)",
          f"v_mov_b32_dpp v{DIAG}, %[bx] {lane0}",
          f"v_mov_b32_dpp v{DIAG}, {xsrc} {shr}",
          f"v_mov_b32_dpp v{DIAG + 1}, %[bz] {lane0}",
          f"v_mov_b32_dpp v{DIAG + 1}, %[zl] {shr}",
          f"v_mov_b32_dpp {ar_dst}, %[ach] {lane0}",
          f"v_mov_b32_dpp {ar_dst}, {ar_src} {shr}",
          f"buffer_store_dwordx2 v[{DIAG}:{DIAG + 1}], %[coloff], %[rs_colin], %[so_c{j // 8}] offen offset:{(j % 8) * 512}",
          f"v_add_f32 %[m], v{DIAG}, %[s0]"]
    for c in range(W):
        zin = f"v{DIAG + 1}" if c == 0 else "%[zl]"
        x = f"v{XB + c}" if c < W - 1 else xnew
        y = f"v{YB + c}"
        L += [f"v_add_f32 %[t1], %[ge], {zin}",           # z2 = I + ge
              f"v_add_f32 %[t2], %[gs], {zin}",           # i1 = I + gs
              f"v_add_f32 %[t3], %[go], %[m]",            # z1 = M + go
              f"v_add_f32 %[t0], %[ng], %[m]",            # m1 = M + ng
              f"v_add_u32 %[s{c}], {ar_dst}, %[bl{c}]",   # LDS address of the next step's score
              f"v_add_f32 %[t4], %[ng], %[t2]",           # x3 = i1 + ng
              f"v_max_f32 %[zl], %[t3], %[t1]",           # Z
              f"v_add_f32 %[t1], %[gs], {y}",             # x2 = D + gs
              f"v_add_f32 %[t3], %[ng], %[t0]",           # x1 = m1 + ng
              f"v_add_f32 %[t2], %[go], %[t2]"]           # y3 = i1 + go
        if c < W - 1:
            L.append(f"v_add_f32 %[m], v{XB + c}, %[s{c + 1}]")  # M of the next column (before X is overwritten)
        L += [f"v_max3_f32 {x}, %[t3], %[t1], %[t4]",     # X
              f"v_add_f32 %[t1], %[ge], {y}",             # y2 = D + ge
              f"v_add_f32 %[t0], %[go], %[t0]",           # y1 = m1 + go
              f"v_max3_f32 {y}, %[t0], %[t1], %[t2]"]     # Y
        # (s[c] was consumed by M of this column; s[c+1] by the M formed above: gather in this order)
        L.append(f"ds_read_b32 %[s{c}], %[s{c}]")
    return L


def block():
    L = ["buffer_load_ubyte %[na], %[vin_a], %[rs_a], 0 offen"]
    for q in range(4):
        L.append(f"buffer_store_dwordx4 v[{XB + 4 * q}:{XB + 4 * q + 3}], %[rkoff], %[rs_rk], %[so_r0] offen offset:{q * 1024}")
    for q in range(4):
        L.append(f"buffer_store_dwordx4 v[{YB + 4 * q}:{YB + 4 * q + 3}], %[rkoff], %[rs_rk], %[so_r1] offen offset:{q * 1024}")
    for j in range(16):
        L += step(j)
    after = sum(1 for x in L[1:] if x.startswith("buffer_"))
    L.append(f"s_waitcnt vmcnt({after}) lgkmcnt(0)")
    return L


def emit(name, lines):
    out = [f"#define {name} \\"]
    for i, x in enumerate(lines):
        end = "" if i + 1 == len(lines) else " \\"
        out.append(f'    "{x}\\n\\t"{end}')
    return "\n".join(out) + "\n"


def main():
    b = block()
    clob = ", ".join(f'"v{r}"' for r in PINNED_CLOBBERS)
    text = ("// GENERATED by gen_viterbi_ck.py -- do not edit (see that script for what the text does)\n"
            f"// {len(b)} instructions per 16-step block\n" + emit("COATI_CK16_BLOCK_ASM", b) + f"#define COATI_CK16_SCRATCH_CLOBBERS {clob}\n")
    Path(__file__).with_name("viterbi_ck_block.inc").write_text(text)
    print(f"ck block: {len(b)} instructions")


if __name__ == "__main__":
    main()
