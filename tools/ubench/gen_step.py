#!/usr/bin/env python3
"""Generate tools/ubench/step_model.hip (round 3): a register-level replay of viterbi_ck's hot loop with its
parts switchable -- the 15-instruction cell (constants in SGPRs or VGPRs), the LDS gather of the next step's
scores (one ds_read_b32 per cell, 64 lanes in 64 random table rows: the kernel's bank-conflict profile), the
hand-off from the left neighbour (3 x readlane + mov + DPP), the checkpoint stores (8 B per lane and step +
16 B per lane every other step = 1 KB per wavefront step, streamed to HBM).  Per variant and wavefronts per
SIMD: shader cycles per cell (s_memtime, per SIMD), the clock the chip held (s_memtime / s_memrealtime) and
the wall-clock ns per cell -- i.e. which part costs issue cycles and which part costs CLOCK.
Workgroups of 256 x wps threads, one per CU, so that wps wavefronts per SIMD are resident by construction.
"""
from pathlib import Path
import sys

ZL, DIAG, LDS = 8, 29, 31
YP, XB, S = 40, 72, 88


def _regs(c, parity):
    xin = f"v{YP+2*c+1}" if parity == 0 else f"v{XB+c}"
    xout = f"v{XB+c}" if parity == 0 else f"v{YP+2*c+1}"
    return xin, xout, f"v{YP+2*c}"


def cell(c, parity, vconst, lds):
    xin, xout, y = _regs(c, parity)
    diag = f"v{DIAG}" if c == 0 else _regs(c - 1, parity)[0]
    t0, t1, t2, t3, t4, zl, s = "v10", "v11", "v12", "v13", "v14", f"v{ZL}", f"v{S+c}"
    ng, gs, go, ge = ("v2", "v3", "v4", "v5") if vconst else ("s4", "s5", "s6", "s7")
    L = []
    if lds == 3:  # the cell with PACKED adds (viterbi_lp's form without the decision bits): is it cheaper in ENERGY? (the
        # fill runs at the 1 400 W cap: tools/power_probe.sh)
        P = f"v[{YP+2*c}:{YP+2*c+1}]"  # [Y : X] pair of this column (roles as in _regs: y = YP+2c, xin/xout alternate)
        L.append("s_waitcnt lgkmcnt(15)")
        L += [f"v_add_f32 v10, {diag}, {s}",
              "v_pk_add_f32 v[12:13], v[10:11], v[2:3] op_sel:[0,0] op_sel_hi:[0,1]",
              f"v_pk_add_f32 v[14:15], v[{ZL}:{ZL+1}], v[4:5] op_sel:[0,0] op_sel_hi:[0,1]",
              f"v_add_u32 {s}, v{LDS}, v7",
              "v_pk_add_f32 v[18:19], v[12:13], v[2:3] op_sel:[1,1] op_sel_hi:[1,0]",
              f"v_pk_add_f32 v[20:21], {P}, v[4:5] op_sel:[0,0] op_sel_hi:[0,1]",
              "v_pk_add_f32 v[22:23], v[14:15], v[2:3] op_sel:[1,1] op_sel_hi:[1,0]",
              f"v_max_f32 {zl}, v12, v14",
              f"v_max3_f32 {xout}, v18, v20, v22",
              f"v_max3_f32 {y}, v19, v21, v23",
              f"ds_read_b32 {s}, {s}"]
        return L
    if lds == 2:  # pair table: ONE ds_read_b64 per two columns (issued after the odd column's cell, address from the even one)
        if c % 2 == 0:
            L.append("s_waitcnt lgkmcnt(7)")
        L += [f"v_add_f32 {t0}, {diag}, {s}", f"v_add_f32 {t1}, {ge}, {zl}", f"v_add_f32 {t2}, {gs}, {zl}",
              f"v_add_f32 {t3}, {go}, {t0}", f"v_add_f32 {t0}, {ng}, {t0}"]
        if c % 2 == 0:
            L.append(f"v_add_lshl_u32 v21, v{LDS}, v7, 1")
        L += [f"v_add_f32 {t4}, {ng}, {t2}", f"v_max_f32 {zl}, {t3}, {t1}", f"v_add_f32 {t1}, {gs}, {y}",
              f"v_add_f32 {t3}, {ng}, {t0}", f"v_add_f32 {t2}, {go}, {t2}", f"v_max3_f32 {xout}, {t3}, {t1}, {t4}",
              f"v_add_f32 {t1}, {ge}, {y}", f"v_add_f32 {t0}, {go}, {t0}", f"v_max3_f32 {y}, {t0}, {t1}, {t2}"]
        if c % 2 == 1:
            L.append(f"ds_read_b64 v[{S+c-1}:{S+c}], v21")
        return L
    if lds:
        L.append("s_waitcnt lgkmcnt(15)")  # the read issued 16 cells ago (this column, previous step) has landed
    L += [f"v_add_f32 {t0}, {diag}, {s}", f"v_add_f32 {t1}, {ge}, {zl}", f"v_add_f32 {t2}, {gs}, {zl}",
          f"v_add_f32 {t3}, {go}, {t0}", f"v_add_f32 {t0}, {ng}, {t0}", f"v_add_u32 {s}, v{LDS}, v7",
          f"v_add_f32 {t4}, {ng}, {t2}", f"v_max_f32 {zl}, {t3}, {t1}", f"v_add_f32 {t1}, {gs}, {y}",
          f"v_add_f32 {t3}, {ng}, {t0}", f"v_add_f32 {t2}, {go}, {t2}", f"v_max3_f32 {xout}, {t3}, {t1}, {t4}",
          f"v_add_f32 {t1}, {ge}, {y}", f"v_add_f32 {t0}, {go}, {t0}", f"v_max3_f32 {y}, {t0}, {t1}, {t2}"]
    if lds:
        L.append(f"ds_read_b32 {s}, {s}")
    return L


def handoff(parity, mode=1):
    """mode 1: what the kernel's compiler emits per value (v_readlane -> SGPR, s_nop, v_mov from the SGPR as the DPP's
    `old`, v_mov_dpp wave_shr:1); 2: the DPP half only; 3: the readlane half only; 4: lane 0's value kept in a VGPR
    that rotates by one lane per step (v_mov_dpp wave_rol:1) instead of a v_readlane; 5: mode 1 for ONE value"""
    L = []
    vals = ((9, DIAG, 22), (ZL, ZL, 23), (39, 39, 24))
    if mode == 5:
        vals = vals[:1]
    for src, dst, rot in vals:
        if mode in (1, 5):
            L += [f"v_readlane_b32 s20, v{src}, 5", "s_nop 0", f"v_mov_b32 v{dst}, s20", f"v_mov_b32_dpp v{dst}, v{src} wave_shr:1 row_mask:0xf bank_mask:0xf"]
        elif mode == 2:
            L += [f"v_mov_b32 v{dst}, v20", f"v_mov_b32_dpp v{dst}, v{src} wave_shr:1 row_mask:0xf bank_mask:0xf"]
        elif mode == 3:
            L += [f"v_readlane_b32 s20, v{src}, 5", "s_nop 0", f"v_mov_b32 v{dst}, s20"]
        elif mode == 4:
            L += [f"v_mov_b32_dpp v{rot}, v{rot} wave_rol:1 row_mask:0xf bank_mask:0xf", f"v_mov_b32 v{dst}, v{rot}",
                  f"v_mov_b32_dpp v{dst}, v{src} wave_shr:1 row_mask:0xf bank_mask:0xf"]
    if mode == 6:
        return [f"v_mov_b32 v31, v{6 if parity else 17}"]
    if mode == 7:
        return ["v_mov_b32 v29, v9"]
    if mode == 8:
        return ["v_mov_b32 v8, v9"]
    if mode == 9:
        return ["v_mov_b32 v39, v9"]
    if mode == 10:
        return ["s_nop 0"]
    if mode == 11:
        return ["v_mov_b32_dpp v29, v9 row_shr:1 row_mask:0xf bank_mask:0xf"]
    if mode == 12:
        return ["ds_bpermute_b32 v29, v19, v9", "s_waitcnt lgkmcnt(0)"]
    if mode == 13:
        return ["ds_bpermute_b32 v29, v19, v9", "ds_bpermute_b32 v8, v19, v8", "ds_bpermute_b32 v39, v19, v39", "s_waitcnt lgkmcnt(0)"]
    if mode == 14:
        return ["v_readlane_b32 s20, v9, 5"]
    if mode == 15:
        return ["v_readfirstlane_b32 s20, v9"]
    if mode == 16:
        return ["v_mov_b32_dpp v29, v9 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf"]
    if mode == 17:
        return ["v_mov_b32_dpp v29, v9 wave_shr:1 row_mask:0xf bank_mask:0xf"]
    if mode == 18:
        return ["v_mov_b32_dpp v29, v9 row_bcast:15 row_mask:0xa bank_mask:0xf"]
    if mode == 19:
        return ["v_writelane_b32 v29, s4, 0"]
    if mode == 20:
        # hand-off through per-wavefront LDS slots: lane l writes (X, Z | row) for its right neighbour to slot l + 1 and
        # reads slot l; slot 0 of a step comes from a 64-entry ring that is filled once per chunk (lane 0's address
        # register advances by one ring entry per step, the others stay) -- no DPP, no v_readlane
        return ["ds_write_b64 v25, v[8:9] offset:12", "ds_write_b32 v25, v39 offset:20", "v_add_u32 v26, v26, v27",
                "s_waitcnt lgkmcnt(0)", "ds_read_b64 v[22:23], v26", "ds_read_b32 v24, v26 offset:8", "s_waitcnt lgkmcnt(0)",
                "v_mov_b32 v29, v22", "v_mov_b32 v8, v23", f"v_mov_b32 v31, v{6 if parity else 17}"]
    if mode == 21:
        return ["ds_bpermute_b32 v29, v19, v9", "ds_bpermute_b32 v8, v19, v8", "ds_bpermute_b32 v39, v19, v39", "s_waitcnt lgkmcnt(0)",
                f"v_mov_b32 v31, v{6 if parity else 17}"]
    # the table row of the next step differs per step: alternate between two rows of the lane
    L += [f"v_mov_b32 v31, v{6 if parity else 17}"]
    return L


def stores(parity, mode):
    """mode 1: the kernel's volume (8 B per lane and step + 16 B per lane every other step), streamed to HBM;
    mode 2: the same instructions into a 4 KB window (stays in L2: what do the store INSTRUCTIONS cost?);
    mode 3: half the volume (8 B per lane and step only); mode 4: a quarter (8 B per lane every other step)"""
    if mode == 5:
        # the kernel's instructions, but only the 16 lanes of a sliding window store (banded checkpoints): per step a
        # per-lane counter advances and a compare makes the lane mask
        L = ["v_add_u32 v28, 1, v28", "v_cmp_gt_u32 vcc, 16, v28", "s_and_saveexec_b64 s[44:45], vcc",
             "buffer_store_dwordx2 v[8:9], v15, s[40:43], s18 offen"]
        if parity:
            L.append("buffer_store_dwordx4 v[40:43], v16, s[40:43], s18 offen offset:1024")
        L += ["s_mov_b64 exec, s[44:45]", "v_and_b32 v28, 63, v28", "s_add_u32 s18, s18, 0x800", "s_and_b32 s18, s18, 0xfffff"]
        return L
    L = []
    if mode in (1, 2, 3) or parity:
        L.append("buffer_store_dwordx2 v[8:9], v15, s[40:43], s18 offen")
    if parity and mode in (1, 2):
        L.append("buffer_store_dwordx4 v[40:43], v16, s[40:43], s18 offen offset:1024")
    L += ["s_add_u32 s18, s18, 0x800", f"s_and_b32 s18, s18, {'0xfff' if mode == 2 else '0xfffff'}"]
    return L


def body(vconst, lds, hand, st, cf=False):
    out = []
    for parity in (0, 1):
        if hand:
            out += handoff(parity, hand)
        if st:
            out += stores(parity, st)
        for c in range(16):
            out += cell(c, parity, vconst, lds)
    if cf:  # conflict-free gather: every lane reads its own bank (address = lane * 4 + 128 * entry)
        out = [x.replace("v_add_u32 v31", "v_add_u32 v31").replace(f"v{LDS}, v7", f"v{LDS}, v18") for x in out]
    return out


VARIANTS = {
    "cell sgpr": body(False, False, False, 0),
    "cell vgpr": body(True, False, False, 0),
    "cell vgpr +lds": body(True, True, False, 0),
    "cell vgpr +lds conflict-free": body(True, True, False, 0, cf=True),
    "cell vgpr +handoff": body(True, False, 1, 0),
    "cell vgpr +handoff dpp only": body(True, False, 2, 0),
    "cell vgpr +handoff readlane only": body(True, False, 3, 0),
    "cell vgpr +handoff rotate": body(True, False, 4, 0),
    "cell vgpr +handoff one value": body(True, False, 5, 0),
    "cell vgpr +mov v31": body(True, False, 6, 0),
    "cell vgpr +mov diag": body(True, False, 7, 0),
    "cell vgpr +mov zl": body(True, False, 8, 0),
    "cell vgpr +mov v39": body(True, False, 9, 0),
    "cell vgpr +s_nop": body(True, False, 10, 0),
    "cell vgpr +dpp row_shr": body(True, False, 11, 0),
    "cell vgpr +bpermute x1": body(True, False, 12, 0),
    "cell vgpr +bpermute x3": body(True, False, 13, 0),
    "cell vgpr +readlane bare": body(True, False, 14, 0),
    "cell vgpr +readfirstlane": body(True, False, 15, 0),
    "cell vgpr +dpp quad_perm": body(True, False, 16, 0),
    "cell vgpr +dpp wave_shr bare": body(True, False, 17, 0),
    "cell vgpr +dpp row_bcast15": body(True, False, 18, 0),
    "cell vgpr +writelane": body(True, False, 19, 0),
    "cell vgpr +slot handoff": body(True, False, 20, 0),
    "step vgpr, banded stores": body(True, True, 1, 5),
    "step vgpr, bperm, banded stores": body(True, True, 21, 5),
    "step vgpr, slot handoff": body(True, True, 20, 1),
    "step vgpr, bpermute handoff": body(True, True, 21, 1),
    "step vgpr, slot, stores quarter": body(True, True, 20, 4),
    "step vgpr, slot, no stores": body(True, True, 20, 0),
    "step vgpr, rotate handoff": body(True, True, 4, 1),
    "cell vgpr +stores HBM": body(True, False, False, 1),
    "cell vgpr +stores L2": body(True, False, False, 2),
    "cell vgpr +stores half": body(True, False, False, 3),
    "cell vgpr +stores quarter": body(True, False, False, 4),
    "step vgpr (all)": body(True, True, True, 1),
    "step vgpr, packed cell, banded stores": body(True, 3, 1, 5),
    "cell vgpr packed +lds": body(True, 3, False, 0),
    "step vgpr, pair gathers": body(True, 2, True, 1),
    "step vgpr, pair gathers, banded stores": body(True, 2, 1, 5),
    "cell vgpr +pair gathers": body(True, 2, False, 0),
    "step sgpr (all)": body(False, True, True, 1),
    "step vgpr, lds cf": body(True, True, True, 1, cf=True),
    "step vgpr, stores half": body(True, True, True, 3),
    "step vgpr, lds cf, half": body(True, True, True, 3, cf=True),
    "step vgpr, lds cf, quarter": body(True, True, True, 4, cf=True),
    "step vgpr, lds cf, no st": body(True, True, True, 0, cf=True),
}


def kernel(idx, lines):
    n_valu = len([x for x in lines if x.startswith("v_")])
    asm = "\\n\\t\"\n        \"".join(lines)
    clobbers = ", ".join(f'"v{i}"' for i in range(2, 104))
    init = "".join('"v_mov_b32 v%d, %%[seed]\\n\\t"' % i for i in range(8, 104))
    sinit = "".join('"v_add_u32 v%d, v31, v7\\n\\t"' % (S + c) for c in range(16))
    return n_valu, f'''
template <bool CF> __global__ __launch_bounds__(1024) void k{idx}(unsigned long long* stamps, float* out, float seed, float ng, float gs, float go, float ge, int iters, char* scratch) {{
    __shared__ float tab[183 * 17 + 64 + 16 * 208];
    for(int i = threadIdx.x; i < 183 * 17 + 64 + 16 * 208; i += blockDim.x) tab[i] = -0.001f * i;
    const unsigned hbase = (183 * 17 + 64) * 4 + (threadIdx.x / 64) * 832;  // this wavefront's hand-off slots
    __syncthreads();
    float r;
    unsigned t0l, t0h, t1l, t1h, r0l, r0h, r1l, r1h;
    const unsigned wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) / 64);
    const unsigned long long sp = reinterpret_cast<unsigned long long>(scratch) + static_cast<unsigned long long>(wave) * (1ull << 20);
    const unsigned sp_lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(sp)), sp_hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(sp >> 32));
    const unsigned lane = threadIdx.x & 63;
    // rows: multiples of 68 bytes (the kernel's 17-float stride) in the normal variants; in the conflict-free variants the
    // column part is lane * 4 (mod 128) and adding a row (a multiple of 128 after the mask below) keeps the bank
    const unsigned row0 = CF ? (((lane * 37 + 11 + wave * 5) % 90) * 128) : ((lane * 37 + 11 + wave * 5) % 183) * 68;
    const unsigned row1 = CF ? (((lane * 53 + 29 + wave * 3) % 90) * 128) : ((lane * 53 + 29 + wave * 3) % 183) * 68;
    const unsigned col = (lane * 7 + (lane >> 3)) & 3, cfcol = (lane & 31) * 4;
    asm volatile(
        "s_mov_b32 s4, %[ng]\\n\\t s_mov_b32 s5, %[gs]\\n\\t s_mov_b32 s6, %[go]\\n\\t s_mov_b32 s7, %[ge]\\n\\t"
        "s_mov_b32 s40, %[splo]\\n\\t s_and_b32 s41, %[sphi], 0xffff\\n\\t s_mov_b32 s42, 0x200000\\n\\t s_mov_b32 s43, 0x00020000\\n\\t s_mov_b32 s18, 0\\n\\t"
        {init}
        "v_mov_b32 v2, s4\\n\\t v_mov_b32 v3, s5\\n\\t v_mov_b32 v4, s6\\n\\t v_mov_b32 v5, s7\\n\\t"
        "v_mov_b32 v31, %[row0]\\n\\t v_mov_b32 v6, %[row0]\\n\\t v_mov_b32 v17, %[row1]\\n\\t v_lshlrev_b32 v7, 2, %[col]\\n\\t v_mov_b32 v18, %[cfcol]\\n\\t"
        "v_lshlrev_b32 v15, 3, %[lane]\\n\\t v_lshlrev_b32 v16, 4, %[lane]\\n\\t v_lshlrev_b32 v19, 2, %[lane]\\n\\t v_subrev_u32 v19, 4, v19\\n\\t v_and_b32 v19, 0xff, v19\\n\\t"
        "v_mul_u32_u24 v25, 12, %[lane]\\n\\t v_add_u32 v25, %[hbase], v25\\n\\t v_mov_b32 v26, v25\\n\\t v_mov_b32 v27, 0\\n\\t v_mov_b32 v28, %[lane]\\n\\t"
        {sinit}
        "s_mov_b32 s8, %[iters]\\n\\t"
        "s_memtime s[20:21]\\n\\t s_memrealtime s[24:25]\\n\\t s_waitcnt lgkmcnt(0)\\n\\t"
        "s_mov_b32 s28, s20\\n\\t s_mov_b32 s29, s21\\n\\t s_mov_b32 s30, s24\\n\\t s_mov_b32 s31, s25\\n\\t"
        "1:\\n\\t"
        "{asm}\\n\\t"
        "s_sub_u32 s8, s8, 1\\n\\t s_cmp_lg_u32 s8, 0\\n\\t s_cbranch_scc1 1b\\n\\t"
        "s_waitcnt lgkmcnt(0)\\n\\t"
        "s_memtime s[22:23]\\n\\t s_memrealtime s[26:27]\\n\\t s_waitcnt lgkmcnt(0)\\n\\t"
        "v_mov_b32 %[t0l], s28\\n\\t v_mov_b32 %[t0h], s29\\n\\t v_mov_b32 %[t1l], s22\\n\\t v_mov_b32 %[t1h], s23\\n\\t"
        "v_mov_b32 %[r0l], s30\\n\\t v_mov_b32 %[r0h], s31\\n\\t v_mov_b32 %[r1l], s26\\n\\t v_mov_b32 %[r1h], s27\\n\\t"
        "s_waitcnt vmcnt(0)\\n\\t"
        "v_add_f32 %[r], v8, v29\\n\\t v_add_f32 %[r], %[r], v40\\n\\t v_add_f32 %[r], %[r], v41\\n\\t v_add_f32 %[r], %[r], v72\\n\\t"
        "v_add_f32 %[r], %[r], v10\\n\\t v_add_f32 %[r], %[r], v11\\n\\t v_add_f32 %[r], %[r], v12\\n\\t v_add_f32 %[r], %[r], v13"
        : [r] "=&v"(r), [t0l] "=&v"(t0l), [t0h] "=&v"(t0h), [t1l] "=&v"(t1l), [t1h] "=&v"(t1h), [r0l] "=&v"(r0l), [r0h] "=&v"(r0h), [r1l] "=&v"(r1l), [r1h] "=&v"(r1h)
        : [seed] "v"(seed - 0.01f * threadIdx.x), [ng] "s"(ng), [gs] "s"(gs), [go] "s"(go), [ge] "s"(ge), [iters] "s"(iters), [splo] "s"(sp_lo), [sphi] "s"(sp_hi),
          [row0] "v"(row0), [row1] "v"(row1), [col] "v"(col), [cfcol] "v"(cfcol), [lane] "v"(lane), [hbase] "v"(hbase)
        : "s4", "s5", "s6", "s7", "s8", "s18", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s40", "s41", "s42", "s43", "s44", "s45", "scc", "vcc", "memory", {clobbers});
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xffff] = r + tab[threadIdx.x & 63];
    if((threadIdx.x & 63) == 0) {{
        stamps[2 * wave] = ((static_cast<unsigned long long>(t1h) << 32) | t1l) - ((static_cast<unsigned long long>(t0h) << 32) | t0l);
        stamps[2 * wave + 1] = ((static_cast<unsigned long long>(r1h) << 32) | r1l) - ((static_cast<unsigned long long>(r0h) << 32) | r0l);
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
    runs.append(f'    if (run("{n}", k{i}<{"true" if "cf" in n or "conflict-free" in n else "false"}>, {cnt})) return 1;')
src = f'''// GENERATED by gen_step.py -- do not edit, do not commit.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHECK(x) do {{ hipError_t e = (x); if (e != hipSuccess) {{ printf("%s: %s\\n", #x, hipGetErrorString(e)); return 1; }} }} while (0)
{"".join(kernels)}
template <typename K> int run(const char* name, K kern, int n_valu) {{
    static float* d_out = nullptr; static unsigned long long* d_st = nullptr; static char* d_scratch = nullptr;
    if(!d_out) {{ CHECK(hipMalloc(&d_out, sizeof(float) * 65536)); CHECK(hipMalloc(&d_st, 16 * 1024 * 8)); CHECK(hipMalloc(&d_scratch, 4096ull << 20)); }}
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-24s %3d VALU per 2 steps |", name, n_valu);
    for (int wps : {{1, 2, 3, 4}}) {{
        const int iters = 1500;  // x 2 steps x 16 cells
        for (int rep = 0; rep < 60; ++rep)  // warm the clock: ~0.3 s of the same kernel back to back
            hipLaunchKernelGGL(kern, dim3(256), dim3(256 * wps), 0, 0, d_st, d_out, -1.0f, -0.001f, -1.79f, -6.9f, -0.18f, iters, d_scratch);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(256), dim3(256 * wps), 0, 0, d_st, d_out, -1.0f, -0.001f, -1.79f, -6.9f, -0.18f, iters, d_scratch);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> st(2 * 1024 * wps);
        CHECK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> cyc, ghz;
        for (int w = 0; w < 1024 * wps; ++w) {{ cyc.push_back(double(st[2 * w])); ghz.push_back(double(st[2 * w]) / double(st[2 * w + 1]) * 0.1); }}
        std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
        const double cells = 32.0 * iters;  // per wavefront
        printf("  w%d: %5.1f cyc/cell %.2f GHz %5.2f ns/cell |", wps, cyc[cyc.size() / 2] / cells / wps, ghz[ghz.size() / 2], ms * 1e6 / (cells * wps));
    }}
    printf("\\n");
    return 0;
}}
int main() {{
{chr(10).join(runs)}
    return 0;
}}
'''
Path(__file__).with_name("step_model.hip").write_text(src)
print("generated", len(names), "kernels")
