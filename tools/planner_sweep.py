#!/usr/bin/env python3
"""Planner sweep for the regime between "a few long pairs" and ">= 4 096 pairs" (VERDICT round 3, item 2): batches of
{16, 64, 256, 1 024} synthetic pairs of {2, 4, 8, 16, 32} kb under the planner's own choice and under every forced
gap_len-1 kernel and strip width.  One process: the knobs are read per batch_create.

usage: planner_sweep.py [--out gpurun_out/planner_sweep.json] [--max-cells 6e10] [--pairs 16,64,...] [--kb 2,4,...]
Prints a table (kernel ms, GCUPS) and, per grid point, the planner's time over the best forced choice.
"""
import argparse
import json
import os
import sys
import time
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip, host  # noqa: E402

KNOBS = ("COATI_HIP_VITERBI_CK", "COATI_HIP_VITERBI_BITS", "COATI_HIP_STRIP_W", "COATI_HIP_L1_LP", "COATI_HIP_CK_PARTS")
CONFIGS = [
    ("planner", {}),
    ("ck16", {"COATI_HIP_VITERBI_CK": "1", "COATI_HIP_STRIP_W": "16"}),
    ("ck8", {"COATI_HIP_VITERBI_CK": "1", "COATI_HIP_STRIP_W": "8"}),
    ("ck4", {"COATI_HIP_VITERBI_CK": "1", "COATI_HIP_STRIP_W": "4"}),
    ("l1_16", {"COATI_HIP_VITERBI_BITS": "1", "COATI_HIP_STRIP_W": "16", "COATI_HIP_L1_LP": "0"}),
    ("l1_8", {"COATI_HIP_VITERBI_BITS": "1", "COATI_HIP_STRIP_W": "8", "COATI_HIP_L1_LP": "0"}),
    ("l1_4", {"COATI_HIP_VITERBI_BITS": "1", "COATI_HIP_STRIP_W": "4", "COATI_HIP_L1_LP": "0"}),
    ("lp4", {"COATI_HIP_VITERBI_BITS": "1", "COATI_HIP_STRIP_W": "4"}),
    ("lp2", {"COATI_HIP_VITERBI_BITS": "1", "COATI_HIP_STRIP_W": "2"}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "planner_sweep.json"))
    ap.add_argument("--max-cells", type=float, default=6e10)
    ap.add_argument("--pairs", default="16,64,256,1024")
    ap.add_argument("--kb", default="2,4,8,16,32")
    ap.add_argument("--configs", default=",".join(c for c, _ in CONFIGS))
    ap.add_argument("--launches", type=int, default=3)
    args = ap.parse_args()
    want = set(args.configs.split(","))
    model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
    rows = []
    for kb in (int(x) for x in args.kb.split(",")):
        n_codons = kb * 1000 // 3
        for pairs in (int(x) for x in args.pairs.split(",")):
            cells_est = float(pairs) * (3 * n_codons) ** 2
            if cells_est > args.max_cells:
                continue
            a_cat, a_off, b_cat, b_off = host.synth_encoded(0, pairs, n_codons=n_codons)
            point = {"pairs": pairs, "kb": kb, "results": {}}
            ref_sum = None
            for name, env in CONFIGS:
                if name not in want:
                    continue
                for k in KNOBS:
                    os.environ.pop(k, None)
                os.environ.update(env)
                try:
                    batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
                except hip.CoatiHipError as e:
                    point["results"][name] = {"error": str(e)[:120]}
                    continue
                ms = []
                for r in range(args.launches + 1):
                    t0 = time.perf_counter()
                    batch.viterbi_launch()
                    batch.sync()
                    wall = (time.perf_counter() - t0) * 1e3
                    f, w = batch.viterbi_timing()
                    if r >= 1:
                        ms.append((f + w, wall))
                scores, ops, off, ln = batch.viterbi_fetch()
                crc = zlib.crc32(scores.tobytes())
                for p in range(pairs):
                    crc = zlib.crc32(ops[int(off[p]):int(off[p]) + int(ln[p])].tobytes(), crc)
                if ref_sum is None:
                    ref_sum = crc
                cells = batch.cells
                kms = float(np.median([m[0] for m in ms]))
                point["results"][name] = {"kernel_ms": kms, "wall_ms": float(np.median([m[1] for m in ms])),
                                          "gcups": cells / kms / 1e6, "same_bits": crc == ref_sum,
                                          "device_gb": batch.device_bytes / 1e9}
                point["cells"] = cells
                batch.close()
                model.trim()
            for k in KNOBS:
                os.environ.pop(k, None)
            res = point["results"]
            forced = {k: v["kernel_ms"] for k, v in res.items() if k != "planner" and "kernel_ms" in v}
            if forced and "planner" in res and "kernel_ms" in res["planner"]:
                best = min(forced, key=forced.get)
                point["best_forced"] = best
                point["planner_over_best"] = res["planner"]["kernel_ms"] / forced[best]
            rows.append(point)
            line = f"{pairs:5d} x {kb:2d}kb: " + "  ".join(
                f"{k} {v['kernel_ms']:.2f}ms/{v['gcups']:.0f}{'' if v['same_bits'] else '!'}" if "kernel_ms" in v else f"{k} ERR"
                for k, v in res.items())
            print(line + f"   planner/best = {point.get('planner_over_best', float('nan')):.2f} ({point.get('best_forced')})", flush=True)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
