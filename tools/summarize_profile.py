#!/usr/bin/env python3
"""Reduce the rocprofv3 output of tools/profile.sh to the small files kept under profiles/:
kernel_stats.csv (as rocprofv3 wrote it), pmc_summary.csv (mean counter value per kernel
dispatch) and traffic.json (HBM bytes per launch of the dominant kernel:
(WRITE_SIZE + 2*FETCH_SIZE) KB * 1024 -- on gfx950 FETCH_SIZE under-reports a coalesced
stream by 2x, MI355X_MICROARCH.md HBM section)."""
import csv
import glob
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path

out = Path(sys.argv[1])
stats = glob.glob(str(out / "stats" / "**" / "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], out / "kernel_stats.csv")
rows = []
per = {}
for counter in ("WRITE_SIZE", "FETCH_SIZE"):
    acc = defaultdict(list)
    for f in glob.glob(str(out / f"pmc_{counter}" / "**" / "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].split("::")[-1]
            acc[name].append(float(r["Counter_Value"]))
    for name, vals in sorted(acc.items()):
        rows.append((name, counter, len(vals), sum(vals) / len(vals)))
        per[(name, counter)] = sum(vals) / len(vals)
with open(out / "pmc_summary.csv", "w") as fh:
    fh.write("kernel,counter,dispatches,mean_value_KB\n")
    for name, counter, n, mean in rows:
        fh.write(f"{name},{counter},{n},{mean:.3f}\n")
traffic = {}
for kern in sorted({k for k, _ in per}):
    w, f = per.get((kern, "WRITE_SIZE")), per.get((kern, "FETCH_SIZE"))
    if w is not None and f is not None:
        traffic[kern] = {"WRITE_SIZE_KB": w, "FETCH_SIZE_KB_raw": f, "bytes_per_launch": (w + 2.0 * f) * 1024.0}
(out / "traffic.json").write_text(json.dumps(traffic, indent=1))
print(json.dumps(traffic))
