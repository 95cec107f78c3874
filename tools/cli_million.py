#!/usr/bin/env python3
"""`coati-alignpair --batch` on a large synthetic FASTA (default 1 000 000 pairs of 1 kb, BASELINE configs[4]'s input
size) end to end: wall time of the process, its stage times, pairs/s.  Writes the FASTA and the JSON under /tmp
(2 x ~2 GB) and removes them.   usage: cli_million.py [pairs] [model]  -> one JSON line on stdout"""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import host  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
model = sys.argv[2] if len(sys.argv) > 2 else "mar-ecm"
fa, js = Path("/tmp/coati_cli_million.fasta"), Path("/tmp/coati_cli_million.json")
t0 = time.perf_counter()
with open(fa, "w", buffering=1 << 24) as f:
    for i in range(pairs):
        anc, des = host.synth_raw(i)
        f.write(f">a{i}\n{anc}\n>d{i}\n{des}\n")
t_gen = time.perf_counter() - t0
exe = ROOT / "coati_amd" / "_build" / "coati-alignpair"
env = dict(os.environ, COATI_HOST_TIMING="1")
runs = []
for _ in range(2):
    t0 = time.perf_counter()
    pr = subprocess.run([str(exe), "--batch", str(fa), "-m", model, "-o", str(js)], capture_output=True, text=True, env=env, timeout=1800)
    dt = time.perf_counter() - t0
    if pr.returncode != 0:
        raise SystemExit(pr.stderr[-2000:])
    stages = {}
    for line in pr.stderr.splitlines():
        if line.startswith("alignpair --batch: ") and " ms (total" in line:
            name, rest = line[len("alignpair --batch: "):].rsplit(" ms (total", 1)[0].rsplit(" ", 1)
            stages[name] = float(rest)
    runs.append({"seconds": dt, "stage_ms": stages})
best = min(runs, key=lambda r: r["seconds"])
# the output is a JSON array of `pairs` alignments: count the objects without loading 2 GB
n_obj = 0
with open(js, "rb") as f:
    while True:
        chunk = f.read(1 << 24)
        if not chunk:
            break
        n_obj += chunk.count(b'"score":')
print(json.dumps({"what": f"coati-alignpair --batch -m {model} on a {pairs}-pair synthetic FASTA -> JSON file, whole process wall time (best of 2)",
                  "pairs": pairs, "seconds": best["seconds"], "pairs_per_s": pairs / best["seconds"], "stage_ms": best["stage_ms"],
                  "all_runs_seconds": [r["seconds"] for r in runs], "fasta_bytes": fa.stat().st_size, "json_bytes": js.stat().st_size,
                  "alignments_in_output": n_obj, "fasta_generation_seconds": t_gen, "host_threads": os.cpu_count()}))
fa.unlink()
js.unlink()
