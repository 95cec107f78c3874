"""Print the interesting parts of a bench.py JSON line.  usage: python tools/show_bench.py <file>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = lambda v: {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if not isinstance(b, (dict, list))}
print("value", round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 3), "roofline", r(d["roofline"]))
print("band", d["roofline"].get("band"))
for k, v in (d.get("pcie_inclusive") or {}).items():
    if isinstance(v, dict): print("pcie", k, r(v))
print("two streams", (d.get("two_stream_pipeline") or {}).get("gcups"))
e = d.get("extra") or {}
if "sample" in e: print("sample", e["sample"])
if "cli_batch" in e: print("cli", round(e["cli_batch"]["s"], 3), e["cli_batch"].get("stage_ms"))
if "long_pair" in e: print("long pair", r(e["long_pair"]))
for c in (e.get("reference_suite") or {}).get("cases", []):
    print("  ", c["case"], "pair ms", round(c["pair_ms"], 3), "batch64 gcups", round(c["batch64_gcups"], 1), c["pair_bit_exact"], c["batch64_bit_exact"])
for k, v in (e.get("band_sensitivity") or {}).items():
    if isinstance(v, dict): print("band bag", k, {a: round(b["ms"], 2) for a, b in v.items() if isinstance(b, dict)}, v.get("same_bits"))
if "power" in e: print("power", e["power"])
if d.get("cpu_baseline"): print("cpu", r(d["cpu_baseline"]))
