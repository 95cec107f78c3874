#!/bin/bash
# Whole-process wall time of `coati-alignpair --batch` on a 10 000-pair FASTA, with the stage timeline.
# usage: tools/cli_batch_time.sh [dir with coati-alignpair and its libraries ...]   (default: coati_amd/_build)
cd "$(dirname "$0")/.."
python3 - <<'PY'
import sys; sys.path.insert(0,'.')
from coati_amd import host
with open('/tmp/p10k.fasta','w') as f:
    for i in range(10000):
        a,d=host.synth_raw(i); f.write(f">a{i}\n{a}\n>d{i}\n{d}\n")
PY
for dir in "${@:-coati_amd/_build}"; do for i in 1 2 3 4; do python3 - "$dir" <<'PY'
import subprocess, time, os, sys
env=dict(os.environ, COATI_HOST_TIMING="1")
t0=time.perf_counter(); r=subprocess.run([sys.argv[1] + "/coati-alignpair","--batch","/tmp/p10k.fasta","-o","/tmp/o.json"],capture_output=True,text=True,env=env); dt=time.perf_counter()-t0
tot=[l for l in r.stderr.splitlines() if "gapped strings" in l]
print(f"{sys.argv[1]}: {dt:.3f} s rc={r.returncode}  in-program {tot[-1].split('total ')[-1] if tot else '?'}")
PY
done; done
