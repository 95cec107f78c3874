#!/bin/bash
# End-to-end wall time of the two CLIs for one pair (process start, HIP runtime init, model, GPU, output).
cd "$(dirname "$0")/.."
printf ">A\nCTCTGGATAGTG\n>B\nCTATAGTG\n" > /tmp/e1.fasta
python3 - <<'PY'
import sys; sys.path.insert(0,'.')
from coati_amd import host
a,d=host.synth_raw(0)
open('/tmp/e1k.fasta','w').write(f">A\n{a}\n>B\n{d}\n")
PY
t() { python3 - "$@" <<'PY'
import subprocess, sys, time
t0 = time.perf_counter(); r = subprocess.run(sys.argv[1:], capture_output=True); dt = time.perf_counter() - t0
print(f"{dt:.3f} s  rc={r.returncode}  {' '.join(sys.argv[1:])}")
PY
}
for f in /tmp/e1.fasta /tmp/e1k.fasta; do for i in 1 2 3; do t coati_amd/_build/coati-alignpair $f; done; done
for i in 1 2; do t coati_amd/_build/coati-sample /tmp/e1k.fasta -n 1000 -s 42; done
