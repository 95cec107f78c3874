"""coati_amd/csrc/glibc_math.hpp (the restatements of glibc's expf / log1pf / logf that make the
Forward and sampling kernels bit-exact) compiled for the HOST and compared with this machine's
libm: the quick sweep of tools/libm_check.cc (every 257th float of the reachable ranges, 14 M
inputs; the exhaustive sweep takes half a minute and is run by hand)."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_libm_restatements_match_the_host_libm(tmp_path):
    exe = tmp_path / "libm_check"
    # the flags of the tool's header: FMA contraction only where the source spells it (the x86-64 FMA
    # build of glibc's expf/logf), none elsewhere
    build = subprocess.run(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off", "-o", str(exe),
                            str(ROOT / "tools" / "libm_check.cc"), "-lm"], capture_output=True, text=True)
    if build.returncode != 0 and "mfma" in build.stderr:
        pytest.skip("host compiler without -mfma")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe), "quick"], capture_output=True, text=True, timeout=600)
    if run.returncode == -4:  # SIGILL: a host CPU without FMA (glibc then dispatches to its non-FMA expf/logf)
        pytest.skip("host CPU without FMA")
    assert run.returncode == 0, run.stdout[-2000:]
    lines = [l for l in run.stdout.splitlines() if "inputs" in l]
    assert len(lines) >= 5 and all(l.rstrip().endswith(" 0 mismatches") for l in lines), run.stdout
