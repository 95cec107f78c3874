"""viterbi_lp's 16-step blocks are generated instruction text (coati_amd/csrc/gen_viterbi_lp.py -> viterbi_lp_block.inc,
both committed): the committed text must be what the committed generator writes."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_viterbi_lp_blocks_match_their_generator(tmp_path):
    out = tmp_path / "viterbi_lp_block.inc"
    r = subprocess.run([sys.executable, str(ROOT / "coati_amd" / "csrc" / "gen_viterbi_lp.py"), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert out.read_text() == (ROOT / "coati_amd" / "csrc" / "viterbi_lp_block.inc").read_text()


def test_block_text_has_the_shapes_the_kernel_names():
    text = (ROOT / "coati_amd" / "csrc" / "viterbi_lp_block.inc").read_text()
    for w in ("2", "3", "4"):
        for tab in ("", "P"):
            for kind in ("FIRST", "MAIN"):
                assert f"#define COATI_LP{w}{tab}_BLOCK_{kind}_ASM" in text
    # every block ends with the counted wait that makes the chunk loads (its oldest vector-memory operations) visible
    assert text.count("s_waitcnt vmcnt(") == 12
