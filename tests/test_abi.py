"""The C-ABI shared library loads and exports every symbol include/coati_hip.h
declares (no compute calls: this runs without a GPU)."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols(header="coati_hip.h"):
    text = (ROOT / "include" / header).read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(coati_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from coati_amd import hip

    lib = hip.load()
    syms = declared_symbols()
    assert len(syms) >= 10
    for name in syms:
        assert hasattr(lib, name), f"{name} declared in coati_hip.h but not exported"
    assert set(hip.EXPORTS) == set(syms)


def test_dist_header_symbols_exported():
    """include/coati_hip_dist.h <-> libcoati_hip_dist.so (links librccl directly; loading it needs no GPU)."""
    from coati_amd import dist

    lib = dist.load()
    syms = declared_symbols("coati_hip_dist.h")
    assert len(syms) >= 8
    for name in syms:
        assert hasattr(lib, name), f"{name} declared in coati_hip_dist.h but not exported"
    assert set(dist.EXPORTS) == set(syms)


def test_version_and_error_string():
    from coati_amd import hip

    lib = hip.load()
    assert lib.coati_hip_version() >= 1
    assert isinstance(lib.coati_hip_last_error(), bytes)


def test_no_fallback_without_device():
    """Without a gfx950 device the product refuses to compute instead of falling back to a CPU path."""
    import numpy as np

    from coati_amd import hip

    if hip.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(hip.CoatiHipError) as ei:
        hip.Model(np.zeros((183, 15), np.float32), [-0.001, -1.79, -6.9, -0.18])
    assert ei.value.code == 2  # COATI_HIP_ENODEVICE


def test_product_does_not_import_oracle():
    for path in (ROOT / "coati_amd").rglob("*"):
        if path.suffix in {".py", ".hip", ".cc", ".cpp", ".h", ".hpp"}:
            text = path.read_text(errors="replace")
            assert "pyoracle" not in text and "coati_oracle" not in text, path
