"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol include/neuma_hip.h
declares, the ctypes table covers the header, and host-only entry points behave (no compute without a GPU)."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _header_functions():
    text = (ROOT / "include" / "neuma_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nm_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    from neuma_amd import _lib
    if not _lib.LIB_PATH.exists():
        ge.build()
    lib = _lib.lib()
    names = _header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in neuma_hip.h but not exported"
    missing = [n for n in names if n not in _lib.SIGNATURES]
    assert not missing, f"ctypes table lacks {missing}"
    extra = [n for n in _lib.SIGNATURES if n not in names]
    assert not extra, f"ctypes table has undeclared {extra}"


def test_ctypes_arity_matches_header_prototypes():
    from neuma_amd import _lib
    text = (ROOT / "include" / "neuma_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    seen = 0
    for m in re.finditer(r"\b(nm_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("void", "") else len(args.split(","))
        assert len(_lib.SIGNATURES[name][1]) == n, f"{name}: header has {n} parameters, ctypes table {len(_lib.SIGNATURES[name][1])}"
        seen += 1
    assert seen >= 25


def test_host_only_entry_points():
    from neuma_amd import _lib
    lib = _lib.lib()
    assert lib.nm_version() >= 100
    out = C.c_void_p()
    bad = _lib.nm_mpm_cfg(32, 1e-3, 1, (C.c_float * 3)(0, 0, 0), 0.0, 7)      # invalid bc -> error before any device work
    assert lib.nm_mpm_create(C.byref(bad), C.byref(out)) == -1
    assert b"boundary condition" in lib.nm_last_error()
    assert lib.nm_material_bwd_workspace(1000) >= 5504 * 4
    assert lib.nm_raster_bwd_workspace(10) >= 10 * 9 * 4


def test_product_path_never_imports_the_oracle():
    for p in (ROOT / "neuma_amd").rglob("*.py"):
        src = p.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{p} imports the oracle"
    for p in (ROOT / "neuma_amd" / "csrc").glob("*"):
        if p.suffix in (".hip", ".h", ".cpp"):
            assert "oracle" not in p.read_text().replace("oracle/raster.py", "")


def test_cpu_tensors_are_rejected_loudly():
    import torch
    from neuma_amd import NeumaHipError
    from neuma_amd.svd import SVD
    with pytest.raises(NeumaHipError):
        SVD()(torch.eye(3)[None])


def test_counter_file_carries_the_digest_of_the_sources_it_was_measured_on():
    """profiles/pmc_traffic.json (copied into bench.py's line as *_static fields) names the sources of its library build; bench.py
    compares it with the checkout's and reports `pmc_stale` instead of a traffic figure when they differ."""
    import json
    from pathlib import Path
    from neuma_amd import _lib
    d = json.loads((Path(__file__).resolve().parent.parent / "profiles" / "pmc_traffic.json").read_text())
    dig = _lib.csrc_digest()
    assert len(dig) == 16 and int(dig, 16) >= 0
    assert isinstance(d.get("csrc_digest"), str) and len(d["csrc_digest"]) == 16
    src = (Path(__file__).resolve().parent.parent / "bench.py").read_text()
    assert "pmc_stale" in src and "csrc_digest()" in src
