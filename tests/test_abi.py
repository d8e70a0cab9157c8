"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports what include/bscnav.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "bscnav.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bsc_[a-z_0-9]+)\s*\(", text)) - {"bsc_draw_fn"})


def test_library_exports_every_declared_symbol():
    import bsc_nav_amd
    L = bsc_nav_amd._lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f"libbscnav.so does not export {name}"
    assert set(declared) == set(bsc_nav_amd._lib.SIGNATURES), "ctypes signatures out of sync with bscnav.h"
    assert b"gfx950" in L.bsc_version()


def test_no_cpu_fallback():
    import torch
    import bsc_nav_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        bsc_nav_amd.VoxelEngine(48, 64, 128, 0.1, -2.0, 4.4, 16, 16)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "bsc-nav_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower() or f == "__never__", f"{f} mentions the oracle"


def test_config_struct_layout_matches_header():
    """Field order of the ctypes mirror == field order of struct bsc_config."""
    import bsc_nav_amd
    text = open(os.path.join(ROOT, "include", "bscnav.h")).read()
    body = re.search(r"typedef struct bsc_config \{(.*?)\} bsc_config;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(",")[0:]:
            m = re.search(r"([A-Za-z_][A-Za-z_0-9]*)\s*(\[\d+\])?\s*$", part.strip())
            names.append(m.group(1))
    assert names == [f[0] for f in bsc_nav_amd._lib.BscConfig._fields_]
