"""The C-ABI shared library: builds for gfx950, loads, and exports exactly what include/bh_engine.h (the drop-in contract) and
include/bh_engine_debug.h (measurement and diagnostics) declare.  No compute calls (runs without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def declared_symbols(header="bh_engine.h"):
    txt = open(os.path.join(REPO, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bh_[a-z_]+)\s*\(", txt)))


def test_library_built_and_exports_every_declared_symbol():
    from bayhunter_amd import engine as E
    assert os.path.exists(E.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(E.LIB_PATH)
    decl = declared_symbols()
    assert 14 <= len(decl) <= 25           # the contract stays small: everything else lives in the debug header
    dbg = declared_symbols("bh_engine_debug.h")
    assert not set(decl) & set(dbg)
    for name in decl + dbg:
        assert hasattr(lib, name), "missing export %s" % name
    assert sorted(E.EXPORTED_SYMBOLS) == decl
    assert sorted(E.DEBUG_SYMBOLS) == dbg
    assert lib.bh_abi_version() == 10


def test_python_constants_mirror_the_header():
    """Enumerations the ctypes layer restates (error codes, target kinds, laws, search modes, ABI version)."""
    from bayhunter_amd import engine as E
    txt = open(os.path.join(REPO, "include", "bh_engine.h")).read() + open(os.path.join(REPO, "include", "bh_engine_debug.h")).read()
    defs = {k: int(v) for k, v in re.findall(r"^#define\s+(BH_[A-Z0-9_]+)\s+(-?\d+)\b", txt, flags=re.M)}
    assert defs["BH_ABI_VERSION"] == 10
    assert (defs["BH_SEARCH_REFERENCE"], defs["BH_SEARCH_FAST"], defs["BH_SEARCH_FAST_RAYLEIGH"]) == (E.SEARCH_REFERENCE, E.SEARCH_FAST, E.SEARCH_FAST_RAYLEIGH) == (0, 1, 2)
    assert (defs["BH_SCAN_STEPS"], defs["BH_SCAN_COUNTED"], defs["BH_SCAN_AUTO"]) == (E.SCAN_STEPS, E.SCAN_COUNTED, E.SCAN_AUTO) == (0, 1, 2)
    assert defs["BH_CHAIN_MAXDEPTH"] == E.BH_CHAIN_MAXDEPTH


def test_library_contains_gfx950_code_object():
    from bayhunter_amd import engine as E
    blob = open(E.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"swd_kernel" in blob and b"rf_synth_kernel" in blob and b"swd_group_kernel" in blob and b"like_kernel" in blob


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(REPO, "bayhunter_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src and "refshim" not in src, f


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from bayhunter_amd import engine as E
    with pytest.raises(E.EngineError):
        E.Engine(0)
    from bayhunter_amd import SurfDisp
    import numpy as np
    sd = SurfDisp(np.linspace(1, 20, 5), "rdispph")
    with pytest.raises(E.EngineError):
        sd.run_model(np.array([5., 0.]), np.array([6., 8.]), np.array([3.5, 4.5]), np.array([2.7, 3.3]))
