"""CPU: the C-ABI library loads and exports every symbol include/spx.h declares; the host mirror's
argument checking works without a GPU; the product has no CPU path."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "spx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(spx_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    import spx_amd
    return spx_amd


def test_header_symbols_are_exported(built):
    syms = _declared_symbols()
    assert len(syms) >= 20
    lib = ctypes.CDLL(built._lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), "libspx.so lacks " + s
    # and the ctypes table binds exactly the declared set
    assert sorted(built._lib.SIGNATURES) == syms
    assert built._lib.load().spx_abi_version() == 1


def test_no_cpu_path(built):
    import torch
    s = built
    x = torch.zeros(4, dtype=torch.float64)
    with pytest.raises(TypeError):
        s.shifted(s.NormL1(1.0), x)
    if not torch.cuda.is_available():
        h = ctypes.c_void_p()
        rc = s._lib.load().spx_ctx_create(0, ctypes.byref(h))
        assert rc != 0 and b"no HIP device" in s._lib.load().spx_last_error()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "shiftedproximaloperators.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".sh")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower(), f


def test_value_type_constructor_errors(built):
    s = built
    with pytest.raises(ValueError):
        s.RootNormLhalf(-1.0)  # src/rootNormLhalf.jl:17-18
    with pytest.raises(ValueError):
        s.GroupNormL2([1.0, -0.5], [range(0, 3), range(3, 6)])  # src/groupNormL2.jl:20-21
    with pytest.raises(ValueError):
        s.GroupNormL2([1.0], [range(0, 3), range(3, 6)])  # :22-23


def test_shard_range_tiles(built):
    s = built
    for n, world, align in ((10, 3, 1), (0, 4, 1), (128 * 1000, 8, 128), (128 * 7, 8, 128), (5, 8, 1)):
        prev = 0
        for r in range(world):
            lo, hi = s.shard_range(n, r, world, align)
            assert lo == prev and lo <= hi and (lo % align == 0 or lo == n)
            prev = hi
        assert prev == n


def test_header_is_plain_c_and_cxx():
    # the drop-in boundary is a C ABI: include/spx.h must compile as C99 and as C++11 on its own
    import shutil
    import subprocess
    hdr = os.path.join(ROOT, "include", "spx.h")
    if shutil.which("gcc"):
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    if shutil.which("g++"):
        subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr])
