"""A host written in plain C against include/spx.h (tests/c/abi_driver.c): built with gcc -std=c11 (no Python, no torch,
no C++ in the process), linked to the in-tree libspx.so, run as a child process on the GPU box."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_host_through_the_abi(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as ge
    ge.build()
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    libdir = os.path.join(ROOT, "shiftedproximaloperators.jl_amd", "lib")
    exe = str(tmp_path / "abi_driver")
    subprocess.check_call([gcc, "-std=c11", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(rocm, "include"), os.path.join(ROOT, "tests", "c", "abi_driver.c"),
                           "-L" + libdir, "-lspx", "-Wl,-rpath," + libdir, "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
                           "-Wl,-rpath," + os.path.join(rocm, "lib"), "-lm", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches=0" in r.stdout
