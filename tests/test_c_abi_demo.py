"""include/cdhip.h is a plain C header: examples/c_abi_demo.c compiles with gcc against it and links
the in-tree libcdhip.so (CPU check); on the GPU box the program solves a Lasso through the C ABI with
no Python in the data path and verifies the reference's KKT assertion (test/lasso.jl:99-100)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "c_abi_demo")


def _build():
    lib = os.path.join(ROOT, "coordinatedescent.jl_amd", "csrc")
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-std=c11", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "c_abi_demo.c"), "-o", EXE, "-L", lib, "-lcdhip",
                    f"-Wl,-rpath,{lib}", "-lm"], check=True)


def test_c_demo_compiles_and_links_as_c11():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_c_demo_runs_on_gpu():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "converged=1" in out.stdout
