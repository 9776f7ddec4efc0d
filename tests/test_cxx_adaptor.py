"""The C++ mirror of the reference's Decoder interface (include/acg_ldpc_decoder.hpp): compiles with plain
g++ and links against the C-ABI library (CPU); on a GPU box the binary also runs a BP and a QP-ADMM decode."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    import acg_alp_ldpc_amd as A
    A.build()
    exe = str(tmp_path / "cxx_adaptor_check")
    libdir = os.path.join(ROOT, "acg_alp_ldpc_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cxx_adaptor_check.cpp"), "-o", exe, "-L" + libdir,
                           "-lacg_ldpc_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-pthread"])
    return exe


def test_cxx_adaptor_compiles_and_links(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe, os.path.join(ROOT, "data", "H05.txt")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "m=160 n=280 E=860" in out.stdout


@pytest.mark.gpu
def test_cxx_adaptor_decodes_on_gpu(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe, os.path.join(ROOT, "data", "H05.txt")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "BP ok=1 size=280 diff=0" in out.stdout
    assert "QP-ADMM ok=1 size=280 diff=0" in out.stdout
    assert "BP hopeless ok=0 size=0" in out.stdout
