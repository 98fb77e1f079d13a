"""The boundary from COMPILED code: include/dhts.h is a C header (C99 and C++), and a plain-C program -- gcc, the HIP runtime's C API for
device memory, libdhts.so for the work -- drives the operator entry points that replace dMacroForwardLayer.forward / .backward
(reference road/lane/dmacro_lane.py:234-309) and checks them against the C oracle.  CPU: the header parses, the client compiles and
links against every symbol it uses.  GPU: it runs (next state bit for bit, cotangents to 1e-6)."""
import os
import shutil
import subprocess

import pytest

from conftest import PKG, ROOT

HDR = os.path.join(ROOT, "include", "dhts.h")
SRC = os.path.join(ROOT, "tests", "c_client", "macro_step_client.c")
ROCM = "/opt/rocm"


def build(out):
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I%s/include" % ROCM, "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "oracle"), "-o", out, SRC, "-L" + os.path.join(PKG, "csrc"), "-L" + os.path.join(ROOT, "oracle"),
           "-ldhts", "-ldhts_oracle", "-L%s/lib" % ROCM, "-lamdhip64", "-lm", "-Wl,-rpath," + os.path.join(PKG, "csrc"),
           "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath,%s/lib" % ROCM]
    return subprocess.run(cmd, capture_output=True, text=True)


def test_header_is_c99_and_cxx():
    for lang, std in (("c", "-std=c99"), ("c++", "-std=c++17")):
        p = subprocess.run(["gcc" if lang == "c" else "g++", std, "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", lang, HDR], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr


@pytest.mark.skipif(shutil.which("gcc") is None or not os.path.isdir(ROCM), reason="needs gcc and the ROCm headers")
def test_c_client_compiles_and_links(tmp_path, oracle):
    p = build(str(tmp_path / "client"))
    assert p.returncode == 0, p.stderr


@pytest.mark.gpu
def test_c_client_runs(cuda, tmp_path, oracle):
    exe = str(tmp_path / "client")
    p = build(exe)
    assert p.returncode == 0, p.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "0 of 600 next-state entries differ" in r.stdout, r.stdout
