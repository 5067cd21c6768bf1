"""The C ABI without Python (level 1): tests/cabi_smoke.cpp is compiled with hipcc against include/fmmt.h,
linked to the in-tree libfmmt_hip.so and run as a plain process -- fmmt_linear_fwd, fmmt_layernorm_fwd and fmmt_window_attn_fwd
(shifted windows, mask tensor) in parity mode against CPU loops written in that file."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_smoke(out):
    from facialmmt_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build libfmmt_hip.so first (python -m facialmmt_amd.build)"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi_smoke.cpp"),
           "-L", libdir, "-lfmmt_hip", f"-Wl,-rpath,{libdir}", "-o", out]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return out


@pytest.mark.gpu
def test_cabi_smoke_program_runs_without_python(tmp_path):
    exe = build_smoke(str(tmp_path / "cabi_smoke"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CABI_SMOKE_OK" in r.stdout, r.stdout + r.stderr
    assert r.stdout.count("OK  ") == 3, r.stdout


def test_cabi_smoke_program_links_against_the_header(tmp_path):
    """no GPU: the C program compiles against include/fmmt.h as C++ (so the header is self-contained), links to the shared library
    and resolves its entry points"""
    exe = build_smoke(str(tmp_path / "cabi_smoke"))
    r = subprocess.run([exe, "--symbols-only"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "CABI_SYMBOLS_OK" in r.stdout, r.stdout + r.stderr
