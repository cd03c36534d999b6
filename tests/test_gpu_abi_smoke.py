"""The C ABI consumed from plain C (tests/abi_smoke.c, gcc, no Python in the call path) against both libraries, plus the struct-layout guard of the mirrors."""
import ctypes as C
import os
import subprocess

import pytest

from conftest import ROOT


def _build(tmp_path, lib):
    exe = str(tmp_path / ("abi_smoke_" + lib))
    csrc = os.path.join(ROOT, "pigeon.jl_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_smoke.c"), "-o", exe,
                           "-L", csrc, "-l:" + lib, "-lm", "-Wl,-rpath," + csrc])
    return exe


def test_abi_smoke_compiles_and_links_against_the_header(tmp_path):
    """CPU: the header is valid C99 on its own and every symbol the program uses resolves in both libraries (running it needs the GPU)."""
    for lib in ("libpigeon_hip.so", "libpigeon_hip_f32.so"):
        assert os.path.exists(_build(tmp_path, lib))


@pytest.mark.gpu
@pytest.mark.parametrize("lib", ["libpigeon_hip.so", "libpigeon_hip_f32.so"])
def test_abi_smoke_runs(tmp_path, lib):
    r = subprocess.run([_build(tmp_path, lib)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_smoke ok" in r.stdout


def test_mirror_layout_matches_the_library(pkg):
    """pg_abi_layout vs the ctypes mirror (also enforced at load time) and vs the Julia mirror's field list."""
    from pigeon_jl_amd import _lib
    for prec in ("f64", "f32"):
        lib = pkg.load_library(prec)
        n = lib.pg_abi_layout(None, 0)
        out = (C.c_int32 * n)(); lib.pg_abi_layout(out, n)
        assert list(out) == _lib.mirror_layout() and n == 23
    # the Julia struct lists the same fields in the same order as the C struct (types: Cdouble / Int32)
    import re
    hdr = open(os.path.join(ROOT, "include", "pigeon_mpc.h")).read()
    body = re.search(r"typedef struct pg_config \{(.*?)\} pg_config;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    c_fields = [f.strip() for decl in re.findall(r"(?:double|int32_t)\s+([^;]+);", body) for f in decl.split(",")]
    assert c_fields[0] == "N_short" and c_fields[-1] == "cold_guess"
    jl = open(os.path.join(ROOT, "julia", "PigeonMI355X.jl")).read()
    jbody = re.search(r"struct PgConfig(.*?)\nend", jl, re.S).group(1)
    j_fields = re.findall(r"^\s*([A-Za-z_0-9]+)::", jbody, re.M)
    assert j_fields == ["vehicle", "control"] + c_fields, (j_fields, c_fields)
