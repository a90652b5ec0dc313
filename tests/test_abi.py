"""CPU: the C-ABI library loads, exports every symbol include/dipgenie_hip.h declares, and refuses to
run without a gfx950 device (no compute calls here)."""
import os
import re

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "dipgenie_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dg_[a-z_0-9]+)\s*\(", txt)))


def test_exports_match_header(built_hip):
    from dipgenie_amd import capi
    syms = header_symbols()
    assert sorted(capi.SYMBOLS) == syms
    for s in syms:
        assert hasattr(capi.lib, s), s


def test_struct_layouts(built_hip):
    import ctypes as C
    from dipgenie_amd import capi
    assert C.sizeof(capi.DpGraph) == 80 and C.sizeof(capi.DpResult) == 72
    assert C.sizeof(capi.DpTiming) == 72 and C.sizeof(capi.SketchTiming) == 24


def test_run_library_exports_match_header(built_hip):
    """libdipgenie_run.so (the host pipeline as a library, include/dipgenie_run.h): every declared entry point is exported, the
    ctypes structs have the header's layout, and it links the HIP library -- never the oracle"""
    import ctypes as C
    import subprocess
    from dipgenie_amd import run_sharded as rs
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dipgenie_run.h")).read(), flags=re.S)
    syms = sorted(set(re.findall(r"\b(dgr_[a-z_0-9]+)\s*\(", txt)))
    assert len(syms) == 9
    lib = rs.RunLib().lib
    for s in syms:
        assert hasattr(lib, s), s
    assert C.sizeof(rs.RunOptions) == 56 and C.sizeof(rs.RunSummary) == 88   # static_assert-ed in run_core.cpp
    out = subprocess.run(["ldd", rs.RUN_LIB], stdout=subprocess.PIPE).stdout.decode()
    assert "libdipgenie_hip.so" in out and "oracle" not in out
    assert "orc_" not in subprocess.run(["nm", "-D", rs.RUN_LIB], stdout=subprocess.PIPE).stdout.decode()


def test_no_cpu_fallback(built_hip):
    import torch
    from dipgenie_amd import capi
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.DgError, match="no CPU fallback|no HIP device"):
        capi.Context(0)


def test_product_does_not_link_oracle(built_hip):
    import subprocess
    for f in (os.path.join(ROOT, "dipgenie_amd", "csrc", "libdipgenie_hip.so"), built_hip, os.path.join(ROOT, "dipgenie_amd", "host", "libdipgenie_run.so")):
        out = subprocess.run(["ldd", f], stdout=subprocess.PIPE).stdout.decode()
        assert "oracle" not in out
        sym = subprocess.run(["nm", "-D", f], stdout=subprocess.PIPE).stdout.decode()
        assert "orc_" not in sym


def test_product_sources_do_not_reference_the_checker():
    """source level: nothing under dipgenie_amd/ includes, imports, links or builds against oracle/ -- the word may appear in comments
    only (C / C++ / HIP: after //; Python: in docstrings and comments; Makefiles: after #)"""
    pkg = os.path.join(ROOT, "dipgenie_amd")
    bad = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            ext = os.path.splitext(f)[1]
            if ext not in (".cpp", ".hpp", ".hip", ".h", ".py") and f != "Makefile":
                continue
            path = os.path.join(dirpath, f)
            txt = open(path, errors="replace").read()
            if ext == ".py":
                code = re.sub(r'"""(.|\n)*?"""', "", txt)
                code = "\n".join(line.split("#", 1)[0] for line in code.split("\n"))
            elif f == "Makefile":
                code = "\n".join(line.split("#", 1)[0] for line in txt.split("\n"))
            else:
                code = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
                code = "\n".join(line.split("//", 1)[0] for line in code.split("\n"))
            for n, line in enumerate(code.split("\n"), 1):
                if re.search(r"oracle|orc_|ORACLE", line):
                    bad.append(f"{os.path.relpath(path, ROOT)}:{n}: {line.strip()}")
    assert not bad, "\n".join(bad)


def test_cli_usage_and_exit_codes(built_hip, tmp_path):
    """process surface (main.cpp:90-110): a missing -g / -r / -o prints the usage on stderr and exits 1, nothing on stdout"""
    import subprocess
    for args in ([], ["-g", "x.gfa"], ["-g", "x.gfa", "-r", "y.fa"], ["-r", "y.fa", "-o", "z.fa"]):
        p = subprocess.run([built_hip, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=tmp_path)
        assert p.returncode == 1, args
        assert p.stdout == b"" and p.stderr.startswith(b"Usage: PHI -g <target.gfa> -r <reads.fa> -o <haplotype.fasta>")
        for flag in (b"-k INT", b"-w INT", b"-R INT", b"-t INT", b"-p INT", b"-T FLOAT"):
            assert flag in p.stderr, flag


def test_cli_fails_loudly_without_gpu(built_hip, tmp_path):
    """the product CLI has no CPU path: on a machine without a gfx950 device it must stop with an error, not fall back"""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    out = tmp_path / "o.fa"
    p = subprocess.run([built_hip, "-p2", "-R2", "-k5", "-w3", "-g", os.path.join(ROOT, "tests", "data", "test.gfa"),
                        "-r", os.path.join(ROOT, "tests", "data", "read.fa"), "-o", str(out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"no CPU fallback" in p.stderr
    assert not out.exists() or out.stat().st_size == 0


def test_integration_md_shows_the_tested_patch():
    """INTEGRATION.md s2.0-2.3 print the code blocks of oracle/ref_hip_patch.py verbatim (the patch that tests/test_gpu_ref_hip.py
    proves on the MI355X with the reference's own host code)"""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    src = open(os.path.join(ROOT, "oracle", "ref_hip_patch.py")).read()
    for name in ("CTX_DEF", "DP_CALL", "SKETCH_READS", "SKETCH_HAP"):
        blk = re.search(name + r" = r\'\'\'\n(.*?)\'\'\'", src, flags=re.S).group(1).rstrip("\n")
        assert blk in doc, name
