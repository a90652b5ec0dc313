"""ctypes bindings of oracle/liboracle.so -- TEST INFRASTRUCTURE (the CPU restatement used as checker)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
ORC_PATH = os.path.join(ORC_DIR, "liboracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", ORC_DIR, "restate"])


if not os.path.exists(ORC_PATH):
    build()
lib = C.CDLL(ORC_PATH)


class OrcGraph(C.Structure):
    _fields_ = [
        ("n_vertices", C.c_int32), ("n_levels", C.c_int32), ("R", C.c_int32),
        ("level_off", C.c_void_p), ("out_off", C.c_void_p), ("out_dst", C.c_void_p), ("out_w", C.c_void_p),
        ("hom_off", C.c_void_p), ("het_off", C.c_void_p), ("hom_col", C.c_void_p), ("het_col", C.c_void_p),
    ]


class OrcResult(C.Structure):
    _fields_ = [
        ("value", C.c_int32), ("s_het", C.c_int32), ("n_p1", C.c_int32), ("n_p2", C.c_int32),
        ("p1_from", C.c_void_p), ("p1_to", C.c_void_p), ("p2_from", C.c_void_p), ("p2_to", C.c_void_p),
        ("cap", C.c_int32), ("cells", C.c_uint64), ("relaxations", C.c_uint64),
    ]


lib.orc_hash_kmer.restype = C.c_uint64
lib.orc_hash_kmer.argtypes = [C.c_char_p, C.c_int]
lib.orc_minimizers.restype = C.c_int64
lib.orc_minimizers.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
lib.orc_compute_hashes.restype = C.c_int64
lib.orc_compute_hashes.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64]
lib.orc_sketch_reads.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
lib.orc_free.argtypes = [C.c_void_p]
lib.orc_free.restype = None
lib.orc_dp_solve_diploid.argtypes = [C.POINTER(OrcGraph), C.POINTER(OrcResult), C.c_void_p]
lib.orc_inter_union2x2.argtypes = [C.c_void_p, C.c_int] * 4
lib.orc_symdiff_union2x2.argtypes = [C.c_void_p, C.c_int] * 4


class OrcHapGraph(C.Structure):
    _fields_ = [("n_vertices", C.c_int32), ("R", C.c_int32), ("out_off", C.c_void_p), ("out_dst", C.c_void_p), ("out_w", C.c_void_p),
                ("n_colours", C.c_void_p)]


lib.orc_dp_haploid.argtypes = [C.POINTER(OrcHapGraph), C.c_void_p, C.c_void_p, C.c_void_p]


def dp_haploid(R, out_off, out_dst, out_w, n_colours):
    """literal scatter DP (approximator.cpp:44-72): returns dp, back_vtx, back_r as [n, R+1] int32 arrays"""
    out_off = np.ascontiguousarray(out_off, np.int64); out_dst = np.ascontiguousarray(out_dst, np.int32)
    out_w = np.ascontiguousarray(out_w, np.uint8); n_colours = np.ascontiguousarray(n_colours, np.int32)
    n = out_off.size - 1
    g = OrcHapGraph(n, R, out_off.ctypes.data, out_dst.ctypes.data if out_dst.size else 0, out_w.ctypes.data if out_w.size else 0, n_colours.ctypes.data)
    arrs = [np.zeros((n, R + 1), np.int32) for _ in range(3)]
    assert lib.orc_dp_haploid(C.byref(g), *(a.ctypes.data for a in arrs)) == 0
    return arrs


def hash_kmer(s: bytes) -> int:
    return lib.orc_hash_kmer(s, len(s))


def minimizers(seq: bytes, k, w):
    n = lib.orc_minimizers(seq, len(seq), k, w, None, None, 0)
    h = np.zeros(max(n, 1), np.uint64)
    p = np.zeros(max(n, 1), np.int64)
    lib.orc_minimizers(seq, len(seq), k, w, h.ctypes.data, p.ctypes.data, n)
    return h[:n], p[:n]


def compute_hashes(read: bytes, k, w):
    cap = max(len(read), 1)
    out = np.zeros(cap, np.uint64)
    n = lib.orc_compute_hashes(read, len(read), k, w, out.ctypes.data, cap)
    return out[:n]


def sketch_reads(reads, k, w):
    bases = b"".join(reads)
    off = np.zeros(len(reads) + 1, np.int64)
    np.cumsum([len(r) for r in reads], out=off[1:])
    hp, cp, n = C.c_void_p(), C.c_void_p(), C.c_int64()
    rc = lib.orc_sketch_reads(bases, off.ctypes.data, len(reads), k, w, C.byref(hp), C.byref(cp), C.byref(n))
    assert rc == 0
    h = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_uint64)), (max(n.value, 1),))[: n.value].copy()
    c = np.ctypeslib.as_array(C.cast(cp, C.POINTER(C.c_int32)), (max(n.value, 1),))[: n.value].copy()
    lib.orc_free(hp)
    lib.orc_free(cp)
    return h, c


def dp_solve(g, want_digest=False):
    """g: dipgenie_amd.capi.DpGraphArrays (plain numpy container). Returns (value, s_het, p1, p2, cells, relax[, digest])."""
    st = g.as_struct(OrcGraph)
    cap = g.R + 8
    bufs = [np.zeros(cap, np.int32) for _ in range(4)]
    res = OrcResult()
    res.p1_from, res.p1_to, res.p2_from, res.p2_to = (b.ctypes.data for b in bufs)
    res.cap = cap
    dig = np.zeros(g.n_levels, np.uint64) if want_digest else None
    rc = lib.orc_dp_solve_diploid(C.byref(st), C.byref(res), dig.ctypes.data if want_digest else None)
    assert rc == 0, rc
    p1 = [(int(bufs[0][i]), int(bufs[1][i])) for i in range(res.n_p1)]
    p2 = [(int(bufs[2][i]), int(bufs[3][i])) for i in range(res.n_p2)]
    out = dict(value=res.value, s_het=res.s_het, p1=p1, p2=p2, cells=res.cells, relaxations=res.relaxations)
    if want_digest:
        out["digest"] = dig
    return out
