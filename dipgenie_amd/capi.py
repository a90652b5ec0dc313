"""ctypes bindings of include/dipgenie_hip.h (libdipgenie_hip.so).  No fallbacks, no oracle imports."""
import ctypes as C
import os
import struct

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DG_LIB") or os.path.join(_HERE, "csrc", "libdipgenie_hip.so")   # DG_LIB: A/B runs of two builds (tools/)

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C dipgenie_amd/csrc` (or __graft_entry__.build()). "
        "dipgenie_amd has no CPU fallback."
    )
# One HIP runtime per process: PyTorch ships its own libamdhip64 / libhsa-runtime64 (ROCm 7.0) under the same sonames as
# the system's (ROCm 7.2) that libdipgenie_hip.so is linked against, and the dynamic loader binds a soname to whichever
# copy came first.  Loaded after this library, torch would get the system copy and then finds "No HIP GPUs"; loaded before
# it, both share torch's copy (what bench.py, dist_sketch.py and the GPU tests need).  So where torch is installed it goes first.
try:
    import torch  # noqa: F401
except ImportError:
    pass
lib = C.CDLL(LIB_PATH)


class DpGraph(C.Structure):
    _fields_ = [
        ("n_vertices", C.c_int32), ("n_levels", C.c_int32), ("R", C.c_int32),
        ("level_off", C.c_void_p), ("out_off", C.c_void_p), ("out_dst", C.c_void_p), ("out_w", C.c_void_p),
        ("hom_off", C.c_void_p), ("het_off", C.c_void_p), ("hom_col", C.c_void_p), ("het_col", C.c_void_p),
    ]


class DpResult(C.Structure):
    _fields_ = [
        ("value", C.c_int32), ("s_het", C.c_int32), ("n_p1", C.c_int32), ("n_p2", C.c_int32),
        ("p1_from", C.c_void_p), ("p1_to", C.c_void_p), ("p2_from", C.c_void_p), ("p2_to", C.c_void_p),
        ("cap", C.c_int32), ("cells", C.c_uint64), ("relaxations", C.c_uint64),
    ]


class DpTiming(C.Structure):
    _fields_ = [
        ("delta_ms", C.c_float), ("forward_ms", C.c_float), ("traceback_ms", C.c_float), ("total_ms", C.c_float),
        ("n_forward_launches", C.c_int64), ("edge_pairs", C.c_uint64), ("colour_entries", C.c_uint64),
        ("state_bytes", C.c_uint64), ("bp_bytes", C.c_uint64), ("delta_bytes", C.c_uint64), ("n_segments", C.c_int32), ("n_chunks", C.c_int32),
    ]


class SketchTiming(C.Structure):
    _fields_ = [("kernel_ms", C.c_float), ("sort_ms", C.c_float), ("total_ms", C.c_float), ("n_emitted", C.c_int64)]


# every symbol include/dipgenie_hip.h declares
SYMBOLS = [
    "dg_create", "dg_destroy", "dg_last_error", "dg_set_stream", "dg_synchronize", "dg_device_info",
    "dg_dp_prealloc", "dg_dp_load_graph", "dg_dp_run", "dg_dp_get_timing", "dg_dp_solve_diploid", "dg_dp_get_level_digest",
    "dg_dp_set_option", "dg_dp_get_launch_profile", "dg_sketch_reads", "dg_sketch_haplotype", "dg_hash_kmers", "dg_free",
    "dg_sketch_get_timing", "dg_sketch_reads_dev", "dg_sketch_count_dictionary_dev", "dg_sketch_merge_runs_dev",
    "dg_sketch_partition_dev", "dg_sketch_rank_dictionary_dev", "dg_sketch_histogram_dev",
    "dg_anchor_begin", "dg_anchor_add_haplotype", "dg_anchor_finish", "dg_dp_solve_haploid", "dg_dp_get_table_digest", "dg_hip_versions", "dg_anchor_add_haplotype_sketched",
    "dg_sketch_set_option", "dg_sketch_get_stat", "dg_sketch_count_rank_dictionary_dev",
    "dg_shard_create", "dg_shard_destroy", "dg_shard_n_ranks", "dg_shard_ctx", "dg_shard_score_reads",
]

lib.dg_create.restype = C.c_void_p
lib.dg_create.argtypes = [C.c_int]
lib.dg_destroy.argtypes = [C.c_void_p]
lib.dg_destroy.restype = None
lib.dg_last_error.restype = C.c_char_p
lib.dg_set_stream.argtypes = [C.c_void_p, C.c_void_p]
lib.dg_synchronize.argtypes = [C.c_void_p]
lib.dg_device_info.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
lib.dg_dp_prealloc.argtypes = [C.c_void_p, C.c_int64]
lib.dg_dp_load_graph.argtypes = [C.c_void_p, C.POINTER(DpGraph)]
lib.dg_dp_run.argtypes = [C.c_void_p, C.POINTER(DpResult)]
lib.dg_dp_get_timing.argtypes = [C.c_void_p, C.POINTER(DpTiming)]
lib.dg_dp_solve_diploid.argtypes = [C.c_void_p, C.POINTER(DpGraph), C.POINTER(DpResult)]
lib.dg_dp_get_level_digest.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
lib.dg_dp_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
lib.dg_dp_get_launch_profile.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
lib.dg_dp_get_table_digest.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.dg_sketch_reads.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
lib.dg_sketch_haplotype.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int, C.c_int,
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
lib.dg_hash_kmers.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int, C.c_void_p]
lib.dg_free.argtypes = [C.c_void_p]
lib.dg_free.restype = None
lib.dg_sketch_get_timing.argtypes = [C.c_void_p, C.POINTER(SketchTiming)]
lib.dg_sketch_reads_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
lib.dg_sketch_count_dictionary_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
lib.dg_sketch_merge_runs_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                         C.c_int64, C.POINTER(C.c_int64)]
lib.dg_sketch_partition_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
lib.dg_sketch_rank_dictionary_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
lib.dg_sketch_histogram_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
lib.dg_sketch_count_rank_dictionary_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
lib.dg_sketch_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
lib.dg_sketch_get_stat.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]


class HapGraph(C.Structure):
    _fields_ = [("n_vertices", C.c_int32), ("R", C.c_int32), ("out_off", C.c_void_p), ("out_dst", C.c_void_p), ("out_w", C.c_void_p), ("n_colours", C.c_void_p)]


lib.dg_dp_solve_haploid.argtypes = [C.c_void_p, C.POINTER(HapGraph), C.c_void_p, C.c_void_p, C.c_void_p]


class AnchorResult(C.Structure):
    _fields_ = [("n_occ", C.c_int64), ("n_vtx", C.c_int64), ("occ_id", C.c_void_p), ("occ_hap", C.c_void_p), ("occ_off", C.c_void_p),
                ("occ_len", C.c_void_p), ("vpool", C.c_void_p), ("n_candidates", C.c_int64), ("n_unstable_groups", C.c_int64)]


lib.dg_anchor_begin.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int, C.c_int]
lib.dg_anchor_add_haplotype.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
lib.dg_anchor_add_haplotype_sketched.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]
lib.dg_anchor_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.POINTER(AnchorResult)]


lib.dg_hip_versions.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]


def hip_versions():
    """(HIP version the library was built against, HIP runtime it is bound to in this process), as "major.minor.patch" strings"""
    a, b = C.c_int(), C.c_int()
    if lib.dg_hip_versions(C.byref(a), C.byref(b)) != 0:
        raise DgError(lib.dg_last_error().decode())
    fmt = lambda v: f"{v // 10000000}.{v // 100000 % 100}.{v % 100000}"
    return fmt(a.value), fmt(b.value)


class DgError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        raise DgError(f"{what} failed (rc={rc}): {lib.dg_last_error().decode()}")


class DpGraphArrays:
    """Levelized DP graph in the dg_dp_graph layout (numpy arrays); loads the host pipeline's .dpg dump."""

    NAMES = ["level_off", "out_off", "out_dst", "out_w", "hom_off", "hom_col", "het_off", "het_col"]
    DTYPES = [np.int32, np.int64, np.int32, np.uint8, np.int64, np.int32, np.int64, np.int32]

    def __init__(self, R, **arrays):
        self.R = int(R)
        for n, dt in zip(self.NAMES, self.DTYPES):
            setattr(self, n, np.ascontiguousarray(arrays[n], dtype=dt))

    @classmethod
    def load(cls, path):
        with open(path, "rb") as f:
            if f.read(8) != b"DGDP0001":
                raise ValueError("not a .dpg file")
            (R,) = struct.unpack("<i", f.read(4))
            arrs = {}
            for n, dt in zip(cls.NAMES, cls.DTYPES):
                (cnt,) = struct.unpack("<Q", f.read(8))
                arrs[n] = np.frombuffer(f.read(cnt * np.dtype(dt).itemsize), dtype=dt).copy()
        return cls(R, **arrs)

    def save(self, path):
        with open(path, "wb") as f:
            f.write(b"DGDP0001")
            f.write(struct.pack("<i", self.R))
            for n in self.NAMES:
                a = getattr(self, n)
                f.write(struct.pack("<Q", a.size))
                f.write(a.tobytes())

    @property
    def n_vertices(self):
        return self.out_off.size - 1

    @property
    def n_levels(self):
        return self.level_off.size - 1

    def as_struct(self, struct_cls=DpGraph):
        g = struct_cls()
        g.n_vertices, g.n_levels, g.R = self.n_vertices, self.n_levels, self.R
        for n in self.NAMES:
            a = getattr(self, n)
            # keep a non-NULL pointer even for empty colour arrays
            setattr(g, n, a.ctypes.data if a.size else np.zeros(1, a.dtype).ctypes.data)
        return g


class DpOutcome:
    def __init__(self, res, p1, p2):
        self.value, self.s_het = res.value, res.s_het
        self.cells, self.relaxations = res.cells, res.relaxations
        self.p1, self.p2 = p1, p2  # lists of (from, to)

    def key(self):
        return (self.value, self.s_het, tuple(self.p1), tuple(self.p2))


def make_result(cap):
    bufs = [np.zeros(cap, np.int32) for _ in range(4)]
    res = DpResult()
    res.p1_from, res.p1_to, res.p2_from, res.p2_to = (b.ctypes.data for b in bufs)
    res.cap = cap
    return res, bufs


def outcome_from(res, bufs):
    p1 = [(int(bufs[0][i]), int(bufs[1][i])) for i in range(res.n_p1)]
    p2 = [(int(bufs[2][i]), int(bufs[3][i])) for i in range(res.n_p2)]
    return DpOutcome(res, p1, p2)


class Context:
    """One dg_ctx (one HIP device + stream)."""

    def __init__(self, device=0):
        self.h = lib.dg_create(device)
        if not self.h:
            raise DgError(f"dg_create({device}) failed: {lib.dg_last_error().decode()}")

    def close(self):
        if self.h:
            lib.dg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        _check(lib.dg_set_stream(self.h, C.c_void_p(stream_ptr)), "dg_set_stream")

    def device_info(self):
        name = C.create_string_buffer(256)
        ncu, hbm = C.c_int(), C.c_int64()
        _check(lib.dg_device_info(self.h, name, 256, C.byref(ncu), C.byref(hbm)), "dg_device_info")
        return name.value.decode(), ncu.value, hbm.value

    # ---- DP ----
    def dp_set_option(self, key, value):
        _check(lib.dg_dp_set_option(self.h, key.encode(), int(value)), "dg_dp_set_option")

    def dp_prealloc(self, nbytes=0):
        """start reserving back-pointer lattice chunks in the background (nbytes <= 0: 60 % of the free HBM)"""
        _check(lib.dg_dp_prealloc(self.h, int(nbytes)), "dg_dp_prealloc")

    def dp_load_graph(self, g):
        self._g = g
        st = g.as_struct()
        _check(lib.dg_dp_load_graph(self.h, C.byref(st)), "dg_dp_load_graph")

    def dp_run(self):
        res, bufs = make_result(self._g.R + 8)
        _check(lib.dg_dp_run(self.h, C.byref(res)), "dg_dp_run")
        return outcome_from(res, bufs)

    def dp_solve(self, g):
        self.dp_load_graph(g)
        return self.dp_run()

    def dp_solve_haploid(self, R, out_off, out_dst, out_w, n_colours):
        """haploid (vertex, r) DP: returns dp, back_vtx, back_r as [n, R+1] int32 arrays"""
        out_off = np.ascontiguousarray(out_off, np.int64); out_dst = np.ascontiguousarray(out_dst, np.int32)
        out_w = np.ascontiguousarray(out_w, np.uint8); n_colours = np.ascontiguousarray(n_colours, np.int32)
        n = out_off.size - 1
        g = HapGraph(n, R, out_off.ctypes.data, out_dst.ctypes.data if out_dst.size else 0, out_w.ctypes.data if out_w.size else 0, n_colours.ctypes.data)
        arrs = [np.zeros((n, R + 1), np.int32) for _ in range(3)]
        _check(lib.dg_dp_solve_haploid(self.h, C.byref(g), *(a.ctypes.data for a in arrs)), "dg_dp_solve_haploid")
        return arrs

    def dp_timing(self):
        t = DpTiming()
        _check(lib.dg_dp_get_timing(self.h, C.byref(t)), "dg_dp_get_timing")
        return t

    def dp_launch_profile(self):
        """{kernel variant: launches} of the last dp_run"""
        buf = C.create_string_buffer(8192)
        _check(lib.dg_dp_get_launch_profile(self.h, buf, 8192), "dg_dp_get_launch_profile")
        return {k: int(v) for k, v in (item.rsplit(":", 1) for item in buf.value.decode().split())}

    TABLES = ["descs", "in_off", "in_edge", "in_dst", "dtrans", "dblk_first", "grp_begin", "dead_cols", "heavy_rows", "rowrec", "rowx", "slots"]

    def dp_table_digest(self):
        """{table: FNV-1a digest} of the tables dg_dp_load_graph built for the resident graph"""
        out = np.zeros(12, np.uint64)
        _check(lib.dg_dp_get_table_digest(self.h, out.ctypes.data, 12), "dg_dp_get_table_digest")
        return dict(zip(self.TABLES, (int(x) for x in out)))

    def dp_level_digest(self, n_levels):
        out = np.zeros(n_levels, np.uint64)
        _check(lib.dg_dp_get_level_digest(self.h, out.ctypes.data, n_levels), "dg_dp_get_level_digest")
        return out

    # ---- sketch ----
    def sketch_reads(self, reads, k, w):
        """reads: list of bytes. Returns (sorted distinct hashes uint64[], n_reads_with_hash int32[])."""
        bases = b"".join(reads)
        off = np.zeros(len(reads) + 1, np.int64)
        np.cumsum([len(r) for r in reads], out=off[1:])
        return self.sketch_reads_flat(bases, off, k, w)

    def sketch_reads_flat(self, bases, off, k, w):
        hp, cp, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        off = np.ascontiguousarray(off, np.int64)
        _check(lib.dg_sketch_reads(self.h, bases, off.ctypes.data, off.size - 1, k, w, C.byref(hp), C.byref(cp), C.byref(n)),
               "dg_sketch_reads")
        h = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_uint64)), (n.value,)).copy() if n.value else np.zeros(0, np.uint64)
        c = np.ctypeslib.as_array(C.cast(cp, C.POINTER(C.c_int32)), (n.value,)).copy() if n.value else np.zeros(0, np.int32)
        lib.dg_free(hp)
        lib.dg_free(cp)
        return h, c

    def sketch_haplotype(self, seq, k, w):
        hp, pp, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        _check(lib.dg_sketch_haplotype(self.h, seq, len(seq), k, w, C.byref(hp), C.byref(pp), C.byref(n)), "dg_sketch_haplotype")
        h = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_uint64)), (n.value,)).copy() if n.value else np.zeros(0, np.uint64)
        p = np.ctypeslib.as_array(C.cast(pp, C.POINTER(C.c_int64)), (n.value,)).copy() if n.value else np.zeros(0, np.int64)
        lib.dg_free(hp)
        lib.dg_free(pp)
        return h, p

    def hash_kmers(self, kmers, k):
        n = len(kmers) // k
        out = np.zeros(n, np.uint64)
        _check(lib.dg_hash_kmers(self.h, kmers, n, k, out.ctypes.data), "dg_hash_kmers")
        return out

    def sketch_set_option(self, name, value):
        _check(lib.dg_sketch_set_option(self.h, name.encode(), int(value)), "dg_sketch_set_option")

    def sketch_stat(self, name):
        v = C.c_int64(0)
        _check(lib.dg_sketch_get_stat(self.h, name.encode(), C.byref(v)), "dg_sketch_get_stat")
        return v.value

    def sketch_timing(self):
        t = SketchTiming()
        _check(lib.dg_sketch_get_timing(self.h, C.byref(t)), "dg_sketch_get_timing")
        return t
