"""ctypes binding of the CPU oracle (oracle/libhagrid_oracle.so) and of the reference-header harness
(oracle/_ref/libhagrid_ref.so).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py -- never by the hagrid_amd package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_LEVELS = 32

HIT_DTYPE = np.dtype([("id", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])
CELL_DTYPE = np.dtype([("min", "<i4", 3), ("begin", "<i4"), ("max", "<i4", 3), ("end", "<i4")])
SMALL_CELL_DTYPE = np.dtype([("min", "<u2", 3), ("max", "<u2", 3), ("begin", "<i4")])


class OBBox(C.Structure):
    _fields_ = [("min", C.c_float * 3), ("pad0", C.c_int), ("max", C.c_float * 3), ("pad1", C.c_int)]


class OGrid(C.Structure):
    _fields_ = [("entries", C.c_void_p), ("ref_ids", C.c_void_p), ("cells", C.c_void_p), ("small_cells", C.c_void_p),
                ("bbox", OBBox), ("dims", C.c_int * 3), ("num_cells", C.c_int), ("num_entries", C.c_int),
                ("num_refs", C.c_int), ("shift", C.c_int), ("num_offsets", C.c_int), ("offsets", C.c_int * MAX_LEVELS)]


class OStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("rays", "rays_hit_grid", "cells", "entry_words", "refs", "sentinels", "hits", "long_list_refs")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build_oracle(force: bool = False) -> None:
    """Compile the oracle (and the reference harness when /root/reference exists)."""
    so = os.path.join(_HERE, "libhagrid_oracle.so")
    src = os.path.join(_HERE, "hagrid_oracle.c")
    ref_so = os.path.join(_HERE, "_ref", "libhagrid_ref.so")
    need = force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)
    need = need or (os.path.isdir("/root/reference/src") and not os.path.exists(ref_so))
    if need:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build_oracle()
        # HAGRID_ORACLE_LIB: another build of the same source, e.g. the AddressSanitizer one of `make -C oracle asan`
        L = C.CDLL(os.environ.get("HAGRID_ORACLE_LIB") or os.path.join(_HERE, "libhagrid_oracle.so"))
        vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
        L.orc_safe_rcp.restype = f32; L.orc_safe_rcp.argtypes = [f32]
        L.orc_prodsign.restype = f32; L.orc_prodsign.argtypes = [f32, f32]
        L.orc_cbrtf.restype = f32; L.orc_cbrtf.argtypes = [f32]
        L.orc_ilog2_i32.restype = i32; L.orc_ilog2_i32.argtypes = [i32]
        L.orc_make_entry.restype = C.c_uint32; L.orc_make_entry.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_tri_bbox.argtypes = [vp, vp]
        L.orc_compute_range.argtypes = [vp, vp, vp, vp]
        L.orc_compute_grid_dims.argtypes = [vp, i32, f32, vp]
        L.orc_lookup_entry.restype = C.c_uint32; L.orc_lookup_entry.argtypes = [vp, i32, vp, vp, vp]
        L.orc_intersect_prim_cell.restype = i32; L.orc_intersect_prim_cell.argtypes = [vp, vp]
        L.orc_intersect_prim_ray.restype = i32; L.orc_intersect_prim_ray.argtypes = [vp, vp, i32, vp]
        L.orc_intersect_prim_ray_uv.restype = i32; L.orc_intersect_prim_ray_uv.argtypes = [vp, vp, i32, vp]
        L.orc_traverse_grid_ex.restype = None; L.orc_traverse_grid_ex.argtypes = [vp, vp, vp, vp, i64, i32, C.c_uint]
        L.orc_grid_init.argtypes = [vp]; L.orc_grid_free.argtypes = [vp]
        L.orc_build_grid.restype = i32; L.orc_build_grid.argtypes = [vp, i32, vp, f32, f32]
        L.orc_merge_grid.restype = i32; L.orc_merge_grid.argtypes = [vp, f32]
        L.orc_flatten_grid.restype = i32; L.orc_flatten_grid.argtypes = [vp]
        L.orc_expand_grid.restype = i32; L.orc_expand_grid.argtypes = [vp, vp, i32]
        L.orc_expand_grid_ex.restype = i32; L.orc_expand_grid_ex.argtypes = [vp, vp, i32, i32]
        L.orc_compress_grid.restype = i32; L.orc_compress_grid.argtypes = [vp]
        L.orc_traverse_grid.argtypes = [vp, vp, vp, vp, i64, vp, vp]
        L.orc_traverse_grid_mt.argtypes = [vp, vp, vp, vp, i64, i32, vp]
        L.orc_traverse_grid_pinned.restype = None; L.orc_traverse_grid_pinned.argtypes = [vp, vp, vp, vp, i64, i32, vp, i32, vp]
        L.orc_brute_force.argtypes = [vp, i32, vp, vp, i64, i32]
        L.orc_check_grid.restype = i32; L.orc_check_grid.argtypes = [vp, vp, i32, i32, C.c_char_p, i32]
        L.orc_set_cuda_quirks.restype = None; L.orc_set_cuda_quirks.argtypes = [i32]
        L.orc_get_cuda_quirks.restype = i32; L.orc_get_cuda_quirks.argtypes = []
        _lib = L
    return _lib


def ref_lib():
    """The reference-header harness, or None when neither the checkout nor a prebuilt file exists."""
    global _ref
    if _ref is None:
        build_oracle()
        p = os.path.join(_HERE, "_ref", "libhagrid_ref.so")
        if not os.path.exists(p):
            return None
        R = C.CDLL(p)
        vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
        R.ref_safe_rcp.restype = f32; R.ref_safe_rcp.argtypes = [f32]
        R.ref_prodsign.restype = f32; R.ref_prodsign.argtypes = [f32, f32]
        R.ref_ilog2_i32.restype = i32; R.ref_ilog2_i32.argtypes = [i32]
        R.ref_make_entry.restype = C.c_uint32; R.ref_make_entry.argtypes = [C.c_uint32, C.c_uint32]
        R.ref_tri_bbox.argtypes = [vp, vp]
        R.ref_compute_range.argtypes = [vp, vp, vp, vp]
        R.ref_compute_grid_dims.argtypes = [vp, i32, f32, vp]
        R.ref_lookup_entry.restype = C.c_uint32; R.ref_lookup_entry.argtypes = [vp, i32, vp, vp]
        R.ref_intersect_prim_cell.restype = i32; R.ref_intersect_prim_cell.argtypes = [vp, vp]
        R.ref_intersect_prim_ray.restype = i32; R.ref_intersect_prim_ray.argtypes = [vp, vp, i32, vp]
        R.ref_foreach_ref_cell.restype = i32; R.ref_foreach_ref_cell.argtypes = [vp, vp, vp]
        R.ref_foreach_ref_small.restype = i32; R.ref_foreach_ref_small.argtypes = [vp, vp, vp]
        R.ref_brute_force.argtypes = [vp, i32, vp, vp, i64, i32]
        _ref = R
    return _ref


_ref_uvs = None


def ref_lib_uvs():
    """The reference-header harness compiled with -DCOMPUTE_UVS (prims.h:285-288); only intersect_prim_ray is used."""
    global _ref_uvs
    if _ref_uvs is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libhagrid_ref_uvs.so")
        R = C.CDLL(path)
        R.ref_intersect_prim_ray.restype = C.c_int; R.ref_intersect_prim_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _ref_uvs = R
    return _ref_uvs


ANY_HIT, UVS = 1, 2


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Grid:
    """Owns an OGrid built by the oracle; exposes its arrays as numpy views (copies on request)."""

    def __init__(self):
        self.g = OGrid()
        lib().orc_grid_init(C.byref(self.g))
        self._external = None  # keeps numpy arrays alive for grids assembled from arrays

    def __del__(self):
        try:
            if self._external is None:
                lib().orc_grid_free(C.byref(self.g))
        except Exception:
            pass

    # -- passes -------------------------------------------------------------------------------
    @staticmethod
    def build(tris: np.ndarray, top_density: float = 0.12, snd_density: float = 2.4) -> "Grid":
        tris = np.ascontiguousarray(tris, dtype=np.float32)
        G = Grid()
        rc = lib().orc_build_grid(_p(tris), tris.shape[0], C.byref(G.g), top_density, snd_density)
        if rc != 0:
            raise RuntimeError(f"orc_build_grid failed: {rc}")
        return G

    def merge(self, alpha: float = 0.995):
        lib().orc_merge_grid(C.byref(self.g), alpha); return self

    def flatten(self):
        lib().orc_flatten_grid(C.byref(self.g)); return self

    def expand(self, tris: np.ndarray, iters: int = 3, subset_only: bool = True):
        tris = np.ascontiguousarray(tris, dtype=np.float32)
        lib().orc_expand_grid_ex(C.byref(self.g), _p(tris), iters, 1 if subset_only else 0); return self

    def compress(self) -> bool:
        return bool(lib().orc_compress_grid(C.byref(self.g)))

    @staticmethod
    def full(tris, top_density=0.12, snd_density=2.4, alpha=0.995, exp_iters=3, compress=False) -> "Grid":
        """build + merge + flatten + expand (+ compress), the sequence of main.cpp:500-506."""
        G = Grid.build(tris, top_density, snd_density).merge(alpha).flatten().expand(tris, exp_iters)
        if compress:
            G.compress()
        return G

    @staticmethod
    def from_arrays(entries, ref_ids, cells, small_cells, bbox_min, bbox_max, dims, shift, offsets) -> "Grid":
        """Wrap arrays (e.g. downloaded from the GPU) so the oracle can traverse / check them."""
        G = Grid()
        keep = []
        entries = np.ascontiguousarray(entries, dtype=np.uint32); keep.append(entries)
        ref_ids = np.ascontiguousarray(ref_ids, dtype=np.int32); keep.append(ref_ids)
        G.g.entries = entries.ctypes.data; G.g.ref_ids = ref_ids.ctypes.data
        if cells is not None:
            cells = np.ascontiguousarray(cells); keep.append(cells); G.g.cells = cells.ctypes.data
            G.g.num_cells = cells.shape[0]
        if small_cells is not None:
            small_cells = np.ascontiguousarray(small_cells); keep.append(small_cells); G.g.small_cells = small_cells.ctypes.data
            G.g.num_cells = small_cells.shape[0]
        for i in range(3):
            G.g.bbox.min[i] = float(bbox_min[i]); G.g.bbox.max[i] = float(bbox_max[i]); G.g.dims[i] = int(dims[i])
        G.g.num_entries = entries.shape[0]; G.g.num_refs = ref_ids.shape[0]; G.g.shift = int(shift)
        G.g.num_offsets = len(offsets)
        for i, o in enumerate(offsets):
            G.g.offsets[i] = int(o)
        G._external = keep
        return G

    # -- views --------------------------------------------------------------------------------
    def _view(self, ptr, n, dtype):
        if not ptr or n == 0:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=n)

    @property
    def entries(self): return self._view(self.g.entries, self.g.num_entries, np.uint32)
    @property
    def ref_ids(self): return self._view(self.g.ref_ids, self.g.num_refs, np.int32)
    @property
    def cells(self): return self._view(self.g.cells, self.g.num_cells, CELL_DTYPE) if self.g.cells else None
    @property
    def small_cells(self): return self._view(self.g.small_cells, self.g.num_cells, SMALL_CELL_DTYPE) if self.g.small_cells else None
    @property
    def dims(self): return tuple(self.g.dims)
    @property
    def shift(self): return self.g.shift
    @property
    def offsets(self): return [self.g.offsets[i] for i in range(self.g.num_offsets)]
    @property
    def bbox_min(self): return np.array(list(self.g.bbox.min), dtype=np.float32)
    @property
    def bbox_max(self): return np.array(list(self.g.bbox.max), dtype=np.float32)
    @property
    def num_cells(self): return self.g.num_cells
    @property
    def num_refs(self): return self.g.num_refs
    @property
    def num_entries(self): return self.g.num_entries

    def summary(self) -> dict:
        return {"dims": self.dims, "shift": self.shift, "num_cells": self.num_cells, "num_refs": self.num_refs,
                "num_entries": self.num_entries, "offsets": self.offsets, "compressed": bool(self.g.small_cells)}

    # -- queries ------------------------------------------------------------------------------
    def traverse(self, tris, rays, nthreads: int = 1, want_steps: bool = False, cpus="allowed"):
        """cpus: hardware threads the worker threads are pinned to, thread t to cpus[t % len] ("allowed": every hardware thread this process may
        run on; physical_cpus(): one per physical core; None: not pinned -- on the boxes this runs on unpinned threads stay on the CPU that
        created them and the traversal does not scale at all)"""
        if isinstance(cpus, str):
            import os
            cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
        tris = np.ascontiguousarray(tris, dtype=np.float32); rays = np.ascontiguousarray(rays, dtype=np.float32)
        n = rays.shape[0]
        hits = np.zeros(n, dtype=HIT_DTYPE)
        st = OStats()
        if want_steps:
            steps = np.zeros(n, dtype=np.int32)
            lib().orc_traverse_grid(C.byref(self.g), _p(tris), _p(rays), _p(hits), n, _p(steps), C.byref(st))
            return hits, st.as_dict(), steps
        if nthreads <= 1:
            lib().orc_traverse_grid(C.byref(self.g), _p(tris), _p(rays), _p(hits), n, None, C.byref(st))
        elif cpus is not None and len(cpus):
            c = np.ascontiguousarray(cpus, dtype=np.int32)
            lib().orc_traverse_grid_pinned(C.byref(self.g), _p(tris), _p(rays), _p(hits), n, nthreads, _p(c), int(c.size), C.byref(st))
        else:
            lib().orc_traverse_grid_mt(C.byref(self.g), _p(tris), _p(rays), _p(hits), n, nthreads, C.byref(st))
        return hits, st.as_dict()

    def traverse_ex(self, tris, rays, flags: int, nthreads: int = 1):
        """any-hit / barycentric variants of the traversal"""
        tris = np.ascontiguousarray(tris, dtype=np.float32); rays = np.ascontiguousarray(rays, dtype=np.float32)
        n = rays.shape[0]
        hits = np.zeros(n, dtype=HIT_DTYPE)
        lib().orc_traverse_grid_ex(C.byref(self.g), _p(tris), _p(rays), _p(hits), n, nthreads, flags)
        return hits

    def check(self, tris, coverage: int = 0):
        tris = np.ascontiguousarray(tris, dtype=np.float32)
        msg = C.create_string_buffer(256)
        rc = lib().orc_check_grid(C.byref(self.g), _p(tris), tris.shape[0], coverage, msg, 256)
        return rc, msg.value.decode()


def brute_force(tris, rays, nthreads: int = 1, use_ref: bool = False):
    tris = np.ascontiguousarray(tris, dtype=np.float32); rays = np.ascontiguousarray(rays, dtype=np.float32)
    hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
    if use_ref:
        ref_lib().ref_brute_force(_p(tris), tris.shape[0], _p(rays), _p(hits), rays.shape[0], nthreads)
    else:
        lib().orc_brute_force(_p(tris), tris.shape[0], _p(rays), _p(hits), rays.shape[0], nthreads)
    return hits


def algorithmic_bytes(stats: dict, compressed: bool) -> dict:
    """B_ray and B_walk summed over a batch from the exact counters."""
    s_cell = 16 if compressed else 32
    walk = 4 * stats["entry_words"] + s_cell * stats["cells"]
    total = 48 * stats["rays"] + walk + 52 * stats["refs"] + 4 * stats["sentinels"]
    return {"B_ray": total, "B_walk": walk}


class cuda_quirks:
    """`with cuda_quirks(mask):` -- the oracle's construction as a literal CUDA run with CUB would do it ("as-CUDA" structure mode):
    bit 0 = the rejected half of the level partition in reverse order (cub::DevicePartition::Flagged, parallel.cuh:59-71 -> descending
    reference lists on odd levels, which count_union / merge_refs / is_subset silently mis-handle), bit 1 = expand leaves unprocessed
    cells stale (expand.cu:154-155,181).  Analysis only: the product and the default oracle implement the documented intent; hits are
    the same either way."""
    D1_REVERSED_PARTITION, D2_STALE_EXPAND = 1, 2

    def __init__(self, mask: int):
        self.mask = int(mask)

    def __enter__(self):
        self.old = lib().orc_get_cuda_quirks()
        lib().orc_set_cuda_quirks(self.mask)
        return self

    def __exit__(self, *exc):
        lib().orc_set_cuda_quirks(self.old)
        return False


def physical_cpus():
    """One hardware thread per physical core among those this process may run on (the first sibling of every
    /sys/devices/system/cpu/cpuN/topology/thread_siblings_list): the threads of the all-cores CPU baseline are pinned to these --
    two traversal threads on one core share its load ports and caches and gain little (bench.py `cpu_baseline`)."""
    import os
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    seen, out = set(), []
    for c in allowed:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib); out.append(c)
    return out
