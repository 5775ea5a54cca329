"""Host-side mirror of the reference's operator interface for the hot path, over the C ABI.

Same names, argument meaning and error behaviour as the reference's C++ API so that tests read like
the reference's own call sequence (main.cpp:471-506, :535, :410-425):

    mem  = MemManager(keep=True)                        # mem_manager.h:34-119
    tris = mem.upload(host_tris)                        # mem.alloc<Tri> + copy<HST_TO_DEV>
    grid = Grid()
    build_grid(mem, tris, n, grid, 0.12, 2.4)           # build.h:17
    merge_grid(mem, grid, 0.995)                        # build.h:20
    flatten_grid(mem, grid)                             # build.h:25
    expand_grid(mem, grid, tris, 3)                     # build.h:28
    compress_grid(mem, grid)                            # build.h:31 (bool)
    setup_traversal(grid)                               # traverse.h:11
    traverse_grid(grid, tris, rays, hits, num_rays)     # traverse.h:14
    ms = profile(lambda: traverse_grid(...))            # common.h:15

Device buffers are plain integer addresses (what the C ABI takes).  Errors raise HagridError carrying the
"file(line): message" text -- the reference prints that text and abort()s (common.h:103-108).
There is no CPU path here: without the compiled library and a GPU every call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import lib as _lib
from .lib import GridPOD, HagridError, TraversalStats
from .scene import CELL_DTYPE, HIT_DTYPE, SMALL_CELL_DTYPE

_current = None  # the most recently created MemManager (profile / setup_traversal take no manager)


def _check(mem: "MemManager", rc: int, what: str) -> int:
    if rc < 0:
        msg = _lib.load().hagrid_last_error(mem._ctx)
        raise HagridError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
    return rc


class MemManager:
    """Buffer pool on one GPU (reference: MemManager, mem_manager.h:34-119).  `keep` retains freed
    buffers between builds (README.md:52-54 recommends it for build benchmarks)."""

    def __init__(self, keep: bool = False, device: int | None = None):
        global _current
        L = _lib.load()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        h = C.c_void_p()
        rc = L.hagrid_ctx_create(C.byref(h), int(device), 1 if keep else 0)
        if rc != 0 or not h:
            raise HagridError(f"hagrid_ctx_create(device={device}) failed ({rc}): no usable gfx950 device")
        self._L = L
        self._ctx = h
        self.device = int(device)
        _current = self

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.hagrid_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def use_stream(self, stream: int | None):
        """Enqueue all further work on a hipStream_t given as an integer (e.g. torch's cuda_stream)."""
        _check(self, self._L.hagrid_ctx_set_stream(self._ctx, C.c_void_p(stream or 0)), "set_stream")

    def synchronize(self):
        """Waits for everything queued on this manager's stream (hagrid_ctx_synchronize)."""
        _check(self, self._L.hagrid_ctx_synchronize(self._ctx), "synchronize")

    def set_ray_binning(self, mode: int):
        """Extension: 1 = bin each ray batch by grid-entry position before traversal (for incoherent batches);
        2 = automatic: the device bins a batch only if it is neither image-ordered nor coherent (no host round trip)."""
        _check(self, self._L.hagrid_set_ray_binning(self._ctx, int(mode)), "set_ray_binning")

    PRODUCT_OPTIONS = ("expand.subset_only", "traverse.id_is_steps", "traverse.image", "traverse.image_max_mb", "traverse.image_width", "traverse.tile_order")

    def set_option(self, key: str, value: int):
        """The product's options (include/hagrid_amd.h: hagrid_set_option); any other key is a code-path selector of the test library
        (csrc/kat/hagrid_amd_kat.h: hagrid_kat_set_option -- tests and dev tools).  Hits never depend on either."""
        if key in self.PRODUCT_OPTIONS:
            _check(self, self._L.hagrid_set_option(self._ctx, key.encode(), int(value)), f"set_option({key})")
        else:
            _check(self, self._K.hagrid_kat_set_option(self._ctx, key.encode(), int(value)), f"kat_set_option({key})")

    def forget_hints(self):
        """dev tools: the context forgets every ray buffer it has traversed (csrc/kat/hagrid_amd_kat.h: hagrid_kat_forget_hints)"""
        _check(self, self._K.hagrid_kat_forget_hints(self._ctx), "forget_hints")

    def order_state(self, d_rays) -> dict:
        """dev tools / tests: what the context remembers about a ray buffer's tile order (csrc/kat/hagrid_amd_kat.h: hagrid_kat_order_state)"""
        out = (C.c_int32 * 12)(); ms = (C.c_float * 4)()
        _check(self, self._K.hagrid_kat_order_state(self._ctx, C.c_void_p(d_rays), out, ms), "order_state")
        keys = ("slot", "valid", "cooling", "head_tiles", "head_dropped", "n_base", "n_head", "share_choice", "share_samples", "n_all", "head_suggested", "share_launches")
        d = dict(zip(keys, list(out))); d["ms_base"] = round(ms[0], 4); d["ms_head"] = round(ms[1], 4); d["ms_all"] = round(ms[2], 4); d["ms_share_best"] = round(ms[3], 4)
        d["order_loses"] = d["share_samples"] >= 10000; d["share_samples"] %= 10000; d["cooldown"] = d["n_all"] // 100; d["learned_all"] = (d["n_all"] // 10) % 10 == 1; d["n_all"] %= 10
        return d

    def device_info(self) -> dict:
        name = C.create_string_buffer(128); cus = C.c_int(); mem = C.c_int64()
        _check(self, self._L.hagrid_device_info(self._ctx, name, 128, C.byref(cus), C.byref(mem)), "device_info")
        return {"arch": name.value.decode(), "compute_units": cus.value, "total_mem": mem.value}

    # -- alloc / free / copy / zero / one ------------------------------------------------------------
    def alloc(self, nbytes: int) -> int:
        p = self._L.hagrid_mem_alloc(self._ctx, int(nbytes))
        if not p:
            raise HagridError("alloc failed: " + self._L.hagrid_last_error(self._ctx).decode())
        return int(p)

    def free(self, ptr: int | None):
        if ptr:
            _check(self, self._L.hagrid_mem_free(self._ctx, C.c_void_p(ptr)), "free")

    def copy_h2d(self, dst: int, src: np.ndarray):
        src = np.ascontiguousarray(src)
        _check(self, self._L.hagrid_mem_copy_h2d(self._ctx, C.c_void_p(dst), src.ctypes.data_as(C.c_void_p), src.nbytes), "copy h2d")

    def copy_d2h(self, dst: np.ndarray, src: int):
        assert dst.flags["C_CONTIGUOUS"]
        _check(self, self._L.hagrid_mem_copy_d2h(self._ctx, dst.ctypes.data_as(C.c_void_p), C.c_void_p(src), dst.nbytes), "copy d2h")

    def copy_d2d(self, dst: int, src: int, nbytes: int):
        _check(self, self._L.hagrid_mem_copy_d2d(self._ctx, C.c_void_p(dst), C.c_void_p(src), int(nbytes)), "copy d2d")

    def zero(self, ptr: int, nbytes: int):
        _check(self, self._L.hagrid_mem_zero(self._ctx, C.c_void_p(ptr), int(nbytes)), "zero")

    def one(self, ptr: int, nbytes: int):
        _check(self, self._L.hagrid_mem_one(self._ctx, C.c_void_p(ptr), int(nbytes)), "one")

    def usage(self) -> int:
        return int(self._L.hagrid_mem_usage(self._ctx))

    def max_usage(self) -> int:
        return int(self._L.hagrid_mem_max_usage(self._ctx))

    def debug_slots(self):
        self._L.hagrid_mem_debug_slots(self._ctx)

    def bandwidth_probe(self, nbytes: int = 1 << 30, iters: int = 5) -> dict:
        """Measured device copy / triad bandwidth in GB/s (SURVEY.md 8(d) BW_peak, 'measured in the same run')."""
        c = C.c_float(); t = C.c_float()
        _check(self, self._L.hagrid_bandwidth_probe(self._ctx, int(nbytes), int(iters), C.byref(c), C.byref(t)), "bandwidth_probe")
        return {"copy_GBps": float(c.value), "triad_GBps": float(t.value)}

    def image_format(self, grid: "Grid") -> dict:
        """Layout of the traversal image held for `grid`: flat / uniform (table-free) / general (a slim record per voxel-map entry) / slim id bits /
        bytes per record ({} without an image)."""
        f = (C.c_int32 * 4)()
        if self._L.hagrid_traversal_image_info(self._ctx, C.byref(grid.pod), f, None) != 0:
            return {}
        return {"flat": bool(f[0]), "uniform": bool(f[1] & 1), "general": f[0] == 2, "slim_id_bits": int(f[2]), "record_bytes": int(f[3]), "two_layouts": bool(f[1] & 2)}

    def image_record_bytes(self, grid: "Grid") -> int:
        """16 when the traversal image of `grid` holds slim records, else 32."""
        return self.image_format(grid).get("record_bytes", 32)

    def image_bytes(self, grid: "Grid") -> int:
        """Size of the traversal image this manager holds for `grid` (0 when it holds none)."""
        b = C.c_int64(0)
        rc = self._L.hagrid_traversal_image_info(self._ctx, C.byref(grid.pod), None, C.byref(b))
        return int(b.value) if rc == 0 else 0

    @property
    def _K(self):
        """libhagrid_amd_kat.so: known-answer hooks and timed diagnostic kernels (tests and dev tools only)."""
        return _lib.load_kat()

    def build_counts(self) -> dict:
        """Sizes the construction passes of this manager went through since its last build_grid."""
        bc = _lib.BuildCounts()
        _check(self, self._L.hagrid_get_build_counts(self._ctx, C.byref(bc)), "get_build_counts")
        return bc.as_dict()

    # -- conveniences ----------------------------------------------------------------------------------
    def upload(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        p = self.alloc(max(arr.nbytes, 4))
        if arr.nbytes:
            self.copy_h2d(p, arr)
        return p

    def download(self, ptr: int, dtype, count: int) -> np.ndarray:
        out = np.empty(int(count), dtype=dtype)
        if out.nbytes:
            self.copy_d2h(out, ptr)
        return out


class Grid:
    """The reference's `struct Grid` (grid.h:48-62); device pointers are integers."""

    def __init__(self):
        self.pod = GridPOD()
        self.mem: MemManager | None = None

    entries = property(lambda s: s.pod.entries or 0)
    ref_ids = property(lambda s: s.pod.ref_ids or 0)
    cells = property(lambda s: s.pod.cells or 0)
    small_cells = property(lambda s: s.pod.small_cells or 0)
    dims = property(lambda s: tuple(s.pod.dims))
    shift = property(lambda s: int(s.pod.shift))
    num_cells = property(lambda s: int(s.pod.num_cells))
    num_entries = property(lambda s: int(s.pod.num_entries))
    num_refs = property(lambda s: int(s.pod.num_refs))
    offsets = property(lambda s: [int(s.pod.offsets[i]) for i in range(s.pod.num_offsets)])
    bbox_min = property(lambda s: np.array(list(s.pod.bbox_min), dtype=np.float32))
    bbox_max = property(lambda s: np.array(list(s.pod.bbox_max), dtype=np.float32))

    def summary(self) -> dict:
        return {"dims": self.dims, "shift": self.shift, "num_cells": self.num_cells, "num_refs": self.num_refs,
                "num_entries": self.num_entries, "offsets": self.offsets, "compressed": bool(self.small_cells)}

    def free(self, mem: MemManager | None = None):
        """mem.free(grid.entries / cells / ref_ids [/ small_cells]) as main.cpp:496-498 does."""
        mem = mem or self.mem
        for f in ("entries", "cells", "ref_ids", "small_cells"):
            p = getattr(self.pod, f)
            if p:
                mem.free(p)
                setattr(self.pod, f, None)

    def download(self, mem: MemManager | None = None) -> dict:
        mem = mem or self.mem
        d = {"entries": mem.download(self.entries, np.uint32, self.num_entries),
             "ref_ids": mem.download(self.ref_ids, np.int32, self.num_refs),
             "cells": mem.download(self.cells, CELL_DTYPE, self.num_cells) if self.cells else None,
             "small_cells": mem.download(self.small_cells, SMALL_CELL_DTYPE, self.num_cells) if self.small_cells else None}
        d.update(bbox_min=self.bbox_min, bbox_max=self.bbox_max, dims=self.dims, shift=self.shift, offsets=self.offsets)
        return d

    @staticmethod
    def load(mem: MemManager, path: str) -> tuple["Grid", int, int]:
        """A grid file written by hagrid_grid_save (hagrid_cli --save-grid): (grid, device pointer of its triangles, their number)."""
        g = Grid(); g.mem = mem
        tris = C.c_void_p(); n = C.c_int32()
        _check(mem, mem._L.hagrid_grid_load(mem._ctx, path.encode(), C.byref(g.pod), C.byref(tris), C.byref(n)), "grid_load")
        return g, tris.value, n.value

    @staticmethod
    def upload(mem: MemManager, entries, ref_ids, cells, small_cells, bbox_min, bbox_max, dims, shift, offsets) -> "Grid":
        """Assemble a device grid from host arrays (fixtures, the broadcast blob of dist.py)."""
        g = Grid(); g.mem = mem
        g.pod.entries = mem.upload(np.ascontiguousarray(entries, dtype=np.uint32))
        g.pod.ref_ids = mem.upload(np.ascontiguousarray(ref_ids, dtype=np.int32))
        n_cells = 0
        if cells is not None:
            g.pod.cells = mem.upload(cells); n_cells = len(cells)
        if small_cells is not None:
            g.pod.small_cells = mem.upload(small_cells); n_cells = len(small_cells)
        for i in range(3):
            g.pod.bbox_min[i] = float(bbox_min[i]); g.pod.bbox_max[i] = float(bbox_max[i]); g.pod.dims[i] = int(dims[i])
        g.pod.num_cells = n_cells; g.pod.num_entries = len(entries); g.pod.num_refs = len(ref_ids)
        g.pod.shift = int(shift); g.pod.num_offsets = len(offsets)
        for i, o in enumerate(offsets):
            g.pod.offsets[i] = int(o)
        return g


# ---- build.h -----------------------------------------------------------------------------------------

def build_grid(mem: MemManager, tris: int, num_tris: int, grid: Grid, top_density: float, snd_density: float):
    grid.mem = mem
    _check(mem, mem._L.hagrid_build_grid(mem._ctx, C.c_void_p(tris), int(num_tris), C.byref(grid.pod), top_density, snd_density), "build_grid")


def merge_grid(mem: MemManager, grid: Grid, alpha: float):
    _check(mem, mem._L.hagrid_merge_grid(mem._ctx, C.byref(grid.pod), alpha), "merge_grid")


def flatten_grid(mem: MemManager, grid: Grid):
    _check(mem, mem._L.hagrid_flatten_grid(mem._ctx, C.byref(grid.pod)), "flatten_grid")


def expand_grid(mem: MemManager, grid: Grid, tris: int, iters: int):
    _check(mem, mem._L.hagrid_expand_grid(mem._ctx, C.byref(grid.pod), C.c_void_p(tris), int(iters)), "expand_grid")


def compress_grid(mem: MemManager, grid: Grid) -> bool:
    return _check(mem, mem._L.hagrid_compress_grid(mem._ctx, C.byref(grid.pod)), "compress_grid") == 1


def build_all(mem: MemManager, tris: int, num_tris: int, top_density=0.12, snd_density=2.4, alpha=0.995,
              exp_iters=3, compress=False, grid: Grid | None = None) -> Grid:
    """The construction sequence of main.cpp:500-506."""
    grid = grid or Grid()
    build_grid(mem, tris, num_tris, grid, top_density, snd_density)
    merge_grid(mem, grid, alpha)
    flatten_grid(mem, grid)
    expand_grid(mem, grid, tris, exp_iters)
    if compress:
        compress_grid(mem, grid)
    return grid


# ---- traverse.h ---------------------------------------------------------------------------------------

def setup_traversal(grid: Grid):
    mem = grid.mem or _current
    _check(mem, mem._L.hagrid_setup_traversal(mem._ctx, C.byref(grid.pod)), "setup_traversal")


def release_for_traversal(grid: Grid):
    """Extension: frees grid.entries and grid.cells | small_cells once setup_traversal has built a self-contained traversal image;
    traverse_grid keeps working (hagrid_grid_release_for_traversal)."""
    mem = grid.mem or _current
    _check(mem, mem._L.hagrid_grid_release_for_traversal(mem._ctx, C.byref(grid.pod)), "release_for_traversal")


def share_traversal(dst: MemManager, grid: Grid) -> Grid:
    """Extension (hagrid_share_traversal): a descriptor of `grid` for the context `dst` (another stream on the same device) that
    traverses with the traversal image of grid.mem instead of a copy of its own -- independent batches in flight over one image.
    The arrays and the image stay the property of grid.mem."""
    src = grid.mem or _current
    _check(dst, dst._L.hagrid_share_traversal(dst._ctx, src._ctx), "share_traversal")
    g = Grid(); g.mem = dst
    C.memmove(C.byref(g.pod), C.byref(grid.pod), C.sizeof(grid.pod))
    return g


ANY_HIT, UVS = 1, 2      # hagrid_traverse_grid_ex flags


def traverse_grid(grid: Grid, tris: int, rays: int, hits: int, num_rays: int, flags: int = 0):
    """traverse_grid (traverse.h:14); flags: ANY_HIT (shadow rays: stop at the first accepted intersection) | UVS
    (barycentrics stored with the hit, the reference's COMPUTE_UVS build)."""
    mem = grid.mem or _current
    _check(mem, mem._L.hagrid_traverse_grid_ex(mem._ctx, C.byref(grid.pod), C.c_void_p(tris), C.c_void_p(rays), C.c_void_p(hits), int(num_rays), int(flags)), "traverse_grid")


def traverse_grid_stats(grid: Grid, tris: int, rays: int, hits: int, num_rays: int, steps: int = 0) -> dict:
    mem = grid.mem or _current
    st = TraversalStats()
    _check(mem, mem._L.hagrid_traverse_grid_stats(mem._ctx, C.byref(grid.pod), C.c_void_p(tris), C.c_void_p(rays), C.c_void_p(hits),
                                                  int(num_rays), C.c_void_p(steps), C.byref(st)), "traverse_grid_stats")
    return st.as_dict()


def profile(fn, mem: MemManager | None = None) -> float:
    """Milliseconds between two events on the manager's stream around fn() (profile.cu:5-18)."""
    mem = mem or _current
    _check(mem, mem._L.hagrid_profile_begin(mem._ctx), "profile")
    fn()
    ms = mem._L.hagrid_profile_end(mem._ctx)
    if ms < 0:
        raise HagridError("profile failed")
    return float(ms)


def build_algorithmic_bytes(bc: dict) -> dict:
    """Compulsory HBM traffic of the construction per stage, SURVEY.md 8(d) "algorithmic bytes -- build", from the sizes the
    passes recorded (MemManager.build_counts): N triangles, R0 top-level references, R_l / C_l / split_l per level, C / R / E."""
    N, R0 = bc["num_tris"], bc["top_refs"]
    build = (48 * N + 32 * N + 32 * N) + (32 * N + 8 * R0) + (R0 * (8 + 48 + 32) + 8 * R0)          # bboxes, emit, filter
    L = bc["num_levels"]
    for l in range(L):
        R_l, C_l = bc["level_refs"][l], bc["level_cells"][l]
        split_l = R_l - bc["level_kept"][l] if l + 1 < L else 0
        R_next = bc["level_refs"][l + 1] if l + 1 < L else 0
        C_next = bc["level_cells"][l + 1] if l + 1 < L else 0
        build += R_l * (12 + 8) + split_l * (8 + 48 + 32) + 8 * R_next + 32 * C_next + 2 * 4 * C_l
    R, Cc, E = bc["build_refs"], bc["build_cells"], bc["build_entries"]
    build += (8 * R + 8 * R + 32 * Cc + 4 * E) + 2 * 16 * R                                        # concat, sort
    merge = sum(c * (32 + 32) + r * (4 + 4) + 2 * 4 * E + 5 * 4 * c for c, r in zip(bc["merge_cells"], bc["merge_refs"]))
    flatten = 4 * bc["flatten_entries_in"] + 4 * bc["flatten_entries_out"]
    expand = bc["expand_passes"] * bc["expand_cells"] * (32 + 32)
    compress = (32 * bc["compress_cells"] + 16 * bc["compress_cells"] + 8 * bc["compress_refs_out"]) if bc["compressed"] else 0
    out = {"build": int(build), "merge": int(merge), "flatten": int(flatten), "expand": int(expand), "compress": int(compress)}
    out["total"] = sum(out.values())
    return out


def algorithmic_bytes(stats: dict, compressed: bool, record_bytes: int = 32) -> dict:
    """DESIGN.md / BASELINE.md section 4: bytes the algorithm must touch for a batch, from exact counters.  `record_bytes`: size of
    a traversal-image record (32, or 16 for slim records: MemManager.image_record_bytes)."""
    s_cell = 16 if compressed else 32
    walk = 4 * stats["entry_words"] + s_cell * stats["cells"]
    total = 48 * stats["rays"] + walk + 52 * stats["refs"] + 4 * stats["sentinels"]
    # what the traversal-image kernel gathers for the same walk: one record per visited cell (bounds + up to four ids inline), a
    # 48-byte triangle per test, and a 4-byte id only for lists of more than four
    image = 48 * stats["rays"] + record_bytes * stats["cells"] + 48 * stats["refs"] + 4 * stats.get("long_list_refs", 0)
    return {"B_ray": int(total), "B_walk": int(walk), "B_image": int(image), "B_image_walk": int(record_bytes * stats["cells"])}


__all__ = ["MemManager", "Grid", "build_grid", "merge_grid", "flatten_grid", "expand_grid", "compress_grid", "build_all",
           "setup_traversal", "traverse_grid", "traverse_grid_stats", "profile", "algorithmic_bytes", "build_algorithmic_bytes", "HagridError",
           "HIT_DTYPE", "CELL_DTYPE", "SMALL_CELL_DTYPE"]
