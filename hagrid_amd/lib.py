"""ctypes loader for libhagrid_amd.so (the C ABI declared in include/hagrid_amd.h).

The library is linked without a HIP runtime of its own: it binds to the libamdhip64 the process already
holds.  From Python that is PyTorch's copy (PyTorch is the plumbing for device memory, streams and
torch.distributed), which is promoted to the global symbol scope before the library is opened.

There is NO fallback: if the library is missing or cannot be loaded, every product entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HAGRID_AMD_LIB") or os.path.join(HERE, "libhagrid_amd.so")     # the override serves A/B runs of two builds
MAX_LEVELS = 32
ABI_VERSION = 3             # HAGRID_ABI_VERSION of include/hagrid_amd.h this loader was written against

OK, EINVAL, EHIP, ENOMEM, ERANGE, ENODEV = 0, -1, -2, -3, -4, -5


class HagridError(RuntimeError):
    pass


class GridPOD(C.Structure):
    """struct hagrid_grid (include/hagrid_amd.h) == the reference's Grid (grid.h:48-62) as a POD."""
    _fields_ = [("entries", C.c_void_p), ("ref_ids", C.c_void_p), ("cells", C.c_void_p), ("small_cells", C.c_void_p),
                ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3), ("dims", C.c_int32 * 3),
                ("num_cells", C.c_int32), ("num_entries", C.c_int32), ("num_refs", C.c_int32), ("shift", C.c_int32),
                ("num_offsets", C.c_int32), ("offsets", C.c_int32 * MAX_LEVELS)]


class TraversalStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("rays", "rays_hit_grid", "cells", "entry_words", "refs", "sentinels", "hits", "long_list_refs")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


MAX_MERGE_PASSES = 96


class BlobHeader(C.Structure):
    """struct hagrid_blob_header (include/hagrid_amd.h): the first 256 bytes of a grid blob / grid file."""
    _fields_ = [("magic", C.c_uint32), ("version", C.c_uint32), ("dims", C.c_int32 * 3), ("shift", C.c_int32),
                ("num_cells", C.c_int32), ("num_entries", C.c_int32), ("num_refs", C.c_int32), ("num_tris", C.c_int32),
                ("compressed", C.c_int32), ("num_offsets", C.c_int32), ("offsets", C.c_int32 * MAX_LEVELS),
                ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3),
                ("off_entries", C.c_uint64), ("off_cells", C.c_uint64), ("off_refs", C.c_uint64), ("off_tris", C.c_uint64), ("total_bytes", C.c_uint64),
                ("reserved", C.c_uint8 * 16)]


class BuildCounts(C.Structure):
    """struct hagrid_build_counts (include/hagrid_amd.h): sizes the construction passes went through."""
    _fields_ = [("num_tris", C.c_int64), ("top_cells", C.c_int64), ("top_refs", C.c_int64),
                ("num_levels", C.c_int32), ("merge_passes", C.c_int32), ("expand_passes", C.c_int32), ("compressed", C.c_int32),
                ("level_refs", C.c_int64 * MAX_LEVELS), ("level_cells", C.c_int64 * MAX_LEVELS), ("level_kept", C.c_int64 * MAX_LEVELS),
                ("build_cells", C.c_int64), ("build_refs", C.c_int64), ("build_entries", C.c_int64),
                ("merge_cells", C.c_int64 * MAX_MERGE_PASSES), ("merge_refs", C.c_int64 * MAX_MERGE_PASSES),
                ("merged_cells", C.c_int64), ("merged_refs", C.c_int64),
                ("flatten_entries_in", C.c_int64), ("flatten_entries_out", C.c_int64),
                ("expand_cells", C.c_int64), ("compress_cells", C.c_int64), ("compress_refs_out", C.c_int64)]

    def as_dict(self) -> dict:
        d = {}
        for n, t in self._fields_:
            v = getattr(self, n)
            d[n] = int(v) if not hasattr(v, "__len__") else [int(x) for x in v]
        d["level_refs"] = d["level_refs"][:d["num_levels"]]; d["level_cells"] = d["level_cells"][:d["num_levels"]]
        d["level_kept"] = d["level_kept"][:d["num_levels"]]
        d["merge_cells"] = d["merge_cells"][:d["merge_passes"]]; d["merge_refs"] = d["merge_refs"][:d["merge_passes"]]
        return d


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
SIGNATURES = {
    "hagrid_abi_version": (_i32, []),
    "hagrid_debug_sync_enabled": (_i32, []),
    "hagrid_ctx_create": (_i32, [C.POINTER(_vp), _i32, _i32]),
    "hagrid_ctx_destroy": (None, [_vp]),
    "hagrid_ctx_set_stream": (_i32, [_vp, _vp]),
    "hagrid_last_error": (C.c_char_p, [_vp]),
    "hagrid_device_info": (_i32, [_vp, C.c_char_p, _i32, C.POINTER(_i32), C.POINTER(_i64)]),
    "hagrid_mem_alloc": (_vp, [_vp, _sz]),
    "hagrid_mem_free": (_i32, [_vp, _vp]),
    "hagrid_mem_copy_h2d": (_i32, [_vp, _vp, _vp, _sz]),
    "hagrid_mem_copy_d2h": (_i32, [_vp, _vp, _vp, _sz]),
    "hagrid_mem_copy_d2d": (_i32, [_vp, _vp, _vp, _sz]),
    "hagrid_mem_zero": (_i32, [_vp, _vp, _sz]),
    "hagrid_mem_one": (_i32, [_vp, _vp, _sz]),
    "hagrid_mem_usage": (_sz, [_vp]),
    "hagrid_mem_max_usage": (_sz, [_vp]),
    "hagrid_mem_debug_slots": (None, [_vp]),
    "hagrid_bandwidth_probe": (_i32, [_vp, _sz, _i32, C.POINTER(_f32), C.POINTER(_f32)]),
    "hagrid_get_build_counts": (_i32, [_vp, C.POINTER(BuildCounts)]),
    "hagrid_profile_begin": (_i32, [_vp]),
    "hagrid_profile_end": (_f32, [_vp]),
    "hagrid_build_grid": (_i32, [_vp, _vp, _i32, C.POINTER(GridPOD), _f32, _f32]),
    "hagrid_merge_grid": (_i32, [_vp, C.POINTER(GridPOD), _f32]),
    "hagrid_flatten_grid": (_i32, [_vp, C.POINTER(GridPOD)]),
    "hagrid_expand_grid": (_i32, [_vp, C.POINTER(GridPOD), _vp, _i32]),
    "hagrid_compress_grid": (_i32, [_vp, C.POINTER(GridPOD)]),
    "hagrid_grid_blob_bytes": (_sz, [C.POINTER(GridPOD), _i32]),
    "hagrid_grid_pack": (_i32, [_vp, C.POINTER(GridPOD), _vp, _i32, C.POINTER(_vp), C.POINTER(_sz)]),
    "hagrid_grid_unpack": (_i32, [_vp, _vp, _sz, C.POINTER(GridPOD), C.POINTER(_vp), C.POINTER(_i32)]),
    "hagrid_grid_save": (_i32, [_vp, C.POINTER(GridPOD), _vp, _i32, C.c_char_p]),
    "hagrid_grid_load": (_i32, [_vp, C.c_char_p, C.POINTER(GridPOD), C.POINTER(_vp), C.POINTER(_i32)]),
    "hagrid_grid_broadcast": (_i32, [_vp, _vp, _i32, _i32, C.POINTER(GridPOD), C.POINTER(_vp), C.POINTER(_i32)]),
    "hagrid_setup_traversal": (_i32, [_vp, C.POINTER(GridPOD)]),
    "hagrid_grid_release_for_traversal": (_i32, [_vp, C.POINTER(GridPOD)]),
    "hagrid_share_traversal": (_i32, [_vp, _vp]),
    "hagrid_ctx_synchronize": (_i32, [_vp]),
    "hagrid_traverse_grid": (_i32, [_vp, C.POINTER(GridPOD), _vp, _vp, _vp, _i32]),
    "hagrid_traverse_grid_ex": (_i32, [_vp, C.POINTER(GridPOD), _vp, _vp, _vp, _i32, C.c_uint32]),
    "hagrid_traverse_grid_stats": (_i32, [_vp, C.POINTER(GridPOD), _vp, _vp, _vp, _i32, _vp, C.POINTER(TraversalStats)]),
    "hagrid_set_ray_binning": (_i32, [_vp, _i32]),
    "hagrid_set_option": (_i32, [_vp, C.c_char_p, _i32]),
    "hagrid_traversal_image_info": (_i32, [_vp, C.POINTER(GridPOD), _vp, C.POINTER(_i64)]),
}

# libhagrid_amd_kat.so (hagrid_amd/csrc/kat/hagrid_amd_kat.h): known-answer hooks and diagnostic instantiations -- tests/ and tools/ only
KAT_LIB_PATH = os.path.join(HERE, "libhagrid_amd_kat.so")
KAT_SIGNATURES = {
    "hagrid_kat_intersect_prim_ray": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "hagrid_kat_intersect_prim_ray_uvs": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "hagrid_kat_intersect_prim_cell": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp]),
    "hagrid_kat_compute_range": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp]),
    "hagrid_kat_compute_grid_dims": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp]),
    "hagrid_kat_lookup_entry": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp]),
    "hagrid_kat_scan": (_i32, [_vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp]),
    "hagrid_kat_set_option": (_i32, [_vp, C.c_char_p, _i32]),
    "hagrid_kat_order_state": (_i32, [_vp, _vp, _vp, _vp]),
    "hagrid_kat_forget_hints": (_i32, [_vp]),
    "hagrid_kat_detect_ray_rows": (_i32, [_vp, _vp, _i32, C.c_float, _vp]),
    "hagrid_kat_image_records": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp]),
    "hagrid_kat_tile_slots": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp]),
    "hagrid_kat_traverse_timed": (_i32, [_vp, C.POINTER(GridPOD), _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
}

_lib = None
_kat = None


def _preload_hip_runtime() -> None:
    """Make the process' HIP runtime visible to libhagrid_amd.so (which names none itself)."""
    candidates = []
    try:
        import torch  # noqa: F401  (PyTorch's bundled runtime; must be the one and only in the process)
        candidates.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:
        pass
    candidates += ["/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    last = None
    for c in candidates:
        try:
            C.CDLL(c, mode=C.RTLD_GLOBAL)
            return
        except OSError as e:  # try the next location
            last = e
    raise HagridError(f"no HIP runtime (libamdhip64.so) could be loaded: {last}")


def load() -> C.CDLL:
    """Loads the library (once) and declares every signature.  Raises HagridError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HagridError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or python hagrid_amd/build.py).  There is no CPU fallback.")
    _preload_hip_runtime()
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise HagridError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means header and library disagree
        fn.restype = res
        fn.argtypes = args
    if lib.hagrid_abi_version() != ABI_VERSION:
        raise HagridError("ABI version mismatch")
    _lib = lib
    return lib


def load_kat() -> C.CDLL:
    """The test library (known-answer hooks); needs the product library in the process first.  Tests and dev tools only."""
    global _kat
    if _kat is not None:
        return _kat
    load()
    if os.path.realpath(LIB_PATH) != os.path.realpath(os.path.join(HERE, "libhagrid_amd.so")):
        # The test library links its SIBLING product library (rpath $ORIGIN): with HAGRID_AMD_LIB pointing elsewhere it would pull a second,
        # different copy of the product into the process and hand it contexts created by the first.
        raise HagridError(f"HAGRID_AMD_LIB={LIB_PATH} overrides the product library, but {KAT_LIB_PATH} is built against its sibling "
                          f"{os.path.join(HERE, 'libhagrid_amd.so')}: the test hooks (known-answer tests, code-path selectors) cannot be used in an A/B run")
    if not os.path.exists(KAT_LIB_PATH):
        raise HagridError(f"{KAT_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    try:
        lib = C.CDLL(KAT_LIB_PATH)
    except OSError as e:
        raise HagridError(f"cannot load {KAT_LIB_PATH}: {e}") from e
    for name, (res, args) in KAT_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _kat = lib
    return lib
