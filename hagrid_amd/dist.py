"""Multi-GPU plumbing: one process per GPU, the grid is built once and broadcast, ray batches are sharded.

The reference has no multi-GPU code at all; the path shards naturally because rays are
independent and the grid is read-only during traversal (traverse.cu:35-38).  So there is exactly ONE exchange
step -- the broadcast of the finished grid from the building rank -- and no collective on the data path afterwards:

    rank 0: build_all(...)                       others: wait
    all   : broadcast_grid(...)                  (RCCL over xGMI: backend "nccl" on ROCm; "gloo" in CPU tests)
    all   : traverse_grid on shard_range(num_rays, rank, world)

What travels is the grid BLOB of the C ABI (include/hagrid_amd.h: hagrid_blob_header + entries + cells | small_cells +
ref_ids + triangles in one buffer, also the file form of hagrid_grid_save / hagrid_grid_load): the root packs it with
hagrid_grid_pack (device-to-device), torch.distributed broadcasts the 256-byte header and then the blob STRAIGHT from / into
pool memory (zero-copy tensor views of the pool pointers), and the receivers split the blob in place into the grid's arrays
(hagrid_grid_unpack) -- no staging copies.  A host program without Python uses hagrid_grid_broadcast, the same two
broadcasts on an ncclComm_t (tools/hagrid_cli.cpp --gpus N).

The numpy packer / unpacker below writes the same bytes as the C one (tests/test_dist_gpu.py compares them) and carries the
CPU tests (tests/test_dist_cpu.py: gloo, CPU tensors, the oracle as the traversal).
"""
from __future__ import annotations

import numpy as np

from .scene import CELL_DTYPE, SMALL_CELL_DTYPE, shard_range  # noqa: F401  (re-exported)

BLOB_MAGIC = 0x42524748       # "HGRB"
BLOB_ALIGN = 128
MAX_LEVELS = 32
# struct hagrid_blob_header (include/hagrid_amd.h), 256 bytes
BLOB_HEADER = np.dtype([("magic", "<u4"), ("version", "<u4"), ("dims", "<i4", 3), ("shift", "<i4"),
                        ("num_cells", "<i4"), ("num_entries", "<i4"), ("num_refs", "<i4"), ("num_tris", "<i4"),
                        ("compressed", "<i4"), ("num_offsets", "<i4"), ("offsets", "<i4", MAX_LEVELS),
                        ("bbox_min", "<f4", 3), ("bbox_max", "<f4", 3),
                        ("off_entries", "<u8"), ("off_cells", "<u8"), ("off_refs", "<u8"), ("off_tris", "<u8"), ("total_bytes", "<u8"),
                        ("reserved", "u1", 16)])
assert BLOB_HEADER.itemsize == 256


def _align(n: int) -> int:
    return (max(int(n), 1) + BLOB_ALIGN - 1) // BLOB_ALIGN * BLOB_ALIGN


def make_header(dims, shift, offsets, bbox_min, bbox_max, num_entries, num_cells, num_refs, num_tris, compressed) -> np.ndarray:
    """The blob header for a grid of the given sizes (one-element array of BLOB_HEADER)."""
    h = np.zeros(1, dtype=BLOB_HEADER)
    h["magic"] = BLOB_MAGIC; h["version"] = 1
    h["dims"][0] = dims; h["shift"] = shift
    h["num_cells"] = num_cells; h["num_entries"] = num_entries; h["num_refs"] = num_refs; h["num_tris"] = num_tris
    h["compressed"] = 1 if compressed else 0
    h["num_offsets"] = len(offsets); h["offsets"][0, :len(offsets)] = offsets
    h["bbox_min"][0] = np.asarray(bbox_min, dtype=np.float32); h["bbox_max"][0] = np.asarray(bbox_max, dtype=np.float32)
    at = BLOB_HEADER.itemsize
    for name, nbytes in (("off_entries", 4 * num_entries), ("off_cells", (16 if compressed else 32) * num_cells),
                         ("off_refs", 4 * num_refs), ("off_tris", 48 * num_tris)):
        h[name] = at
        at += _align(nbytes)
    h["total_bytes"] = at
    return h


def parse_header(raw) -> dict:
    """Validates 256 header bytes and returns them as a dict (ValueError on a foreign or inconsistent header)."""
    h = np.frombuffer(np.ascontiguousarray(raw, dtype=np.uint8)[:256].tobytes(), dtype=BLOB_HEADER)[0]
    if int(h["magic"]) != BLOB_MAGIC or int(h["version"]) != 1:
        raise ValueError("bad grid blob header")
    n_off = int(h["num_offsets"])
    if not 0 <= n_off <= MAX_LEVELS:
        raise ValueError("bad grid blob header")
    d = {"dims": tuple(int(v) for v in h["dims"]), "shift": int(h["shift"]), "num_entries": int(h["num_entries"]), "num_cells": int(h["num_cells"]),
         "num_refs": int(h["num_refs"]), "num_tris": int(h["num_tris"]), "compressed": bool(h["compressed"]),
         "bbox_min": h["bbox_min"].astype(np.float32).copy(), "bbox_max": h["bbox_max"].astype(np.float32).copy(),
         "offsets": [int(v) for v in h["offsets"][:n_off]], "total_bytes": int(h["total_bytes"])}
    want = make_header(d["dims"], d["shift"], d["offsets"], d["bbox_min"], d["bbox_max"], d["num_entries"], d["num_cells"], d["num_refs"], d["num_tris"], d["compressed"])[0]
    for k in ("off_entries", "off_cells", "off_refs", "off_tris", "total_bytes"):
        if int(want[k]) != int(h[k]):
            raise ValueError("inconsistent grid blob section table")
        d[k] = int(h[k])
    return d


def pack_blob_host(entries, ref_ids, cells, small_cells, bbox_min, bbox_max, dims, shift, offsets, tris) -> np.ndarray:
    """Host arrays -> blob bytes (uint8), the layout hagrid_grid_pack writes on the device."""
    compressed = small_cells is not None
    c = np.ascontiguousarray(small_cells if compressed else cells)
    entries = np.ascontiguousarray(entries, dtype=np.uint32); ref_ids = np.ascontiguousarray(ref_ids, dtype=np.int32)
    tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 12)
    h = make_header(dims, shift, offsets, bbox_min, bbox_max, entries.size, c.shape[0], ref_ids.size, tris.shape[0], compressed)
    blob = np.zeros(int(h["total_bytes"][0]), dtype=np.uint8)
    blob[:256] = h.view(np.uint8)
    for off, a in ((h["off_entries"], entries), (h["off_cells"], c), (h["off_refs"], ref_ids), (h["off_tris"], tris)):
        raw = a.view(np.uint8).reshape(-1)
        blob[int(off[0]):int(off[0]) + raw.size] = raw
    return blob


def unpack_blob_host(blob: np.ndarray) -> dict:
    """Blob bytes -> header dict + array views (entries, cells | small_cells, ref_ids, tris)."""
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    d = parse_header(blob[:256])
    if d["total_bytes"] > blob.size:
        raise ValueError("truncated grid blob")
    take = lambda off, n, dt: blob[off:off + n * np.dtype(dt).itemsize].view(dt)
    d["entries"] = take(d["off_entries"], d["num_entries"], np.uint32)
    d["ref_ids"] = take(d["off_refs"], d["num_refs"], np.int32)
    cells = take(d["off_cells"], d["num_cells"], SMALL_CELL_DTYPE if d["compressed"] else CELL_DTYPE)
    d["cells"] = None if d["compressed"] else cells
    d["small_cells"] = cells if d["compressed"] else None
    d["tris"] = take(d["off_tris"], d["num_tris"] * 12, np.float32).reshape(-1, 12)
    return d


def broadcast_blob(blob, make_buffer, src: int = 0, group=None):
    """The exchange step, transport-agnostic: `blob` is a 1-D uint8 torch tensor on rank `src` (None elsewhere; the other
    ranks get theirs from make_buffer(nbytes)).  Two broadcasts -- the 256-byte header, then the whole blob: few, large
    messages, which is what a point-to-point xGMI fabric wants.  Returns the blob tensor valid on the calling rank."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    head = blob[:256].clone() if rank == src else make_buffer(256)[:256]
    dist.broadcast(head, src=src, group=group)
    total = parse_header(head.cpu().numpy())["total_bytes"]
    if rank != src:
        blob = make_buffer(total)[:total]
    assert blob.numel() == total and blob.dtype == torch.uint8
    dist.broadcast(blob, src=src, group=group)
    return blob


# ---- GPU side: pool pointers <-> torch tensors ---------------------------------------------------------------

class _PoolView:
    """A pool buffer as a zero-copy torch tensor (the CUDA array interface is all torch needs to wrap device memory)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def pool_tensor(ptr: int, nbytes: int, device):
    import torch
    return torch.as_tensor(_PoolView(ptr, nbytes), device=device)


def broadcast_grid(mem, grid, d_tris: int, num_tris: int, src: int = 0, group=None):
    """Broadcasts rank src's device grid (and triangles) to every rank.  Returns (Grid, d_tris) valid on the
    calling rank; on rank src they are the inputs.  torch.distributed must be initialised (backend nccl = RCCL)."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from . import api
    rank = dist.get_rank(group)
    dev = torch.device("cuda", mem.device)
    blob_ptr = 0
    blob = None
    if rank == src:
        p = C.c_void_p(); n = C.c_size_t()
        api._check(mem, mem._L.hagrid_grid_pack(mem._ctx, C.byref(grid.pod), C.c_void_p(d_tris), int(num_tris), C.byref(p), C.byref(n)), "grid_pack")
        blob_ptr = int(p.value)
        blob = pool_tensor(blob_ptr, n.value, dev)
    received = {}

    def make_buffer(nbytes):           # header scratch from torch, the blob itself from the pool
        if nbytes <= 256:
            return torch.empty(256, dtype=torch.uint8, device=dev)
        received["ptr"] = mem.alloc(nbytes); received["bytes"] = int(nbytes)
        return pool_tensor(received["ptr"], nbytes, dev)

    torch.cuda.synchronize(dev)        # the pack ran on the manager's stream, the collective runs on torch's
    blob = broadcast_blob(blob, make_buffer, src, group)
    torch.cuda.synchronize(dev)
    if rank == src:
        mem.free(blob_ptr)
        return grid, d_tris
    g = api.Grid(); g.mem = mem
    t = C.c_void_p(); nt = C.c_int()
    api._check(mem, mem._L.hagrid_grid_unpack(mem._ctx, C.c_void_p(received["ptr"]), received["bytes"], C.byref(g.pod), C.byref(t), C.byref(nt)), "grid_unpack")
    return g, int(t.value)


def save_grid(mem, grid, d_tris: int, num_tris: int, path: str):
    """The blob as a file (hagrid_grid_save)."""
    import ctypes as C
    from . import api
    api._check(mem, mem._L.hagrid_grid_save(mem._ctx, C.byref(grid.pod), C.c_void_p(d_tris), int(num_tris), path.encode()), "grid_save")


def load_grid(mem, path: str):
    """-> (Grid, d_tris, num_tris) from a file written by save_grid / hagrid_grid_save."""
    import ctypes as C
    from . import api
    g = api.Grid(); g.mem = mem
    t = C.c_void_p(); nt = C.c_int()
    api._check(mem, mem._L.hagrid_grid_load(mem._ctx, path.encode(), C.byref(g.pod), C.byref(t), C.byref(nt)), "grid_load")
    return g, int(t.value), int(nt.value)
