"""Multi-GPU plumbing: one process per GPU, the grid is built once and broadcast, ray batches are sharded.

The reference has no multi-GPU code at all; the path shards naturally because rays are
independent and the grid is read-only during traversal (traverse.cu:35-38).  So there is exactly ONE exchange
step -- the broadcast of the finished grid (header + entries + cells|small_cells + ref_ids + triangles) from
the building rank -- and no collective on the data path afterwards:

    rank 0: build_all(...)                       others: wait
    all   : broadcast_grid(...)                  (RCCL over xGMI: backend "nccl" on ROCm; "gloo" in CPU tests)
    all   : traverse_grid on shard_range(num_rays, rank, world)

The broadcast works on torch tensors (torch.distributed is the transport); the tensors are filled from /
copied into the C ABI's buffer pool with device-to-device copies.  The same code runs on CPU tensors under
gloo (tests/test_dist_cpu.py), where the "device" arrays are numpy-backed.
"""
from __future__ import annotations

import numpy as np

from .scene import CELL_DTYPE, SMALL_CELL_DTYPE, shard_range  # noqa: F401  (re-exported)

HEADER_WORDS = 64
_MAGIC = 0x48414752   # 'HAGR'


def pack_header(dims, shift, offsets, bbox_min, bbox_max, num_entries, num_cells, num_refs, num_tris, compressed) -> np.ndarray:
    """Grid descriptor as 64 int64 words (floats bit-cast), so one small broadcast announces all sizes."""
    h = np.zeros(HEADER_WORDS, dtype=np.int64)
    h[0] = _MAGIC; h[1] = 1
    h[2:5] = dims; h[5] = shift
    h[6] = num_entries; h[7] = num_cells; h[8] = num_refs; h[9] = num_tris; h[10] = 1 if compressed else 0
    h[11:14] = np.asarray(bbox_min, dtype=np.float32).view(np.int32)
    h[14:17] = np.asarray(bbox_max, dtype=np.float32).view(np.int32)
    h[17] = len(offsets)
    h[18:18 + len(offsets)] = offsets
    return h


def unpack_header(h: np.ndarray) -> dict:
    h = np.asarray(h, dtype=np.int64)
    if h[0] != _MAGIC or h[1] != 1:
        raise ValueError("bad grid header")
    n_off = int(h[17])
    return {"dims": tuple(int(v) for v in h[2:5]), "shift": int(h[5]), "num_entries": int(h[6]), "num_cells": int(h[7]),
            "num_refs": int(h[8]), "num_tris": int(h[9]), "compressed": bool(h[10]),
            "bbox_min": h[11:14].astype(np.int32).view(np.float32).copy(), "bbox_max": h[14:17].astype(np.int32).view(np.float32).copy(),
            "offsets": [int(v) for v in h[18:18 + n_off]]}


def array_nbytes(hd: dict) -> dict:
    """Byte sizes of the four broadcast payloads."""
    cell_bytes = 16 if hd["compressed"] else 32
    return {"entries": 4 * hd["num_entries"], "cells": cell_bytes * hd["num_cells"], "ref_ids": 4 * hd["num_refs"], "tris": 48 * hd["num_tris"]}


def broadcast_payload(header: np.ndarray | None, arrays: dict | None, make_buffer, src: int = 0, group=None):
    """Core exchange step, transport-agnostic.

    header / arrays are given on rank `src` (arrays: name -> 1-D uint8 torch tensor on the transport's device);
    other ranks pass None and get buffers from make_buffer(nbytes).  Returns (header dict, arrays dict).
    Five broadcasts: one 512-byte header, four payloads -- few, large messages, which is what a point-to-point
    xGMI fabric wants."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    dev = make_buffer(8).device
    ht = torch.zeros(HEADER_WORDS, dtype=torch.int64, device=dev)
    if rank == src:
        ht.copy_(torch.from_numpy(header))
    dist.broadcast(ht, src=src, group=group)
    hd = unpack_header(ht.cpu().numpy())
    sizes = array_nbytes(hd)
    out = {}
    for name in ("entries", "cells", "ref_ids", "tris"):
        if rank == src:
            buf = arrays[name]
            assert buf.numel() == sizes[name], (name, buf.numel(), sizes[name])
        else:
            buf = make_buffer(max(sizes[name], 1))[:sizes[name]]
        if sizes[name]:
            dist.broadcast(buf, src=src, group=group)
        out[name] = buf
    return hd, out


# ---- GPU side: pool pointers <-> torch tensors ---------------------------------------------------------------

def broadcast_grid(mem, grid, d_tris: int, num_tris: int, src: int = 0, group=None):
    """Broadcasts rank src's device grid (and triangles) to every rank.  Returns (Grid, d_tris) valid on the
    calling rank; on rank src they are the inputs.  Needs torch.distributed initialised with backend nccl."""
    import torch
    import torch.distributed as dist
    from . import api
    rank = dist.get_rank(group)
    dev = torch.device("cuda", mem.device)

    def make_buffer(nbytes):
        return torch.empty(int(nbytes), dtype=torch.uint8, device=dev)

    header = arrays = None
    if rank == src:
        compressed = bool(grid.small_cells)
        header = pack_header(grid.dims, grid.shift, grid.offsets, grid.bbox_min, grid.bbox_max, grid.num_entries, grid.num_cells,
                             grid.num_refs, num_tris, compressed)
        sizes = array_nbytes(unpack_header(header))
        ptrs = {"entries": grid.entries, "cells": grid.small_cells if compressed else grid.cells, "ref_ids": grid.ref_ids, "tris": d_tris}
        arrays = {}
        for name, p in ptrs.items():
            t = make_buffer(max(sizes[name], 1))[:sizes[name]]
            if sizes[name]:
                mem.copy_d2d(t.data_ptr(), p, sizes[name])     # same (null) stream as torch's default stream
            arrays[name] = t
        torch.cuda.synchronize(dev)
    hd, out = broadcast_payload(header, arrays, make_buffer, src, group)
    torch.cuda.synchronize(dev)
    if rank == src:
        return grid, d_tris
    g = api.Grid(); g.mem = mem
    sizes = array_nbytes(hd)
    dst = {}
    for name in ("entries", "cells", "ref_ids", "tris"):
        dst[name] = mem.alloc(max(sizes[name], 4))
        if sizes[name]:
            mem.copy_d2d(dst[name], out[name].data_ptr(), sizes[name])
    torch.cuda.synchronize(dev)
    g.pod.entries = dst["entries"]; g.pod.ref_ids = dst["ref_ids"]
    if hd["compressed"]:
        g.pod.small_cells = dst["cells"]
    else:
        g.pod.cells = dst["cells"]
    for i in range(3):
        g.pod.bbox_min[i] = float(hd["bbox_min"][i]); g.pod.bbox_max[i] = float(hd["bbox_max"][i]); g.pod.dims[i] = hd["dims"][i]
    g.pod.num_cells = hd["num_cells"]; g.pod.num_entries = hd["num_entries"]; g.pod.num_refs = hd["num_refs"]
    g.pod.shift = hd["shift"]; g.pod.num_offsets = len(hd["offsets"])
    for i, o in enumerate(hd["offsets"]):
        g.pod.offsets[i] = o
    return g, dst["tris"]
