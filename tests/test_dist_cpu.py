"""The N > 1 path on CPU: world_size-2 gloo processes broadcast an (oracle-built) grid with the same
exchange code the GPU path uses, traverse their ray shards with the oracle, and the union equals the
single-process result.  No GPU."""
import os
import socket

import numpy as np
import pytest

from hagrid_amd import dist as hdist
from hagrid_amd import scene


def test_header_roundtrip():
    h = hdist.pack_header((50, 52, 54), 2, [10, 20, 30], np.float32([-0.5, 0.25, 1e-3]), np.float32([1.5, 2.0, 3.0]), 30, 17, 99, 12, True)
    d = hdist.unpack_header(h)
    assert d["dims"] == (50, 52, 54) and d["shift"] == 2 and d["offsets"] == [10, 20, 30] and d["compressed"]
    assert (d["bbox_min"] == np.float32([-0.5, 0.25, 1e-3])).all() and (d["bbox_max"] == np.float32([1.5, 2.0, 3.0])).all()
    assert hdist.array_nbytes(d) == {"entries": 120, "cells": 16 * 17, "ref_ids": 396, "tris": 48 * 12}
    with pytest.raises(ValueError):
        hdist.unpack_header(np.zeros(64, dtype=np.int64))


def test_shard_ranges_partition_the_batch():
    for n in (0, 1, 7, 1 << 20, 1000003):
        for w in (1, 2, 3, 8):
            r = [scene.shard_range(n, i, w) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1


def _worker(rank, world, port, compress, q):
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tris_full = scene.make_soup(4000)
        header = arrays = None
        if rank == 0:       # only the building rank has the scene and the grid
            G = O.Grid.full(tris_full, compress=compress)
            cells = G.small_cells if compress else G.cells
            header = hdist.pack_header(G.dims, G.shift, G.offsets, G.bbox_min, G.bbox_max, G.num_entries, G.num_cells, G.num_refs, tris_full.shape[0], compress)
            as_u8 = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy())
            arrays = {"entries": as_u8(G.entries), "cells": as_u8(cells), "ref_ids": as_u8(G.ref_ids), "tris": as_u8(tris_full)}
        hd, out = hdist.broadcast_payload(header, arrays, lambda n: torch.empty(int(n), dtype=torch.uint8), src=0)
        ent = out["entries"].numpy().view(np.uint32); refs = out["ref_ids"].numpy().view(np.int32)
        tris = out["tris"].numpy().view(np.float32).reshape(-1, 12)
        cells = out["cells"].numpy().view(O.SMALL_CELL_DTYPE if hd["compressed"] else O.CELL_DTYPE)
        G2 = O.Grid.from_arrays(ent, refs, None if hd["compressed"] else cells, cells if hd["compressed"] else None,
                                hd["bbox_min"], hd["bbox_max"], hd["dims"], hd["shift"], hd["offsets"])
        n_rays = 20001
        rays = scene.make_rays_incoherent(hd["bbox_min"], hd["bbox_max"], n_rays, 77)
        b, e = scene.shard_range(n_rays, rank, world)
        hits, _ = G2.traverse(tris, rays[b:e])
        q.put((rank, b, e, hits["id"].copy(), hits["t"].copy(), bool((tris == tris_full).all())))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("compress", [False, True])
def test_two_rank_broadcast_and_sharded_traversal(compress):
    import torch.multiprocessing as mp
    from oracle import oracle as O
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, compress, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    tris = scene.make_soup(4000)
    G = O.Grid.full(tris, compress=compress)
    rays = scene.make_rays_incoherent(G.bbox_min, G.bbox_max, 20001, 77)
    want, _ = G.traverse(tris, rays)
    got_id = np.full(20001, -9, dtype=np.int32); got_t = np.zeros(20001, dtype=np.float32)
    for rank, b, e, hid, ht, tris_ok in res:
        assert tris_ok
        got_id[b:e] = hid; got_t[b:e] = ht
    assert (got_id == want["id"]).all() and (got_t.view(np.uint32) == want["t"].view(np.uint32)).all()
