"""The N > 1 path on CPU: world_size-2 gloo processes broadcast an (oracle-built) grid with the same
exchange code the GPU path uses, traverse their ray shards with the oracle, and the union equals the
single-process result.  No GPU."""
import os
import socket

import numpy as np
import pytest

from hagrid_amd import dist as hdist
from hagrid_amd import scene


def test_blob_header_and_host_roundtrip():
    """The numpy packer writes the layout of struct hagrid_blob_header (include/hagrid_amd.h) and reads it back."""
    import ctypes as C
    from hagrid_amd import lib
    assert C.sizeof(lib.BlobHeader) == hdist.BLOB_HEADER.itemsize == 256
    for name, _ in lib.BlobHeader._fields_:
        assert getattr(lib.BlobHeader, name).offset == hdist.BLOB_HEADER.fields[name][1], name
    rng = np.random.default_rng(3)
    for compressed in (False, True):
        entries = rng.integers(0, 1 << 30, 30, dtype=np.uint32); refs = rng.integers(0, 12, 99).astype(np.int32)
        cells = np.zeros(17, dtype=hdist.SMALL_CELL_DTYPE if compressed else hdist.CELL_DTYPE)
        cells["begin"] = np.arange(17)
        tris = rng.normal(size=(12, 12)).astype(np.float32)
        blob = hdist.pack_blob_host(entries, refs, None if compressed else cells, cells if compressed else None,
                                    np.float32([-0.5, 0.25, 1e-3]), np.float32([1.5, 2.0, 3.0]), (50, 52, 54), 2, [10, 20, 30], tris)
        assert blob.size % 128 == 0
        d = hdist.unpack_blob_host(blob)
        assert d["dims"] == (50, 52, 54) and d["shift"] == 2 and d["offsets"] == [10, 20, 30] and d["compressed"] == compressed
        assert (d["bbox_min"] == np.float32([-0.5, 0.25, 1e-3])).all() and (d["bbox_max"] == np.float32([1.5, 2.0, 3.0])).all()
        assert (d["entries"] == entries).all() and (d["ref_ids"] == refs).all() and (d["tris"] == tris).all()
        got = d["small_cells"] if compressed else d["cells"]
        assert got.tobytes() == cells.tobytes() and (d["cells"] is None) == compressed
        assert all(d[k] % 128 == 0 for k in ("off_entries", "off_cells", "off_refs", "off_tris"))
    with pytest.raises(ValueError):
        hdist.parse_header(np.zeros(256, dtype=np.uint8))
    bad = blob.copy(); bad[200:208] = 255                      # off_entries
    with pytest.raises(ValueError):
        hdist.parse_header(bad[:256])
    with pytest.raises(ValueError):
        hdist.unpack_blob_host(blob[:-128])


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
        blob = None
        if rank == 0:       # only the building rank has the scene and the grid
            G = O.Grid.full(tris_full, compress=compress)
            blob = torch.from_numpy(hdist.pack_blob_host(G.entries, G.ref_ids, G.cells, G.small_cells, G.bbox_min, G.bbox_max, G.dims, G.shift, G.offsets, tris_full))
        blob = hdist.broadcast_blob(blob, lambda n: torch.empty(int(n), dtype=torch.uint8), src=0)
        hd = hdist.unpack_blob_host(blob.numpy())
        tris = hd["tris"]
        G2 = O.Grid.from_arrays(hd["entries"], hd["ref_ids"], hd["cells"], hd["small_cells"], hd["bbox_min"], hd["bbox_max"], hd["dims"], hd["shift"], hd["offsets"])
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
