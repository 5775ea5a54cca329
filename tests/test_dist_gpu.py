"""The N > 1 flow on one GPU: two ranks (gloo transport, both on cuda:0) run bench.py's multi-rank path --
rank 0 builds, broadcast_grid ships the grid, each rank traverses its shard -- and a direct check that the
received grid equals the built one.  On a multi-GPU node the same code runs with backend nccl (= RCCL)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import _subproc

from hagrid_amd import scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_share_one_gpu():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--tris", "100000", "--width", "512", "--height", "512", "--backend", "gloo", "--device", "0", "--build-iter", "1"]
    r = _subproc.run(cmd, timeout=300, cwd=ROOT)
    assert not r.timed_out, r.diagnosis
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["config"]["rays_rank0"] == 512 * 512 and out["config"]["rays_total"] == 2 * 512 * 512
    assert 0.3 < out["hit_fraction"] <= 1.0


def test_bench_started_bare_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (the shape of the driver's N = 1 command) re-executes itself under
    torch.distributed.run instead of exiting: one JSON line from rank 0, two ranks in it."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR")}
    env["MASTER_PORT"] = str(free_port())
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--tris", "100000", "--width", "256",
           "--height", "256", "--backend", "gloo", "--device", "0", "--build-iter", "1", "--no-cpu-baseline"]
    r = _subproc.run(cmd, timeout=300, cwd=ROOT, env=env)
    assert not r.timed_out, r.diagnosis
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["rays_total"] == 2 * 256 * 256


@pytest.mark.parametrize("config,extra", [(4, ["--total-rays", "300001"]), (3, ["--width", "512", "--height", "256"]),
                                          (5, ["--width", "256", "--height", "256"])])
def test_bench_strong_scaling_two_ranks(config, extra):
    """bench.py --config 3/4/5 shards ONE batch over the ranks (strong scaling): the job total is the batch, not N batches."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", str(config),
           "--tris", "100000", "--backend", "gloo", "--device", "0", "--build-iter", "1"] + extra
    r = _subproc.run(cmd, timeout=300, cwd=ROOT)
    assert not r.timed_out, r.diagnosis
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    total = 300001 if config == 4 else int(extra[1]) * int(extra[3])
    assert out["scaling"] == "strong" and out["n_gpus"] == 2 and out["config"]["rays_total"] == total
    assert out["config"]["rays_rank0"] == total // 2 and out["config"]["baseline_config"] == config
    assert out["config"]["grid"]["compressed"] == (config == 5)
    assert 0.2 < out["hit_fraction"] <= 1.0


def _worker(rank, port, q):
    import torch
    import torch.distributed as dist
    from hagrid_amd import api, dist as hdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        mem = api.MemManager(keep=True, device=0)
        n_tris = 30000
        grid = None; d_tris = 0
        if rank == 0:
            tris = scene.make_soup(n_tris)
            d_tris = mem.upload(tris)
            grid = api.build_all(mem, d_tris, n_tris, compress=True)
        grid, d_tris = hdist.broadcast_grid(mem, grid, d_tris, n_tris, src=0)
        n_rays = 100001
        rays = scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, n_rays, 11)
        b, e = scene.shard_range(n_rays, rank, 2)
        d_rays = mem.upload(rays[b:e]); d_hits = mem.alloc(16 * (e - b))
        api.traverse_grid(grid, d_tris, d_rays, d_hits, e - b)
        hits = mem.download(d_hits, api.HIT_DTYPE, e - b)
        d = grid.download()
        q.put((rank, b, e, hits["id"].copy(), hits["t"].copy(), grid.summary(), int(d["entries"].sum(dtype=np.int64)), int(d["ref_ids"].sum(dtype=np.int64))))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_broadcast_grid_and_sharded_traversal():
    import torch.multiprocessing as mp
    from hagrid_amd import api
    port = free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    assert res[0][5] == res[1][5] and res[0][6] == res[1][6] and res[0][7] == res[1][7]      # same grid on both ranks
    # the union of the shards equals the single-process traversal
    mem = api.MemManager(keep=True)
    tris = scene.make_soup(30000); d_tris = mem.upload(tris)
    grid = api.build_all(mem, d_tris, 30000, compress=True)
    rays = scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 100001, 11)
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * 100001)
    api.traverse_grid(grid, d_tris, d_rays, d_hits, 100001)
    want = mem.download(d_hits, api.HIT_DTYPE, 100001)
    for rank, b, e, hid, ht, *_ in res:
        assert (hid == want["id"][b:e]).all() and (ht.view(np.uint32) == want["t"][b:e].view(np.uint32)).all()
    mem.close()


@pytest.mark.parametrize("compress", [False, True])
def test_grid_blob_pack_unpack_save_load(compress, tmp_path):
    """The C packer writes the bytes of the numpy packer; a blob unpacks IN PLACE into a grid whose arrays are ordinary pool
    buffers (freed one by one, the memory goes with the last); the file form loads into another context; hits never change."""
    import ctypes as C
    from hagrid_amd import api, dist as hdist
    n_tris = 50_000
    tris = scene.make_soup(n_tris)
    mem = api.MemManager(keep=False)
    d_tris = mem.upload(tris)
    grid = api.build_all(mem, d_tris, n_tris, compress=compress)
    rays = scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 100_000, 3)
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * rays.shape[0])

    def hits_of(m, g, t, dr, dh):
        api.setup_traversal(g)
        api.traverse_grid(g, t, dr, dh, rays.shape[0])
        return m.download(dh, api.HIT_DTYPE, rays.shape[0])

    want = hits_of(mem, grid, d_tris, d_rays, d_hits)
    # pack == the host packer on the downloaded arrays
    p = C.c_void_p(); nb = C.c_size_t()
    api._check(mem, mem._L.hagrid_grid_pack(mem._ctx, C.byref(grid.pod), C.c_void_p(d_tris), n_tris, C.byref(p), C.byref(nb)), "pack")
    assert nb.value == mem._L.hagrid_grid_blob_bytes(C.byref(grid.pod), n_tris) and nb.value % 128 == 0
    blob = mem.download(p.value, np.uint8, nb.value)
    d = grid.download()
    host = hdist.pack_blob_host(d["entries"], d["ref_ids"], d["cells"], d["small_cells"], d["bbox_min"], d["bbox_max"], d["dims"], d["shift"], d["offsets"], tris)
    assert blob.tobytes() == host.tobytes()
    # unpack in place: no new memory, four separately freeable arrays
    usage = mem.usage()
    g2 = api.Grid(); g2.mem = mem
    t2 = C.c_void_p(); nt = C.c_int()
    api._check(mem, mem._L.hagrid_grid_unpack(mem._ctx, p, nb.value, C.byref(g2.pod), C.byref(t2), C.byref(nt)), "unpack")
    assert mem.usage() == usage and nt.value == n_tris and g2.summary() == grid.summary()
    with pytest.raises(api.HagridError):
        mem.free(p.value)                                   # the blob pointer is no pool pointer any more
    got = hits_of(mem, g2, t2.value, d_rays, d_hits)
    assert (got["id"] == want["id"]).all() and (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all()
    g2.free()                                               # three of the four parts (and the traversal image derived from them)
    left = mem.usage()
    assert (mem.download(t2.value, np.float32, 12 * n_tris).reshape(-1, 12) == tris).all()      # the triangles of the blob are still alive
    mem.free(t2.value)
    assert mem.usage() == left - (nb.value + 255) // 256 * 256      # keep = False: the slot (256-byte granules) is released with its last part
    # a damaged header is refused and leaves the buffer alone
    api._check(mem, mem._L.hagrid_grid_pack(mem._ctx, C.byref(grid.pod), C.c_void_p(d_tris), n_tris, C.byref(p), C.byref(nb)), "pack")
    mem.zero(p.value + 200, 8)
    rc = mem._L.hagrid_grid_unpack(mem._ctx, p, nb.value, C.byref(g2.pod), C.byref(t2), C.byref(nt))
    assert rc < 0
    mem.free(p.value)
    # the file form, into another context
    path = str(tmp_path / "grid.blob")
    hdist.save_grid(mem, grid, d_tris, n_tris, path)
    assert open(path, "rb").read() == host.tobytes()
    mem2 = api.MemManager(keep=True)
    g3, t3, n3 = hdist.load_grid(mem2, path)
    assert n3 == n_tris and g3.summary() == grid.summary()
    dr = mem2.upload(rays); dh = mem2.alloc(16 * rays.shape[0])
    got = hits_of(mem2, g3, t3, dr, dh)
    assert (got["id"] == want["id"]).all() and (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all()
    open(path, "wb").write(host.tobytes()[:-4096])
    with pytest.raises(api.HagridError):
        hdist.load_grid(mem2, path)
    mem2.close(); mem.close()


def test_bench_rccl_path_with_one_rank():
    """The RCCL code path (process group on backend nccl, barrier, grid broadcast, all-reduce) with world size 1."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--tris", "100000",
           "--width", "512", "--height", "512", "--build-iter", "1", "--force-dist", "--no-cpu-baseline"]
    r = _subproc.run(cmd, timeout=300, cwd=ROOT, env=env)
    assert not r.timed_out, r.diagnosis
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["grid_broadcast_ms"] >= 0


def test_bench_single_gpu_line_carries_the_pipelined_block():
    """The default (single process) bench line: `pipelined` = the same steps with two calls in flight over one shared traversal image,
    hits identical to the single-stream run; never part of `value`.  --inflight 0 leaves it out."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--tris", "100000",
            "--width", "512", "--height", "512", "--build-iter", "1", "--no-cpu-baseline"]
    r = _subproc.run(base, timeout=300, cwd=ROOT)
    assert not r.timed_out, r.diagnosis
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    p = out["pipelined"]
    assert p and p["in_flight"] == 2 and p["hits_identical_to_single_stream"] is True and p["value"] > 0 and p["steps"] >= 6
    assert out["value"] > 0 and out["n_gpus"] == 1
    r = _subproc.run(base + ["--inflight", "0"], timeout=300, cwd=ROOT)
    assert not r.timed_out, r.diagnosis
    assert r.returncode == 0, r.stderr[-3000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["pipelined"] is None


@pytest.mark.parametrize("compress", [False, True])
def test_damaged_grid_file_is_refused_before_it_becomes_a_grid(tmp_path, compress):
    """a grid file is foreign data and the traversal kernels check nothing.  Every index a walk would follow is validated
    when the blob is unpacked (header: top-level cells vs entries, level offsets; device pass: entry targets, cells' reference ranges,
    reference ids, the last sentinel of a compressed grid): a damaged file gives an error with its reason, not a grid, and the pool
    gets the memory back."""
    from hagrid_amd import api, dist as hdist, lib
    n_tris = 20_000
    tris = scene.make_soup(n_tris)
    mem = api.MemManager(keep=False)
    d_tris = mem.upload(tris)
    grid = api.build_all(mem, d_tris, n_tris, compress=compress)
    path = str(tmp_path / "g.blob")
    hdist.save_grid(mem, grid, d_tris, n_tris, path)
    good = bytearray(open(path, "rb").read())
    h = lib.BlobHeader.from_buffer_copy(bytes(good[:256]))
    mem2 = api.MemManager(keep=False)
    base_usage = mem2.usage()

    def refused(blob, why):
        open(path, "wb").write(bytes(blob))
        with pytest.raises(api.HagridError, match=why):
            hdist.load_grid(mem2, path)
        assert mem2.usage() == base_usage, why

    def put32(blob, off, value):
        blob[off:off + 4] = int(value & 0xffffffff).to_bytes(4, "little")

    b = bytearray(good); put32(b, h.off_entries + 4 * 5, (h.num_cells + 7) << 2)                     # a leaf entry beyond the cells
    refused(b, "voxel-map entry")
    b = bytearray(good); put32(b, h.off_entries + 4 * 9, ((h.num_entries - 3) << 2) | 1)            # a node whose eight children run off the end
    refused(b, "voxel-map entry")
    if not compress:
        b = bytearray(good); put32(b, h.off_cells + 32 * 11 + 28, h.num_refs + 1)                   # Cell.end beyond the references
        refused(b, "reference range")
    else:
        b = bytearray(good); put32(b, h.off_cells + 16 * 11 + 12, h.num_refs)                       # SmallCell.begin beyond the references
        refused(b, "reference range")
        b = bytearray(good); put32(b, h.off_refs + 4 * (h.num_refs - 1), 3)                          # the last list loses its sentinel
        refused(b, "names no triangle|no end")
    b = bytearray(good); put32(b, h.off_refs + 4 * 17, n_tris)                                       # a reference to a triangle that is not there
    refused(b, "names no triangle")
    b = bytearray(good); put32(b, 8, h.dims[0] * 40)                                                 # header: more top-level cells than entries
    refused(b, "top-level cells|level offsets")
    b = bytearray(good); put32(b, 48 + 0, h.offsets[0] + 1)                                          # header: offsets[0] != number of top-level cells
    refused(b, "level offsets")
    b = bytearray(good); b[232:240] = (1 << 60).to_bytes(8, "little")                                # header: a size nobody has
    refused(b, "inconsistent section table|no host memory")
    # and the undamaged file still loads
    open(path, "wb").write(bytes(good))
    g2, t2, n2 = hdist.load_grid(mem2, path)
    assert n2 == n_tris and g2.summary() == grid.summary()
    mem2.close(); mem.close()
