"""GPU parity of the traversal kernels against the CPU oracle (bit-exact ids AND t), through the C ABI."""
import ctypes as C
import os

import numpy as np
import pytest

from hagrid_amd import scene

pytestmark = pytest.mark.gpu


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def mem():
    from hagrid_amd import api
    m = api.MemManager(keep=True)
    yield m
    m.close()


def upload_oracle_grid(mem, G):
    from hagrid_amd import api
    return api.Grid.upload(mem, G.entries, G.ref_ids, G.cells, G.small_cells, G.bbox_min, G.bbox_max, G.dims, G.shift, G.offsets)


def gpu_traverse(mem, grid, d_tris, rays, stats=False):
    from hagrid_amd import api
    n = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(max(16 * n, 16))
    api.setup_traversal(grid)
    st = None
    if stats:
        d_steps = mem.alloc(max(4 * n, 4))
        st = api.traverse_grid_stats(grid, d_tris, d_rays, d_hits, n, d_steps)
        steps = mem.download(d_steps, np.int32, n); mem.free(d_steps)
    else:
        api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    hits = mem.download(d_hits, api.HIT_DTYPE, n)
    mem.free(d_rays); mem.free(d_hits)
    return (hits, st, steps) if stats else hits


def test_l0_device_functions_match_golden(mem, golden_dir):
    """The DEVICE versions of the L0 functions against the reference-header golden vectors."""
    kat = np.load(os.path.join(golden_dir, "l0_kat.npz"))
    L, ctx = mem._K, mem._ctx
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    tris = np.ascontiguousarray(kat["tris"])
    n = kat["ipr_rays"].shape[0]
    ret = np.zeros(n, np.int32); hid = np.zeros(n, np.int32); ht = np.zeros(n, np.float32)
    tid = np.ascontiguousarray(kat["ipr_tid"]); rays = np.ascontiguousarray(kat["ipr_rays"])
    assert L.hagrid_kat_intersect_prim_ray(ctx, p(tris), p(rays), p(tid), n, p(ret), p(hid), p(ht)) == 0
    assert (ret == kat["ipr_ret"]).all() and (hid == kat["ipr_hit_id"]).all() and (bits(ht) == bits(kat["ipr_hit_t"])).all()
    n = kat["ipc_boxes"].shape[0]
    ret = np.zeros(n, np.int32); tid = np.ascontiguousarray(kat["ipc_tid"]); boxes = np.ascontiguousarray(kat["ipc_boxes"])
    assert L.hagrid_kat_intersect_prim_cell(ctx, p(tris), p(boxes), p(tid), n, p(ret)) == 0
    assert (ret == kat["ipc_ret"]).all()
    n = kat["range_dims"].shape[0]; out = np.zeros((n, 6), np.int32)
    a, b, c = (np.ascontiguousarray(kat[k]) for k in ("range_dims", "range_grid_bb", "range_obj_bb"))
    assert L.hagrid_kat_compute_range(ctx, p(a), p(b), p(c), n, p(out)) == 0
    assert (out == kat["range_out"]).all()
    n = kat["gd_bb"].shape[0]; out = np.zeros((n, 3), np.int32)
    a, b, c = (np.ascontiguousarray(kat[k]) for k in ("gd_bb", "gd_nprims", "gd_density"))
    assert L.hagrid_kat_compute_grid_dims(ctx, p(a), p(b), p(c), n, p(out)) == 0
    assert (out == kat["gd_out"]).all()
    for tag in ("octree", "flat"):
        ent = np.ascontiguousarray(kat[f"lk_{tag}_entries"]); vox = np.ascontiguousarray(kat[f"lk_{tag}_voxels"])
        td = np.ascontiguousarray(kat[f"lk_{tag}_dims"]); out = np.zeros(vox.shape[0], np.uint32)
        assert L.hagrid_kat_lookup_entry(ctx, p(ent), ent.shape[0], int(kat[f"lk_{tag}_shift"]), p(td), p(vox), vox.shape[0], p(out)) == 0
        assert (out == kat[f"lk_{tag}_out"]).all()


@pytest.mark.parametrize("stage", ["build", "merge", "flatten", "expand", "compress"])
def test_traverse_config1_every_stage(mem, golden_dir, stage):
    """BASELINE config 1 (soup-10k, 64k incoherent rays): GPU hits == oracle hits == reference brute force."""
    from oracle import oracle as O
    g = np.load(os.path.join(golden_dir, "config1_hits.npz"))
    tris = scene.make_soup(10000)
    lo, hi = scene.tris_bbox(tris)
    rays = scene.make_rays_incoherent(lo, hi, 65536, scene.RAY_SEED_BASE + 1)
    G = O.Grid.build(tris)
    order = ["build", "merge", "flatten", "expand", "compress"]
    for s in order[1:order.index(stage) + 1]:
        {"merge": lambda: G.merge(0.995), "flatten": G.flatten, "expand": lambda: G.expand(tris, 3), "compress": G.compress}[s]()
    d_tris = mem.upload(tris)
    grid = upload_oracle_grid(mem, G)
    hits, st, steps = gpu_traverse(mem, grid, d_tris, rays, stats=True)
    oh, ost, osteps = G.traverse(tris, rays, want_steps=True)
    assert (hits["id"] == oh["id"]).all() and (bits(hits["t"]) == bits(oh["t"])).all()
    assert (hits["id"] == g["id"]).all() and (bits(hits["t"]) == bits(g["t"])).all()
    assert (hits["u"] == 0).all() and (hits["v"] == 0).all()
    assert st == ost, (st, ost)
    assert (steps == osteps).all()
    # the plain (non-stats) kernel gives the same hits
    h2 = gpu_traverse(mem, grid, d_tris, rays)
    assert (h2["id"] == hits["id"]).all() and (bits(h2["t"]) == bits(hits["t"])).all()
    grid.free(); mem.free(d_tris)


def test_traverse_primary_rays_and_edge_cases(mem):
    from oracle import oracle as O
    from hagrid_amd import api
    tris = scene.make_soup(30000)
    G = O.Grid.full(tris)
    rays = scene.make_rays_primary(G.bbox_min, G.bbox_max, 256, 256)
    assert (rays[:, 4] == 0).any()
    # rays that miss the grid, rays starting inside, tmin/tmax windows, a degenerate zero direction
    extra = scene.make_rays_incoherent(G.bbox_min - 2, G.bbox_max + 2, 8192, 5)
    extra[:2048, 3] = 0.3; extra[1024:4096, 7] = 0.6
    extra[0, 4:7] = 0.0
    rays = np.concatenate([rays, extra]).astype(np.float32)
    d_tris = mem.upload(tris)
    for compressed in (False, True):
        if compressed:
            assert G.compress()
        grid = upload_oracle_grid(mem, G)
        hits = gpu_traverse(mem, grid, d_tris, rays)
        oh, _ = G.traverse(tris, rays, nthreads=8)
        assert (hits["id"] == oh["id"]).all() and (bits(hits["t"]) == bits(oh["t"])).all()
        # zero rays: a no-op
        api.traverse_grid(grid, d_tris, 0, 0, 0)
        grid.free()
    mem.free(d_tris)


def test_traverse_rejects_incomplete_grid(mem):
    from hagrid_amd import api
    g = api.Grid(); g.mem = mem
    with pytest.raises(api.HagridError):
        api.traverse_grid(g, 0, 0, 0, 16)


def test_mem_manager_contract(mem):
    from hagrid_amd import api
    m = api.MemManager(keep=False)
    a = m.alloc(1000); b = m.alloc(1 << 20)
    assert m.usage() >= 1000 + (1 << 20) and m.max_usage() >= m.usage()
    x = np.arange(250, dtype=np.int32)
    m.copy_h2d(a, x)
    assert (m.download(a, np.int32, 250) == x).all()
    m.one(a, 1000); assert (m.download(a, np.int32, 250) == -1).all()
    m.zero(a, 1000); assert (m.download(a, np.int32, 250) == 0).all()
    m.free(a); m.free(b); m.free(None)
    assert m.usage() == 0
    with pytest.raises(api.HagridError):
        m.free(12345678)
    peak = m.max_usage()
    assert peak >= (1 << 20)
    ms = api.profile(lambda: None, m)
    assert ms >= 0
    m.close()


def test_ray_binning_gives_identical_hits(mem):
    """Extension hagrid_set_ray_binning: same hits in the same slots, for unordered and for ordered batches."""
    from oracle import oracle as O
    from hagrid_amd import api
    tris = scene.make_soup(60000)
    G = O.Grid.full(tris)
    d_tris = mem.upload(tris)
    grid = upload_oracle_grid(mem, G)
    batches = [scene.make_rays_incoherent(G.bbox_min - 0.3, G.bbox_max + 0.3, 300001, 21),      # some start outside / miss
               scene.make_rays_primary(G.bbox_min, G.bbox_max, 512, 384),                          # one shared origin
               scene.make_rays_incoherent(G.bbox_min, G.bbox_max, 5000, 22)]                       # below one tile: not binned
    batches[0][:100, 4:7] = 0.0                                                                     # degenerate directions
    try:
        for rays in batches:
            mem.set_ray_binning(0)
            plain = gpu_traverse(mem, grid, d_tris, rays)
            mem.set_ray_binning(1)
            binned = gpu_traverse(mem, grid, d_tris, rays)
            assert (plain["id"] == binned["id"]).all() and (bits(plain["t"]) == bits(binned["t"])).all()
        oh, _ = G.traverse(tris, batches[0], nthreads=8)
        assert (binned["id"] == plain["id"]).all()
        mem.set_ray_binning(1)
        b0 = gpu_traverse(mem, grid, d_tris, batches[0])
        assert (b0["id"] == oh["id"]).all() and (bits(b0["t"]) == bits(oh["t"])).all()
    finally:
        mem.set_ray_binning(0)
    with pytest.raises(api.HagridError):
        mem.set_ray_binning(7)
    grid.free(); mem.free(d_tris)


@pytest.mark.parametrize("compressed", [False, True])
def test_every_traversal_kernel_gives_the_oracle_hits(mem, compressed):
    """The kernels that walk the construction format (the reference-shaped one and the latency-oriented v2, with and without 32-bit
    addressing), forced through hagrid_set_option: each must reproduce the oracle bit for bit -- including batches that are not a
    multiple of the wavefront size.  (The traversal-image kernels: the tests further down.)"""
    from oracle import oracle as O
    from hagrid_amd import api
    tris = scene.make_soup(80000)
    G = O.Grid.full(tris, compress=compressed)
    d_tris = mem.upload(tris)
    grid = upload_oracle_grid(mem, G)
    rays = np.concatenate([scene.make_rays_primary(G.bbox_min, G.bbox_max, 640, 480),
                           scene.make_rays_incoherent(G.bbox_min - 0.2, G.bbox_max + 0.2, 200003, 31)]).astype(np.float32)
    want, _ = G.traverse(tris, rays, nthreads=8)
    settings = [{"traverse.variant": 1}, {"traverse.variant": 2}, {"traverse.variant": 2, "traverse.narrow": 0}]
    defaults = {"traverse.variant": 0, "traverse.narrow": 1}
    try:
        for st in settings:
            for k, v in {**defaults, **st}.items():
                mem.set_option(k, v)
            for n in (rays.shape[0], 64, 63, 1, 4097):
                got = gpu_traverse(mem, grid, d_tris, rays[:n])
                assert (got["id"] == want["id"][:n]).all() and (bits(got["t"]) == bits(want["t"][:n])).all(), (st, n)
    finally:
        for k, v in defaults.items():
            mem.set_option(k, v)
    for bad in (9, 3):                                       # (3 was the persistent kernel of rounds 1-2: gone)
        with pytest.raises(api.HagridError):
            mem.set_option("traverse.variant", bad)
    with pytest.raises(api.HagridError):
        mem.set_option("no.such.key", 1)
    for gone in ("traverse.refill", "traverse.tri_pad", "traverse.order_moving", "traverse.bin_bits"):          # (round 6: measured, below the bar, removed)
        with pytest.raises(api.HagridError):
            mem.set_option(gone, 1)
    grid.free(); mem.free(d_tris)


def _tile_slots(mem, n, w, super_log2=5, chunked=6):
    """slot of every lane, blocks in dispatch order"""
    out = np.full(64 * ((n + 63) // 64), -1, np.int32)
    assert mem._K.hagrid_kat_tile_slots(mem._ctx, n, w, super_log2, chunked, out.ctypes.data_as(C.c_void_p)) == 0
    return out


@pytest.mark.parametrize("n,w", [(1024 * 1024, 1024), (640 * 480, 640), (200 * 77 + 13, 200), (256 * 256, 256), (2048 * 24, 2048),
                                 (72 * 72, 72), (4096 * 520, 4096), (1000, 0), (1000, 12), (64 * 9, 64), (4097, 8)])
def test_tile_packets_assign_every_ray_to_exactly_one_lane(mem, n, w):
    """The lane <-> ray assignment of the tile packets is a bijection for every batch size / row length / super-tile
    size / XCD order (values >= n are idle lanes), and where it applies a wavefront really holds an 8 x 8 pixel tile."""
    tiled = w >= 8 and w % 8 == 0 and (n // w) >= 8
    for sl in (0, 2, 5, 3 | (11 << 8), 2 | (3 << 8), 3 | (1000 << 8)):            # (bits 8..: rows of super-tiles per band, 0 = one)
        for chunked in (-1, 0, 3, 10):
            slots = _tile_slots(mem, n, w, sl, chunked)
            live = slots[slots < n]
            assert live.min() >= 0 and live.size == n and (np.sort(live) == np.arange(n)).all(), (n, w, sl, chunked)
            t = slots.reshape(-1, 64)
            x, y = (t % w, t // w) if tiled else (t, t)
            is_tile = (x.max(axis=1) - x.min(axis=1) == 7) & (y.max(axis=1) - y.min(axis=1) == 7)
            is_strip = (t == t[:, :1] + np.arange(64)).all(axis=1)
            tiles = (w // 8) * ((n // w) // 8) if tiled else 0
            assert is_tile.sum() >= tiles and (is_tile | is_strip).all() and is_strip.sum() >= t.shape[0] - tiles


def test_row_length_detection(mem):
    lo, hi = np.zeros(3, np.float32), np.ones(3, np.float32)
    def detect(rays, diag=1.7320508):
        d = mem.upload(np.ascontiguousarray(rays, np.float32)); w = C.c_int32(-1)
        assert mem._K.hagrid_kat_detect_ray_rows(mem._ctx, C.c_void_p(d), rays.shape[0], C.c_float(diag), C.byref(w)) == 0
        mem.free(d)
        return w.value
    assert detect(scene.make_rays_primary(lo, hi, 1024, 512)) == 1024
    assert detect(scene.make_rays_primary(lo, hi, 640, 480)) == 640
    assert detect(scene.make_rays_primary(lo, hi, 4096, 16)) == 4096
    assert detect(scene.make_rays_primary(lo, hi, 1000, 100, sample=3, num_samples=8)) == 1000
    assert detect(scene.make_rays_primary(lo, hi, 1001, 100)) == 0                      # not a multiple of 8
    assert detect(scene.make_rays_primary(lo, hi, 512, 4)) == 0                         # fewer than 8 rows
    assert detect(scene.make_rays_primary(lo, hi, 64, 64)[:100]) == 0                   # tiny batch
    assert detect(scene.make_rays_incoherent(lo, hi, 100000, 5)) == 0
    prim = scene.make_rays_primary(lo, hi, 256, 256)
    assert detect(np.concatenate([prim, scene.make_rays_incoherent(lo, hi, 1000, 6)])) == 256   # ragged tail is fine
    same = np.repeat(prim[:1], 4096, axis=0)
    assert detect(same) == 0                                                            # no step at all
    # orthographic camera: one direction, the origin steps across the image plane
    ys, xs = np.divmod(np.arange(320 * 200), 320)
    ortho = np.zeros((320 * 200, 8), np.float32)
    ortho[:, 0] = xs / 320.0; ortho[:, 1] = ys / 200.0; ortho[:, 2] = -1.0; ortho[:, 6] = 1.0; ortho[:, 7] = 10.0
    assert detect(ortho) == 320
    nan = prim.copy(); nan[1, 4] = np.nan
    assert detect(nan) == 0
    # second criterion: image-ordered ORIGINS with unrelated directions (bounce rays leaving the primary hit points)
    from oracle import oracle as O
    tris = scene.make_soup(20000, seed=41)
    G = O.Grid.full(tris)
    diag = float(np.linalg.norm(G.bbox_max - G.bbox_min))
    for w, h in ((512, 256), (640, 480), (800, 400)):
        p = scene.make_rays_primary(G.bbox_min, G.bbox_max, w, h)
        hp, _ = G.traverse(tris, p, nthreads=8)
        b = scene.make_rays_bounce(tris, p, hp, G.bbox_min, G.bbox_max, 99)
        assert detect(b, diag) == w, (w, h)
        assert detect(b[: 40000], diag) == 0                                   # below 64k rays the origin criterion is not run
    assert detect(scene.make_rays_incoherent(G.bbox_min, G.bbox_max, 200000, 8), diag) == 0
    assert detect(scene.make_rays_primary(G.bbox_min, G.bbox_max, 1024, 256), diag) == 1024     # the first criterion still decides


def test_tile_packets_give_identical_hits(mem):
    """Image-ordered batches through v2 with the row length detected on the device, given by the caller (right, wrong and
    useless values) and switched off: always the oracle's hits, in the rays' own slots."""
    from oracle import oracle as O
    tris = scene.make_soup(60000)
    G = O.Grid.full(tris)
    d_tris = mem.upload(tris)
    grid = upload_oracle_grid(mem, G)
    batches = {"640x480": scene.make_rays_primary(G.bbox_min, G.bbox_max, 640, 480),
               "200x77+13": np.concatenate([scene.make_rays_primary(G.bbox_min, G.bbox_max, 200, 77), scene.make_rays_incoherent(G.bbox_min, G.bbox_max, 13, 3)]),
               "incoherent": scene.make_rays_incoherent(G.bbox_min - 0.1, G.bbox_max + 0.1, 70001, 9)}
    try:
        for name, rays in batches.items():
            rays = np.ascontiguousarray(rays, np.float32)
            want, _ = G.traverse(tris, rays, nthreads=8)
            for width in (0, -1, 640, 200, 8, 24, 1 << 20, 333):
                for sl in (5, 1):
                    mem.set_option("traverse.image_width", width); mem.set_option("traverse.super_tile", sl)
                    got = gpu_traverse(mem, grid, d_tris, rays)
                    assert (got["id"] == want["id"]).all() and (bits(got["t"]) == bits(want["t"])).all(), (name, width, sl)
    finally:
        mem.set_option("traverse.image_width", 0); mem.set_option("traverse.super_tile", 3)
    grid.free(); mem.free(d_tris)


# ---- traversal image ----------------------------------------------------------------------------------------------------

O_CELL_DTYPE = np.dtype([("min", "<i4", 3), ("begin", "<i4"), ("max", "<i4", 3), ("end", "<i4")])


def _np_lookup(G, vox):
    """lookup_entry (grid.h:103-116) for many voxels at once: cell index per voxel"""
    e = G.entries; shift = G.shift; dx, dy, dz = G.dims
    top = vox >> shift
    w = e[top[:, 0] + dx * (top[:, 1] + dy * top[:, 2])].astype(np.int64)
    depth = np.zeros(len(vox), np.int64)
    while True:
        k = w & 3
        live = k != 0
        if not live.any():
            break
        depth = depth + k
        s = shift - depth
        m = (1 << k) - 1
        sub = ((vox[:, 0] >> s) & m) + ((((vox[:, 1] >> s) & m) + (((vox[:, 2] >> s) & m) << k)) << k)
        nxt = e[np.where(live, (w >> 2) + sub, 0)].astype(np.int64)
        w = np.where(live, nxt, w)
        depth = np.where(live, depth, depth - k)
    return (w >> 2).astype(np.int64)


def _expected_records(G, vox):
    if G.cells is None:                     # compressed grid: SmallCell + lists that end with a negative id
        sc = G.small_cells; refs_all = G.ref_ids
        ends = np.flatnonzero(refs_all < 0)
        begin_all = sc["begin"].astype(np.int64)
        n_all = np.where(begin_all >= 0, ends[np.searchsorted(ends, np.maximum(begin_all, 0))] - begin_all, 0)
        cells = np.zeros(len(sc), dtype=O_CELL_DTYPE)
        cells["min"] = sc["min"]; cells["max"] = sc["max"]; cells["begin"] = np.maximum(begin_all, 0); cells["end"] = np.maximum(begin_all, 0) + n_all
        cells = cells[_np_lookup(G, vox)]
    else:
        cells = G.cells[_np_lookup(G, vox)]
    lo, hi = cells["min"].astype(np.uint32), cells["max"].astype(np.uint32)
    n = (cells["end"] - cells["begin"]).astype(np.int64)
    rec = np.zeros((len(vox), 8), np.uint32)
    rec[:, 0] = lo[:, 0] | (hi[:, 0] << 16); rec[:, 1] = lo[:, 1] | (hi[:, 1] << 16); rec[:, 2] = lo[:, 2] | (hi[:, 2] << 16)
    rec[:, 3] = n.astype(np.uint32)
    refs = np.concatenate([G.ref_ids, np.zeros(4, np.int32)])
    for j in range(4):
        rec[:, 4 + j] = np.where(n > j, refs[cells["begin"] + j], -1).astype(np.int32).view(np.uint32)
    return rec, cells["begin"].astype(np.uint32)


def _check_records(got, want, begin):
    """got: records as the image resolves them; want: inline form of the construction format"""
    by_index = (got[:, 3] >> 31) == 1
    deep = ((got[:, 3] >> 30) & 1) == 1
    assert (got[:, :3] == want[:, :3]).all() and ((got[:, 3] & 0x3fffffff) == want[:, 3]).all()
    assert (got[~by_index, 4:] == want[~by_index, 4:]).all() and (want[~by_index, 3] <= 4).all()
    assert (got[by_index, 4] == begin[by_index]).all()
    assert ((want[by_index & ~deep, 3] > 4)).all()           # image records list by index only when the list is long
    return by_index, deep


def _walk_meets_link(G, vox, vtop):
    """general layout: does the walk to the voxel's record go through a link?  From the map's top level: where the top-level entry is subdivided.  From the
    image's virtual top level (one level down): where the top-level block has more than 2^3 entries, or the voxel's child of it is subdivided again."""
    t = vox.astype(np.int64) >> G.shift
    e = G.entries[t[:, 0] + G.dims[0] * (t[:, 1] + G.dims[1] * t[:, 2])].astype(np.int64)
    if not vtop or G.shift == 0: return (e & 3) != 0
    h = (vox.astype(np.int64) >> (G.shift - 1)) & 1
    child = G.entries[np.where((e & 3) == 1, (e >> 2) + h[:, 0] + 2 * h[:, 1] + 4 * h[:, 2], 0)].astype(np.int64)
    return ((e & 3) > 1) | (((e & 3) == 1) & ((child & 3) != 0))


def _image_scenes():
    sparse = scene.make_soup(3000, seed=5).copy()                      # two clusters far apart: top-level cells without subdivision
    sparse[:1500, 0:3] *= np.float32(0.2); sparse[1500:, 0:3] = sparse[1500:, 0:3] * np.float32(0.2) + np.float32(3.0)
    coincident = np.repeat(scene.make_soup(40, seed=6), 30, axis=0)    # lists far longer than four references
    return {"soup20k": (scene.make_soup(20000), {}), "soup30k_shift3": (scene.make_soup(30000, seed=11), dict(top_density=0.15, snd_density=3.0)),
            "dense_wide": (scene.make_soup(8000, seed=12), dict(top_density=0.08, snd_density=10.0)),     # a top-level cell with > 255 cells
            "deep": (scene.make_soup(6000, seed=12), dict(top_density=0.01, snd_density=40.0)),   # shift 5: blocks stop at depth 3, deep links below
            "sparse": (sparse, {}), "coincident": (np.concatenate([coincident, scene.make_soup(2000, seed=7)]), {}),
            "tiny": (scene.make_soup(3, seed=8), {}),
            "compressed": (scene.make_soup(20000, seed=14), dict(compress=True)),
            "compressed_deep": (scene.make_clustered(3000, 3, 4000), dict(compress=True)),          # shift 5, SmallCells: blocks + nested blocks, no deep links
            "compressed_long_lists": (np.concatenate([np.repeat(scene.make_soup(30, seed=15), 12, axis=0), scene.make_soup(6000, seed=16)]), dict(compress=True, top_density=0.3, snd_density=1.0))}


# (traverse.image, traverse.image_slim, traverse.image_general): the traversal image holds 16-byte slim records in one of three layouts -- grids of at most three
# levels: a block of records per top-level cell, table-free where (nearly) every top-level cell has the full depth (uniform layout), through the table otherwise
# (table layout; wide records for cells whose bounds do not fit a byte); every other grid a record per voxel-map entry (general layout: any depth, links to child
# blocks, wide records).  "image1": the value 1 of the option (round 1-4's compact form) builds the same image as 2; the 26-bit form of the record; the general
# layout forced on grids the block layouts would serve
_IMAGE_FORMATS = {"flat": (2, 1, 1), "image1": (1, 1, 1), "flat_slim26": (2, 2, 1), "flat_general": (2, 1, 2)}


@pytest.mark.parametrize("fmt_name", list(_IMAGE_FORMATS))
@pytest.mark.parametrize("name", list(_image_scenes()))
def test_traversal_image_resolves_every_voxel_to_its_cell(mem, name, fmt_name):
    """Every voxel of the virtual grid resolves, through the image's table / slot / record, to exactly the bounds, list
    length and reference ids that lookup_entry + cells + ref_ids give in the construction format."""
    from oracle import oracle as O
    fmt, slim, general = _IMAGE_FORMATS[fmt_name]
    tris, params = _image_scenes()[name]
    G = O.Grid.full(tris, **params)
    grid = upload_oracle_grid(mem, G)
    from hagrid_amd import api
    mem.set_option("traverse.image", fmt); mem.set_option("traverse.image_slim", slim); mem.set_option("traverse.image_general", general)
    mem.set_option("traverse.image_uniform", 0 if name == "soup30k_shift3" else 1)        # (the table layout alone: by default it comes next to a uniform layout, which the records below would be read from)
    api.setup_traversal(grid)
    mem.set_option("traverse.image", 2); mem.set_option("traverse.image_slim", 1); mem.set_option("traverse.image_general", 1); mem.set_option("traverse.image_uniform", 1)
    res = np.array(G.dims) << G.shift
    total = int(res[0]) * int(res[1]) * int(res[2])
    rng = np.random.default_rng(1)
    flat = np.arange(total) if total <= 400000 else rng.choice(total, 400000, replace=False)
    vox = np.stack([flat % res[0], (flat // res[0]) % res[1], flat // (res[0] * res[1])], axis=1).astype(np.int32)
    got = np.zeros((len(vox), 8), np.uint32); nbytes = C.c_int64(0)
    rc = mem._K.hagrid_kat_image_records(mem._ctx, C.byref(grid.pod), vox.ctypes.data_as(C.c_void_p), len(vox), got.ctypes.data_as(C.c_void_p), C.byref(nbytes))
    assert rc == 0
    want, begin = _expected_records(G, vox.astype(np.int64))
    by_index, deep = _check_records(got, want, begin)
    info = mem.image_format(grid)
    # every grid here fits slim records: what a byte cannot say goes into wide records
    assert info["flat"] and info["slim_id_bits"] == (26 if slim == 2 else 20) and info["record_bytes"] == 16, "slim records expected"
    assert not (info["uniform"] and info["general"])
    assert info["general"] == (general == 2 or G.shift > 3 or G.shift == 0)
    if info["uniform"]:
        assert nbytes.value == 16 * total + 8 * int(np.prod(G.dims))
    if info["general"]:
        assert nbytes.value >= 16 * G.num_entries and nbytes.value <= 16 * (G.num_entries + G.num_cells + 8 * int(np.prod(G.dims))) + 256       # (entries, wide records, virtual top level)
    if name == "soup30k_shift3":
        assert not info["uniform"] and info["general"] == (general == 2), "the table layout and the general layout on a grid of three levels are exercised"
    assert nbytes.value >= 16 * G.num_cells / 64 and nbytes.value < 600 * 32 * G.num_cells + 128 * np.prod(G.dims) + 4096
    if name == "coincident":
        assert by_index.any()
    if info["general"]:
        # bit 30 of the resolved record: the walk went through a link -- from the image's virtual top level, one level below the map's
        assert (deep == _walk_meets_link(G, vox, True)).all()
        if name in ("deep", "sparse", "coincident", "compressed_deep"): assert G.shift > 3 and deep.any()
    else:
        assert not deep.any()
    if name == "dense_wide":
        assert G.shift == 3
        top = vox >> 3
        per_top = {}
        key = (top[:, 0] + G.dims[0] * (top[:, 1] + G.dims[1] * top[:, 2])).astype(np.int64) * (G.num_cells + 1) + _np_lookup(G, vox.astype(np.int64))
        assert np.bincount(np.unique(key) // (G.num_cells + 1)).max() > 255        # (a top-level cell with more than 255 cells)
    grid.free()
    assert mem._K.hagrid_kat_image_records(mem._ctx, C.byref(grid.pod), None, 0, None, None) != 0      # the image went with the grid


@pytest.mark.parametrize("fmt_name", list(_IMAGE_FORMATS))
@pytest.mark.parametrize("name", list(_image_scenes()))
def test_image_kernel_gives_the_oracle_hits(mem, name, fmt_name):
    from oracle import oracle as O
    from hagrid_amd import api
    fmt, slim, general = _IMAGE_FORMATS[fmt_name]
    tris, params = _image_scenes()[name]
    G = O.Grid.full(tris, **params)
    d_tris = mem.upload(tris)
    grid = upload_oracle_grid(mem, G)
    rays = np.concatenate([scene.make_rays_primary(G.bbox_min, G.bbox_max, 256, 128),
                           scene.make_rays_incoherent(G.bbox_min - 0.2, G.bbox_max + 0.2, 60001, 17)]).astype(np.float32)
    want, _ = G.traverse(tris, rays, nthreads=8)
    try:
        mem.set_option("traverse.image", fmt); mem.set_option("traverse.image_slim", slim); mem.set_option("traverse.image_general", general)
        for uniform in ((1, 0) if slim == 1 and general == 1 else (1,)):          # blocks: table-free layout allowed / not allowed
            mem.set_option("traverse.image_uniform", uniform)
            for variant in (4, 0, 2):
                mem.set_option("traverse.variant", variant)
                for n in (rays.shape[0], 256 * 128, 65, 1):
                    got = gpu_traverse(mem, grid, d_tris, rays[:n])          # calls setup_traversal first
                    assert (got["id"] == want["id"][:n]).all() and (bits(got["t"]) == bits(want["t"][:n])).all(), (uniform, variant, n)
        mem.set_option("traverse.image_uniform", 1)
        mem.set_ray_binning(1); mem.set_option("traverse.variant", 4)
        got = gpu_traverse(mem, grid, d_tris, rays)
        assert (got["id"] == want["id"]).all() and (bits(got["t"]) == bits(want["t"])).all()
    finally:
        mem.set_ray_binning(0); mem.set_option("traverse.variant", 0); mem.set_option("traverse.image", 2); mem.set_option("traverse.image_uniform", 1)
        mem.set_option("traverse.image_slim", 1); mem.set_option("traverse.image_general", 1)
    grid.free(); mem.free(d_tris)


def test_cells_too_long_for_a_byte_get_wide_records(mem):
    """A cell that reaches more than 255 voxels away from one of its voxels does not fit the byte offsets of a slim record: the uniform layout
    cannot hold it, the table layout and the general layout give it a WIDE record (absolute 16-bit bounds, one per cell), and the hits stay the
    oracle's.  The 26-bit id form gives the same hits."""
    from oracle import oracle as O
    from hagrid_amd import api
    a = scene.make_soup(2000, seed=31).copy(); b = scene.make_soup(2000, seed=32).copy()
    b[:, 0] += np.float32(40.0)                                   # two clusters 40 units apart: long empty cells between them
    tris = np.ascontiguousarray(np.concatenate([a, b]))
    G = O.Grid.full(tris, top_density=0.5, snd_density=2.4)
    assert G.shift == 3 and int((G.cells["max"].astype(int) - G.cells["min"].astype(int)).max()) > 255
    total = int(np.prod(np.array(G.dims) << G.shift))
    d_tris = mem.upload(tris); grid = upload_oracle_grid(mem, G)
    rays = np.concatenate([scene.make_rays_primary(G.bbox_min, G.bbox_max, 128, 64),
                           scene.make_rays_incoherent(G.bbox_min - 0.2, G.bbox_max + 0.2, 30000, 19)]).astype(np.float32)
    want, _ = G.traverse(tris, rays, nthreads=8)
    nb = C.c_int64(0)
    try:
        for uniform in (2, 1):            # 2: the table-free layout whatever it costs (this grid is mostly empty) -- it does not fit; 1: not asked for
            mem.set_option("traverse.image_uniform", uniform)
            for slim in (1, 2):
                mem.set_option("traverse.image_slim", slim)
                for tail in (1, 0):
                    mem.set_option("traverse.tail", tail)
                    got = gpu_traverse(mem, grid, d_tris, rays)
                    assert (got["id"] == want["id"]).all() and (bits(got["t"]) == bits(want["t"])).all(), (uniform, slim, tail)
                mem.set_option("traverse.tail", 1)
                info = mem.image_format(grid)
                assert mem._K.hagrid_kat_image_records(mem._ctx, C.byref(grid.pod), None, 0, None, C.byref(nb)) == 0
                assert not info["general"] and not info["uniform"] and info["record_bytes"] == 16, "the table layout with wide records expected"
                assert 16 * G.num_cells / 8 < nb.value < 32 * total
                mem.set_option("traverse.image_general", 2)                      # the same grid in the general layout: wide records there as well
                got = gpu_traverse(mem, grid, d_tris, rays)
                assert (got["id"] == want["id"]).all() and (bits(got["t"]) == bits(want["t"])).all(), (uniform, slim, "general")
                info = mem.image_format(grid)
                assert mem._K.hagrid_kat_image_records(mem._ctx, C.byref(grid.pod), None, 0, None, C.byref(nb)) == 0
                assert info["general"] and 16 * G.num_entries + 16 <= nb.value <= 16 * (G.num_entries + G.num_cells + 8 * int(np.prod(G.dims)))      # (+ the virtual top level)
                mem.set_option("traverse.image_general", 1)
        mem.set_option("traverse.image_uniform", 2)
        # the same clusters close together: every cell fits, slim records in both id widths
        b[:, 0] -= np.float32(39.0)
        tris2 = np.ascontiguousarray(np.concatenate([a, b]))
        G2 = O.Grid.full(tris2, top_density=0.5, snd_density=2.4)
        total2 = int(np.prod(np.array(G2.dims) << G2.shift))
        assert 1 <= G2.shift <= 3 and int((np.array(G2.dims) << G2.shift).max()) <= 256
        d_tris2 = mem.upload(tris2); grid2 = upload_oracle_grid(mem, G2)
        rays2 = scene.make_rays_incoherent(G2.bbox_min - 0.2, G2.bbox_max + 0.2, 30000, 19).astype(np.float32)
        want2, _ = G2.traverse(tris2, rays2, nthreads=8)
        for slim in (1, 2):
            mem.set_option("traverse.image_slim", slim)
            got = gpu_traverse(mem, grid2, d_tris2, rays2)
            assert (got["id"] == want2["id"]).all() and (bits(got["t"]) == bits(want2["t"])).all(), slim
            assert mem._K.hagrid_kat_image_records(mem._ctx, C.byref(grid2.pod), None, 0, None, C.byref(nb)) == 0
            assert nb.value == 16 * total2 + 8 * int(np.prod(G2.dims)), "slim records expected"
            assert mem.image_format(grid2) == {"flat": True, "uniform": True, "general": False, "slim_id_bits": 26 if slim == 2 else 20, "record_bytes": 16, "two_layouts": False}
        grid2.free(); mem.free(d_tris2)
    finally:
        mem.set_option("traverse.image_uniform", 1); mem.set_option("traverse.image_slim", 1); mem.set_option("traverse.tail", 1); mem.set_option("traverse.image_general", 1)
    grid.free(); mem.free(d_tris)


def test_marker_ids_in_long_lists_widen_the_id_field(mem):
    """an id that a 20-bit field cannot tell from its markers (2^20 - 4 ... 2^20 - 1; the kernels also end a list BY INDEX at id 2^20 - 1) must
    widen the whole image to 26-bit fields wherever it occurs -- also when it occurs only in lists of more than four ids, which the general layout did not
    look at.  A grid whose four largest ids live only in a stack of twelve triangles (the LAST one nearest to the rays): general and table layout."""
    from oracle import oracle as O
    n = 3000
    base = scene.make_soup(n - 12, seed=41).copy()
    one = base[7:8].copy()
    one[0, 0:3] = np.float32(0.5)                                   # v0 in the middle of the scene; e1, e2, n as drawn
    stack = np.repeat(one, 12, axis=0)
    nrm = stack[0, [3, 7, 11]] / np.linalg.norm(stack[0, [3, 7, 11]])                      # (Tri: v0, nx | e1, ny | e2, nz)
    for i in range(12): stack[i, 0:3] += (np.float32(2e-5 * i) * nrm).astype(np.float32)   # id i of the stack lies 2e-5 i along the normal
    small = np.ascontiguousarray(np.concatenate([base, stack]).astype(np.float32))
    off = (1 << 20) - n                                              # ids off ... 2^20 - 1
    tris = np.zeros((off + n, small.shape[1]), np.float32); tris[off:] = small
    for params, general in ((dict(top_density=0.15, snd_density=3.0), 1), ({}, 2)):
        G = O.Grid.full(small, **params)
        refs = G.ref_ids; refs += off
        cells = G.cells
        lens = (cells["end"] - cells["begin"]).astype(np.int64)
        long_ids = np.concatenate([refs[b:e] for b, e, l in zip(cells["begin"], cells["end"], lens) if l > 4] or [np.zeros(0, np.int32)])
        short_ids = np.concatenate([refs[b:e] for b, e, l in zip(cells["begin"], cells["end"], lens) if 0 < l <= 4] or [np.zeros(0, np.int32)])
        assert long_ids.max() == (1 << 20) - 1 and short_ids.max() < (1 << 20) - 4, "the scene no longer isolates the case: marker ids only in lists by index"
        d_tris = mem.upload(tris); grid = upload_oracle_grid(mem, G)
        # rays along the stack's normal from both sides (the nearest of the twelve is the last or the first id), plus a general batch
        c = (small[-1, 0:3] + 0.25 * (-small[-1, 4:7] + small[-1, 8:11])).astype(np.float32)       # a point inside the triangle (v0 + (e2 - e1) / 4)
        rng = np.random.default_rng(5)
        k = 4096
        org = np.concatenate([c + 0.3 * nrm + 1e-3 * rng.standard_normal((k, 3)), c - 0.3 * nrm + 1e-3 * rng.standard_normal((k, 3))]).astype(np.float32)
        dirs = np.concatenate([np.tile(-nrm, (k, 1)), np.tile(nrm, (k, 1))]).astype(np.float32)
        aimed = np.zeros((2 * k, 8), np.float32); aimed[:, 0:3] = org; aimed[:, 3] = 0.0; aimed[:, 4:7] = dirs; aimed[:, 7] = np.float32(3.4e38)
        rays = np.ascontiguousarray(np.concatenate([aimed, scene.make_rays_incoherent(G.bbox_min - 0.1, G.bbox_max + 0.1, 20000, 7)]).astype(np.float32))
        want, _ = G.traverse(tris, rays, nthreads=8)
        assert (want["id"] == (1 << 20) - 1).any() and (want["id"] >= off).sum() > 1000
        try:
            mem.set_option("traverse.image_general", general)
            for tail in (1, 0):
                mem.set_option("traverse.tail", tail)
                got = gpu_traverse(mem, grid, d_tris, rays)
                assert (got["id"] == want["id"]).all() and (bits(got["t"]) == bits(want["t"])).all(), (general, tail)
            info = mem.image_format(grid)
            assert info["general"] == (general == 2) and info["slim_id_bits"] == 26, info
        finally:
            mem.set_option("traverse.image_general", 1); mem.set_option("traverse.tail", 1)
        grid.free(); mem.free(d_tris)


def test_general_layout_virtual_top_level_on_and_off(mem):
    """The general layout's virtual top level ("traverse.image_vtop", trav_image.hip image_general_vtop: eight records per top-level cell, where look-ups that left
    their block start again) only shortens the walk: with it and without it every kernel gives the oracle's hits, and every voxel resolves to its cell."""
    from oracle import oracle as O
    from hagrid_amd import api
    scenes = {"deep": (scene.make_soup(6000, seed=12), dict(top_density=0.01, snd_density=40.0)),       # shift 5, top-level blocks of more than 2^3 entries (links seen from one level down)
              "clustered": (scene.make_clustered(3000, 3, 4000), {}),
              "shallow": (scene.make_soup(20000, seed=11), dict(top_density=0.15, snd_density=3.0)),    # three levels, general layout forced
              "one_level": (scene.make_soup(300, seed=5), dict(top_density=0.12, snd_density=0.01))}   # shift 0: no level below the top level, no virtual one
    nb = C.c_int64(0)
    try:
        for name, (tris, params) in scenes.items():
            G = O.Grid.full(tris, **params)
            assert (G.shift == 0) == (name == "one_level"), (name, G.shift)
            d_tris = mem.upload(tris); grid = upload_oracle_grid(mem, G)
            lo, hi = np.asarray(G.bbox_min), np.asarray(G.bbox_max)
            rays = np.concatenate([scene.make_rays_primary(lo, hi, 96, 64), scene.make_rays_incoherent(lo - 0.3, hi + 0.3, 20011, 29)]).astype(np.float32)
            want, _ = G.traverse(tris, rays, nthreads=8)
            rng = np.random.default_rng(3); total = np.array(G.dims) << G.shift
            vox = np.ascontiguousarray(np.stack([rng.integers(0, total[c], 50000) for c in range(3)], axis=1).astype(np.int32))
            mem.set_option("traverse.image_general", 2)
            size = {}
            for vtop in (1, 0):
                mem.set_option("traverse.image_vtop", vtop)
                for tail in (1, 0):
                    mem.set_option("traverse.tail", tail)
                    for flags in (0, api.ANY_HIT):
                        if flags and tail: continue
                        got = gpu_traverse(mem, grid, d_tris, rays) if not flags else None
                        if got is not None: assert (got["id"] == want["id"]).all() and (bits(got["t"]) == bits(want["t"])).all(), (name, vtop, tail)
                mem.set_option("traverse.tail", 1)
                assert mem.image_format(grid)["general"]
                recs = np.zeros((vox.shape[0], 8), np.uint32)
                assert mem._K.hagrid_kat_image_records(mem._ctx, C.byref(grid.pod), vox.ctypes.data_as(C.c_void_p), vox.shape[0], recs.ctypes.data_as(C.c_void_p), C.byref(nb)) == 0
                size[vtop] = nb.value
                # every voxel's record resolves to the cell (bounds, list) the construction format gives, through links exactly where the walk meets one
                rec_want, begin = _expected_records(G, vox.astype(np.int64))
                _, deep = _check_records(recs, rec_want, begin)
                assert (deep == _walk_meets_link(G, vox, vtop == 1)).all(), (name, vtop)
                if name == "clustered": assert deep.any() and not deep.all()
                if name == "deep": assert deep.all()                                  # top-level blocks of 8^3 entries: a link either way (seen from one level down: up = 1)
            assert size[1] - size[0] == (128 * int(np.prod(G.dims)) if G.shift >= 1 else 0), (name, size)
            grid.free(); mem.free(d_tris)
    finally:
        mem.set_option("traverse.image_vtop", 1); mem.set_option("traverse.image_general", 1); mem.set_option("traverse.tail", 1)


@pytest.mark.parametrize("family", ["gradient", "shell"])
def test_head_share_trial_never_changes_hits(mem, family):
    """The tile order's head share ("traverse.quad_head") on scene families its counting rule was not fitted on: a soup with a density gradient (the share is
    suggested, measured and dropped: the order is stored rotated, then un-rotated again) and a sphere shell.  140 launches over one buffer of a launch of more than one
    round: every checked launch -- before the share, with it, after a drop -- gives the oracle's hits; with a lower threshold (more tiles at the head) as well."""
    from oracle import oracle as O
    from hagrid_amd import api
    tris = scene.make_gradient(150000, seed=3) if family == "gradient" else scene.make_shell(150000, seed=4)
    G = O.Grid.full(tris)
    d_tris = mem.upload(tris); grid = upload_oracle_grid(mem, G)
    rays = scene.make_rays_primary(np.asarray(G.bbox_min), np.asarray(G.bbox_max), 1024, 640).astype(np.float32); n = rays.shape[0]
    want, _ = G.traverse(tris, rays, nthreads=8)
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    api.setup_traversal(grid)
    try:
        for head in (20, 11):
            mem.set_option("traverse.quad_head", head)
            for launch in range(1, 141):
                check = launch in (1, 2, 3, 30, 60, 61, 62, 90, 100, 101, 120, 140)
                if check: mem.zero(d_hits, 16 * n)
                api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
                if check:
                    mem.synchronize()
                    got = mem.download(d_hits, api.HIT_DTYPE, n)
                    assert (got["id"] == want["id"]).all() and (bits(got["t"]) == bits(want["t"])).all(), (family, head, launch)
    finally:
        mem.set_option("traverse.quad_head", 20)
    mem.free(d_rays); mem.free(d_hits); grid.free(); mem.free(d_tris)


def test_share_trial_and_the_order_held_against_it_never_change_hits(mem):
    """Round 6: launches in the default tile order measure the share of tiles that start with four lanes per ray (candidates: the rule's share, a half, none, all --
    every sample with its own event pair), and a learned order is then held against the winner (traverse.hip "traverse.share_trial").  Every launch of the sequence --
    the samples in the default order, the order learned behind them, its own trial, the choice, a re-trial -- gives the oracle's hits, on an image-ordered batch, on
    bounce-like rays (image order without coherent directions) and on a binned incoherent batch; the trial reaches a choice (synchronising callers: after a few dozen
    launches), a buffer of the same shape inherits it, and with the option off the state stays untouched."""
    from oracle import oracle as O
    from hagrid_amd import api
    tris = scene.make_clustered(30000, 3, 40000)
    G = O.Grid.full(tris)
    d_tris = mem.upload(tris); grid = upload_oracle_grid(mem, G)
    lo, hi = np.asarray(G.bbox_min), np.asarray(G.bbox_max)
    prim = scene.make_rays_primary(lo, hi, 1024, 640).astype(np.float32)
    inc = scene.make_rays_incoherent(lo, hi, prim.shape[0], 31).astype(np.float32)
    bounce = prim.copy(); bounce[:, 4:7] = inc[:, 4:7]                     # origins in image order, directions anywhere: rows found from the origins alone (or none: both fine)
    api.setup_traversal(grid)
    try:
        for name, rays, binning in (("primary", prim, 0), ("bounce-like", bounce, 0), ("incoherent binned", inc, 1)):
            n = rays.shape[0]
            want, _ = G.traverse(tris, rays, nthreads=8)
            d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
            mem.set_ray_binning(binning)
            for launch in range(1, 91):
                mem.zero(d_hits, 16 * n)
                api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
                mem.synchronize()
                if launch <= 30 or launch % 6 == 0:
                    got = mem.download(d_hits, api.HIT_DTYPE, n)
                    assert (got["id"] == want["id"]).all() and (bits(got["t"]) == bits(want["t"])).all(), (name, launch, mem.order_state(d_rays))
            st = mem.order_state(d_rays)
            assert st["slot"] >= 0 and st["share_choice"] in (0, 25, 37, 50, 100), (name, st)          # a choice was made (a percentage of the tiles)
            if name == "primary":
                # another buffer of the same shape starts with the answer
                d_rays2 = mem.upload(rays)
                api.traverse_grid(grid, d_tris, d_rays2, d_hits, n); mem.synchronize()
                st2 = mem.order_state(d_rays2)
                assert st2["share_choice"] == st["share_choice"], (st, st2)
                got = mem.download(d_hits, api.HIT_DTYPE, n)
                assert (got["id"] == want["id"]).all() and (bits(got["t"]) == bits(want["t"])).all()
                mem.free(d_rays2)
            mem.set_ray_binning(0)
            mem.free(d_rays); mem.free(d_hits)
        # off: no samples are taken
        mem.set_option("traverse.share_trial", 0)
        n = prim.shape[0]; d_rays = mem.upload(prim[::-1].copy()); d_hits = mem.alloc(16 * n)
        api.traverse_grid(grid, d_tris, d_rays, d_hits, n); mem.synchronize()
        before = mem.order_state(d_rays)["share_samples"]                      # (a recycled address finds the slot of the buffer that lived there)
        for _ in range(20): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        mem.synchronize()
        assert mem.order_state(d_rays)["share_samples"] == before
        mem.free(d_rays); mem.free(d_hits)
    finally:
        mem.set_option("traverse.share_trial", 1); mem.set_ray_binning(0)
    grid.free(); mem.free(d_tris)


def test_policy_state_machine_under_a_random_sequence_of_launches(mem):
    """The measured dispatch policy (share trial, order or no order, head share, all tiles, re-trials, hint slots taken over by other buffers, another traversal image
    under the same buffers) is a state machine per ray buffer; whatever state a sequence of calls leaves it in, a launch gives the oracle's hits.  300 launches over SIX
    ray buffers (the context remembers four) of five shapes and three ray kinds, in random order, with refills (another frame into the same buffer), ray binning switched
    on and off, occasional synchronisation, and the grid rebuilt with other parameters half-way (same scene: same hits)."""
    from oracle import oracle as O
    from hagrid_amd import api
    rng = np.random.default_rng(77)
    tris = scene.make_clustered(20000, 3, 30000)
    G = O.Grid.full(tris)
    d_tris = mem.upload(tris)
    grid = api.build_all(mem, d_tris, tris.shape[0]); api.setup_traversal(grid)
    lo, hi = np.asarray(G.bbox_min), np.asarray(G.bbox_max)
    shapes = [(512, 512), (1024, 256), (640, 480), (200, 77), (768, 768)]
    def frame(shape, kind, f):
        w, h = shape
        prim = scene.make_rays_primary(lo, hi, w, h, yaw=0.01 * f, strafe=0.01 * f).astype(np.float32)
        if kind == "primary": return prim
        inc = scene.make_rays_incoherent(lo, hi, prim.shape[0], 100 + f).astype(np.float32)
        if kind == "incoherent": return inc
        b = prim.copy(); b[:, 4:7] = inc[:, 4:7]; return b                                  # bounce-like: origins in image order, directions anywhere
    want_cache = {}
    def want(shape, kind, f):
        key = (shape, kind, f)
        if key not in want_cache:
            want_cache[key] = G.traverse(tris, np.ascontiguousarray(frame(shape, kind, f)), nthreads=8)[0]
        return want_cache[key]
    bufs = []
    for i in range(6):
        shape = shapes[i % len(shapes)]; kind = ("primary", "incoherent", "bounce")[i % 3]
        n = shape[0] * shape[1]
        bufs.append(dict(shape=shape, kind=kind, f=0, n=n, d_rays=mem.upload(np.ascontiguousarray(frame(shape, kind, 0))), d_hits=mem.alloc(16 * n)))
    try:
        for launch in range(300):
            b = bufs[int(rng.integers(len(bufs))) if launch % 5 else int(rng.integers(2))]        # (two buffers get most of the launches: their trials conclude)
            r = rng.random()
            if r < 0.08:                                             # another frame into the same buffer
                b["f"] = int(rng.integers(4)); mem.copy_h2d(b["d_rays"], np.ascontiguousarray(frame(b["shape"], b["kind"], b["f"])))
            if launch == 150:                                        # another grid (and traversal image) of the same scene under the same buffers
                grid.free(); grid = api.build_all(mem, d_tris, tris.shape[0], top_density=0.2, snd_density=3.0); api.setup_traversal(grid)
            mem.set_ray_binning(1 if (b["kind"] == "incoherent" and rng.random() < 0.7) else 0)
            mem.zero(b["d_hits"], 16 * b["n"])
            api.traverse_grid(grid, d_tris, b["d_rays"], b["d_hits"], b["n"])
            if rng.random() < 0.5: mem.synchronize()
            got = mem.download(b["d_hits"], api.HIT_DTYPE, b["n"])
            w = want(b["shape"], b["kind"], b["f"])
            assert (got["id"] == w["id"]).all() and (bits(got["t"]) == bits(w["t"])).all(), (launch, b["shape"], b["kind"], b["f"], mem.order_state(b["d_rays"]))
    finally:
        mem.set_ray_binning(0)
        for b in bufs: mem.free(b["d_rays"]); mem.free(b["d_hits"])
    grid.free(); mem.free(d_tris)


def test_image_lifetime(mem):
    """The image belongs to the grid of the last setup_traversal call and never outlives its source arrays."""
    from oracle import oracle as O
    from hagrid_amd import api
    tris = scene.make_soup(5000, seed=21)
    G = O.Grid.full(tris)
    d_tris = mem.upload(tris)
    grid = upload_oracle_grid(mem, G)
    rays = scene.make_rays_incoherent(G.bbox_min, G.bbox_max, 20000, 4)
    want, _ = G.traverse(tris, rays, nthreads=4)
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * rays.shape[0])
    has_image = lambda g: mem._K.hagrid_kat_image_records(mem._ctx, C.byref(g.pod), None, 0, None, None) == 0
    def check():
        api.traverse_grid(grid, d_tris, d_rays, d_hits, rays.shape[0])
        got = mem.download(d_hits, api.HIT_DTYPE, rays.shape[0])
        assert (got["id"] == want["id"]).all() and (bits(got["t"]) == bits(want["t"])).all()
    try:
        api.setup_traversal(grid); assert has_image(grid)          # built by default
        mem.set_option("traverse.image", 0); api.setup_traversal(grid); assert not has_image(grid)
        mem.set_option("traverse.image", 1)
        mem.set_option("traverse.variant", 4)
        with pytest.raises(api.HagridError):            # forced image kernel, no image yet
            api.traverse_grid(grid, d_tris, d_rays, d_hits, rays.shape[0])
        mem.set_option("traverse.variant", 0)
        assert not has_image(grid); check()                 # construction format
        api.setup_traversal(grid); assert has_image(grid); check()
        # a second grid in the same context takes the image over; the first one still traverses (construction format)
        grid2 = upload_oracle_grid(mem, G)
        api.setup_traversal(grid2); assert has_image(grid2) and not has_image(grid); check()
        grid2.free()
        api.setup_traversal(grid); assert has_image(grid)
        # overwriting a source array through the API drops the image
        cells = mem.download(grid.pod.cells, np.uint8, 32 * G.num_cells)
        mem.copy_h2d(grid.pod.cells, cells); assert not has_image(grid); check()
        # a construction pass in the context drops it as well
        api.setup_traversal(grid); assert has_image(grid)
        other = api.build_all(mem, d_tris, tris.shape[0]); assert not has_image(grid); check()
        other.free()
        # switched off: setup_traversal builds nothing
        mem.set_option("traverse.image", 0); api.setup_traversal(grid); assert not has_image(grid); check()
        mem.set_option("traverse.image", 1); api.setup_traversal(grid); assert has_image(grid); check()
        # an image above the size limit is not built: traversal reads the construction format
        mem.set_option("traverse.image", 2); mem.set_option("traverse.image_max_mb", 1)
        big = O.Grid.full(scene.make_soup(40000, seed=22)); gb = upload_oracle_grid(mem, big)
        nb = C.c_int64(0)
        api.setup_traversal(gb); assert not has_image(gb)
        mem.set_option("traverse.image_max_mb", 0); api.setup_traversal(gb)
        assert mem._K.hagrid_kat_image_records(mem._ctx, C.byref(gb.pod), None, 0, None, C.byref(nb)) == 0 and (4 << 20) < nb.value < (5 << 20)
        mem.set_option("traverse.image_max_mb", 5); api.setup_traversal(gb); assert has_image(gb)          # (it fits five megabytes)
        mem.set_option("traverse.image_max_mb", 0)
        gb.free(); mem.set_option("traverse.image", 1)
        # compressed grids get one as well
        Gc = O.Grid.full(tris, compress=True); gc = upload_oracle_grid(mem, Gc)
        api.setup_traversal(gc); assert has_image(gc); gc.free()
        api.setup_traversal(grid); assert has_image(grid)
        grid.free()                                          # freeing a source array drops the image
        assert not mem._K.hagrid_kat_image_records(mem._ctx, C.byref(grid.pod), None, 0, None, None) == 0
    finally:
        mem.set_option("traverse.variant", 0); mem.set_option("traverse.image", 2); mem.set_option("traverse.image_max_mb", 0); mem.set_option("traverse.image_slim", 1)
    mem.free(d_rays); mem.free(d_hits); mem.free(d_tris)


# ---- tail mode (traverse_kernel_tail) ---------------------------------------------------------------------------------------

@pytest.mark.parametrize("slim", [1, 2])
def test_tail_mode_gives_the_oracle_hits(mem, slim):
    """traverse_kernel_tail: a wavefront that holds at most 16 live rays compacts to four lanes per ray and tests a cell's inline list
    in one round, replaying the acceptance in list order; the last tiles of a launch may start in that form ("traverse.quad_tail": four blocks
    of 16 rays per tile, a 4 x 4 pixel quadrant each).  Hits must stay the oracle's bit for bit: wavefronts that are sparse from the
    start (batches of 1 .. 17 rays, rays that miss the grid), that thin out on the way (64 coherent rays), lists longer than a record
    holds (by index) met inside the tail phase, the 20- and 26-bit id forms, the general layout (grids of three levels with top-level cells of
    different depth, grids of five levels, wide records, compressed), binned batches -- and the kernel without the tail mode for comparison."""
    from oracle import oracle as O
    from hagrid_amd import api
    coincident = np.repeat(scene.make_soup(60, seed=41), 9, axis=0)          # nine copies of every triangle: equal t, lists of 9+ ids
    scenes = {"soup": (scene.make_soup(30000, seed=42), {}),
              "long_lists": (np.concatenate([coincident, scene.make_soup(4000, seed=43)]), dict(top_density=0.3, snd_density=1.0)),
              "table_layout": (scene.make_soup(30000, seed=11), dict(top_density=0.15, snd_density=3.0)),     # three levels, top-level cells of different depth: table layout
              "two_layouts": (scene.make_soup(30000, seed=11), dict(top_density=0.15, snd_density=3.0)),      # the same grid as the defaults hold it: the uniform layout for rays in image order, the table layout next to it for binned batches
              "general_shallow": (scene.make_soup(30000, seed=11), dict(top_density=0.15, snd_density=3.0)),  # the same grid in the general layout (forced)
              "deep": (scene.make_soup(6000, seed=12), dict(top_density=0.01, snd_density=40.0)),             # shift 5: links below the top level
              "clustered": (scene.make_clustered(3000, 3, 4000), {}),                                          # shift 5, blobs in a sparse soup: wide records between them
              "clustered_compressed": (scene.make_clustered(3000, 3, 4000), dict(compress=True))}
    try:
        mem.set_option("traverse.image_slim", slim)
        for name, (tris, params) in scenes.items():
            G = O.Grid.full(tris, **params)
            assert (G.shift > 3) == (name in ("deep", "clustered", "clustered_compressed"))
            d_tris = mem.upload(tris); grid = upload_oracle_grid(mem, G)
            lo, hi = np.asarray(G.bbox_min), np.asarray(G.bbox_max)
            primary = scene.make_rays_primary(lo, hi, 64, 48)
            rays = np.concatenate([primary, scene.make_rays_incoherent(lo - 0.3, hi + 0.3, 20011, 23)]).astype(np.float32)
            want, _ = G.traverse(tris, rays, nthreads=8)
            mem.set_option("traverse.image_general", 2 if name == "general_shallow" else 1)
            mem.set_option("traverse.image_uniform", 0 if name == "table_layout" else 1)            # (the table layout alone: forced; by default it comes next to the uniform one)
            api.setup_traversal(grid)
            info = mem.image_format(grid)
            assert info["slim_id_bits"] == (26 if slim == 2 else 20), (name, info)
            if name != "long_lists": assert info["uniform"] == (name in ("soup", "two_layouts")) and info["general"] == (name not in ("soup", "table_layout", "two_layouts")), (name, info)      # the three slim layouts are exercised
            if name != "long_lists": assert info["two_layouts"] == (name == "two_layouts"), (name, info)          # ("long_lists": two layouts as well, whatever its levels give)
            # (tail mode, per cent of the tiles that START with four lanes per ray -- "traverse.quad_tail", 16 rays per wavefront)
            # ... and "traverse.tail_dual": two ids of an inline list per round trip in phase 1, the second triangle through LDS (forced on
            # for binned batches as well, where the default switches it off)
            # ... and "traverse.mailbox": a ray skips a triangle it was tested against among its last four tests (nine coincident copies of every
            # triangle in "long_lists": equal t, the first copy must win every time)
            for tail, quad, dual, mbox in ((1, -1, -1, -1), (1, 0, 1, 0), (1, 0, 0, 1), (1, 30, 1, 1), (1, 100, 0, 1), (1, -1, -1, 1), (0, 0, -1, 0)):
                mem.set_option("traverse.tail", tail); mem.set_option("traverse.quad_tail", quad); mem.set_option("traverse.tail_dual", dual)
                mem.set_option("traverse.mailbox", mbox)
                for binning in (0, 1):
                    mem.set_ray_binning(binning)
                    for first, n in ((0, rays.shape[0]), (0, 64 * 48), (0, 64), (5, 1), (7, 15), (0, 16), (3, 17), (64 * 48, 4099)):
                        got = gpu_traverse(mem, grid, d_tris, rays[first:first + n])
                        w = want[first:first + n]
                        assert (got["id"] == w["id"]).all() and (bits(got["t"]) == bits(w["t"])).all(), (name, tail, quad, dual, mbox, binning, first, n)
            if name == "long_lists":
                assert (want["id"] >= 0).any()
            grid.free(); mem.free(d_tris)
    finally:
        mem.set_option("traverse.tail", 1); mem.set_option("traverse.quad_tail", -1); mem.set_option("traverse.tail_dual", -1)
        mem.set_option("traverse.image_slim", 1); mem.set_ray_binning(0); mem.set_option("traverse.mailbox", -1)
        mem.set_option("traverse.image_general", 1); mem.set_option("traverse.image_uniform", 1)


def test_row_length_cache_never_changes_hits(mem):
    """The row length found for a ray buffer is reused by later calls with the same buffer and count; it only steers the lane <-> ray
    assignment.  One buffer refilled in turn with images of two widths and with unordered rays: every call gives the oracle's hits."""
    from oracle import oracle as O
    from hagrid_amd import api
    tris = scene.make_soup(20000, seed=51)
    G = O.Grid.full(tris)
    d_tris = mem.upload(tris); grid = upload_oracle_grid(mem, G)
    lo, hi = np.asarray(G.bbox_min), np.asarray(G.bbox_max)
    n = 128 * 64
    batches = [scene.make_rays_primary(lo, hi, 128, 64), scene.make_rays_primary(lo, hi, 64, 128),
               scene.make_rays_incoherent(lo - 0.2, hi + 0.2, n, 29)]
    batches = [np.ascontiguousarray(b, np.float32) for b in batches]
    want = [G.traverse(tris, b, nthreads=8)[0] for b in batches]
    api.setup_traversal(grid)
    d_rays = mem.upload(batches[0]); d_hits = mem.alloc(16 * n)
    try:
        for cache in (1, 0):
            mem.set_option("traverse.row_cache", cache)
            for call in range(40):
                k = (call * 7 + call // 5) % 3
                mem.copy_h2d(d_rays, batches[k])
                api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
                got = mem.download(d_hits, api.HIT_DTYPE, n)
                assert (got["id"] == want[k]["id"]).all() and (bits(got["t"]) == bits(want[k]["t"])).all(), (cache, call, k)
    finally:
        mem.set_option("traverse.row_cache", 1)
    mem.free(d_rays); mem.free(d_hits); grid.free(); mem.free(d_tris)


@pytest.mark.parametrize("params", [{}, dict(top_density=0.15, snd_density=3.0)], ids=["table_free", "table_layout"])
def test_tile_order_never_changes_hits(mem, params):
    """Launches over a ray buffer the context has seen before dispatch their tiles longest first, by the costs the previous launches left
    ("traverse.tile_order"; the order is sorted behind the launch that learns and behind every 32nd one after it).  It only steers which
    wavefront takes which rays: one buffer traversed again and again, refilled in between with an image of another width, with unordered rays
    and with the first image again -- stale orders, orders of another tiling, no order -- gives the oracle's hits at every call, with every share
    of the tiles starting with four lanes per ray, forced on, by default and off."""
    from oracle import oracle as O
    from hagrid_amd import api
    tris = scene.make_soup(20000, seed=52)
    G = O.Grid.full(tris, **params)
    d_tris = mem.upload(tris); grid = upload_oracle_grid(mem, G)
    lo, hi = np.asarray(G.bbox_min), np.asarray(G.bbox_max)
    n = 128 * 64
    batches = [scene.make_rays_primary(lo, hi, 128, 64), scene.make_rays_primary(lo, hi, 64, 128),
               scene.make_rays_incoherent(lo - 0.2, hi + 0.2, n, 31), scene.make_rays_primary(lo, hi, 256, 32)]
    batches = [np.ascontiguousarray(b, np.float32) for b in batches]
    want = [G.traverse(tris, b, nthreads=8)[0] for b in batches]
    mem.set_option("traverse.image_uniform", 0 if params else 1)          # (the table layout alone for the second grid: rays in image order would gather from the uniform layout the defaults put next to it)
    api.setup_traversal(grid)
    mem.set_option("traverse.image_uniform", 1)
    info = mem.image_format(grid)
    assert info["slim_id_bits"] == 20 and info["uniform"] == (not params), info          # both layouts of the tail kernel
    d_rays = mem.upload(batches[0]); d_hits = mem.alloc(16 * n)
    try:
        for order, quad in ((1, -1), (-1, -1), (1, 30), (1, 100), (1, 0), (0, -1)):
            mem.set_option("traverse.tile_order", order); mem.set_option("traverse.quad_tail", quad)
            for call in range(126):
                k = (call // 36) % 4 if call < 108 else (call * 7) % 4     # long runs over one filling (the order is learned, refreshed, reused), then a new filling per call
                mem.copy_h2d(d_rays, batches[k])
                api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
                got = mem.download(d_hits, api.HIT_DTYPE, n)
                assert (got["id"] == want[k]["id"]).all() and (bits(got["t"]) == bits(want[k]["t"])).all(), (order, quad, call, k)
        # a shorter batch in the same buffer (another tile count: the order starts over), then the long one again
        mem.set_option("traverse.tile_order", 1); mem.set_option("traverse.quad_tail", -1)
        mem.copy_h2d(d_rays, batches[0])
        for call in range(40):
            m = n if (call // 10) % 2 == 0 else 128 * 40
            api.traverse_grid(grid, d_tris, d_rays, d_hits, m)
            got = mem.download(d_hits, api.HIT_DTYPE, m)
            assert (got["id"] == want[0]["id"][:m]).all() and (bits(got["t"]) == bits(want[0]["t"][:m])).all(), (call, m)
        # several buffers in turn (the context keeps the hints of four): five buffers over four slots, two of them traded places half way
        bufs = [mem.upload(batches[k % 4]) for k in range(5)]
        for call in range(150):
            j = call % 5 if call < 100 else (call * 3) % 5
            if call == 75: mem.copy_h2d(bufs[1], batches[3]); mem.copy_h2d(bufs[3], batches[1])
            k = (j % 4) if call < 75 or j not in (1, 3) else (3 if j == 1 else 1)
            api.traverse_grid(grid, d_tris, bufs[j], d_hits, n)
            got = mem.download(d_hits, api.HIT_DTYPE, n)
            assert (got["id"] == want[k]["id"]).all() and (bits(got["t"]) == bits(want[k]["t"])).all(), ("buffers in turn", call, j, k)
        for b in bufs: mem.free(b)
        # the row length given by the caller instead of looked for ("traverse.image_width"): the order applies from the second call on
        mem.set_option("traverse.image_width", 128); mem.copy_h2d(d_rays, batches[0])
        for call in range(40):
            api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
            got = mem.download(d_hits, api.HIT_DTYPE, n)
            assert (got["id"] == want[0]["id"]).all() and (bits(got["t"]) == bits(want[0]["t"])).all(), ("given width", call)
    finally:
        mem.set_option("traverse.image_width", 0)
        mem.set_option("traverse.tile_order", -1); mem.set_option("traverse.quad_tail", -1)
    mem.free(d_rays); mem.free(d_hits); grid.free(); mem.free(d_tris)


# ---- any-hit and barycentrics ------------------------------------------------------------------------

def test_intersect_prim_ray_with_uvs_matches_reference_header(mem, golden_dir):
    """The device's COMPUTE_UVS form of intersect_prim_ray against the reference header compiled with -DCOMPUTE_UVS."""
    kat = np.load(os.path.join(golden_dir, "l0_kat.npz")); uv = np.load(os.path.join(golden_dir, "l0_kat_uvs.npz"))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    tris = np.ascontiguousarray(kat["tris"]); rays = np.ascontiguousarray(kat["ipr_rays"]); tid = np.ascontiguousarray(kat["ipr_tid"])
    n = rays.shape[0]
    ret = np.zeros(n, np.int32); hid = np.zeros(n, np.int32); ht = np.zeros(n, np.float32); hu = np.zeros(n, np.float32); hv = np.zeros(n, np.float32)
    assert mem._K.hagrid_kat_intersect_prim_ray_uvs(mem._ctx, p(tris), p(rays), p(tid), n, p(ret), p(hid), p(ht), p(hu), p(hv)) == 0
    assert (ret == uv["ret"]).all() and (hid == uv["id"]).all() and (bits(ht) == bits(uv["t"])).all()
    assert (bits(hu) == bits(uv["u"])).all() and (bits(hv) == bits(uv["v"])).all()


@pytest.mark.parametrize("compressed", [False, True])
def test_any_hit_and_uvs_traversal(mem, compressed):
    from oracle import oracle as O
    from hagrid_amd import api
    tris = scene.make_soup(50000, seed=77)
    G = O.Grid.full(tris, compress=compressed)
    d_tris = mem.upload(tris)
    grid = upload_oracle_grid(mem, G)
    rays = np.concatenate([scene.make_rays_primary(G.bbox_min, G.bbox_max, 320, 200),
                           scene.make_rays_incoherent(G.bbox_min - 0.1, G.bbox_max + 0.1, 90001, 23)]).astype(np.float32)
    rays[1000:2000, 7] = 0.3                    # short shadow-ray-like segments
    n = rays.shape[0]
    nearest, _ = G.traverse(tris, rays, nthreads=8)
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    def run(flags):
        api.traverse_grid(grid, d_tris, d_rays, d_hits, n, flags)
        return mem.download(d_hits, api.HIT_DTYPE, n)
    api.setup_traversal(grid)                   # uncompressed: the image kernel has these variants too; compressed: v2
    for binning, variant in ((0, 0), (1, 0), (0, 2)):
        mem.set_ray_binning(binning); mem.set_option("traverse.variant", variant)
        try:
            got = run(api.UVS)
            want = G.traverse_ex(tris, rays, O.UVS, nthreads=8)
            assert (got["id"] == nearest["id"]).all() and (bits(got["t"]) == bits(nearest["t"])).all()
            assert (bits(got["u"]) == bits(want["u"])).all() and (bits(got["v"]) == bits(want["v"])).all()
            hit = got["id"] >= 0
            assert hit.any() and (got["u"][hit] >= -1e-6).all() and (got["v"][hit] >= -1e-6).all() and (got["u"][hit] + got["v"][hit] <= 1 + 1e-5).all()
            assert (got["u"][~hit] == 0).all() and (got["v"][~hit] == 0).all()
            # the barycentrics reproduce the hit point: v0 - u*e1 + v*e2 = org + t*dir (Tri stores e1 = v0 - v1, e2 = v2 - v0)
            t = tris[got["id"][hit]]
            pt = t[:, 0:3] - got["u"][hit, None] * t[:, 4:7] + got["v"][hit, None] * t[:, 8:11]
            pr = rays[hit, 0:3] + got["t"][hit, None] * rays[hit, 4:7]
            assert np.abs(pt - pr).max() < 2e-3

            anyh = run(api.ANY_HIT)
            want = G.traverse_ex(tris, rays, O.ANY_HIT, nthreads=8)
            assert (anyh["id"] == want["id"]).all() and (bits(anyh["t"]) == bits(want["t"])).all()
            assert ((anyh["id"] >= 0) == (nearest["id"] >= 0)).all()             # occluded <=> the nearest-hit walk finds something
            assert (anyh["t"][hit] >= nearest["t"][hit]).all()
            assert (anyh["id"] != nearest["id"]).any()                               # and it really stops early somewhere
            both = run(api.ANY_HIT | api.UVS)
            want = G.traverse_ex(tris, rays, O.ANY_HIT | O.UVS, nthreads=8)
            for f in ("id", "t", "u", "v"):
                assert (bits(both[f]) == bits(want[f])).all() if f != "id" else (both[f] == want[f]).all()
        finally:
            mem.set_ray_binning(0); mem.set_option("traverse.variant", 0)
    with pytest.raises(api.HagridError):
        api.traverse_grid(grid, d_tris, d_rays, d_hits, n, 8)
    mem.free(d_rays); mem.free(d_hits); grid.free(); mem.free(d_tris)


def test_any_hit_and_uvs_on_a_grid_with_two_layouts(mem):
    """A grid of three levels with unevenly deep top-level cells holds a uniform layout and the table layout next to it (round 6): rays in image order gather from
    the first, binned batches from the second -- in the nearest-hit kernel and in the any-hit / barycentric variants alike; a second context that borrows the image
    (hagrid_share_traversal) gets both.  Every combination gives the oracle's hits."""
    from oracle import oracle as O
    from hagrid_amd import api
    tris = scene.make_soup(30000, seed=11)
    G = O.Grid.full(tris, top_density=0.15, snd_density=3.0)
    assert G.shift == 3
    d_tris = mem.upload(tris); grid = upload_oracle_grid(mem, G)
    rays = np.concatenate([scene.make_rays_primary(G.bbox_min, G.bbox_max, 256, 128),
                           scene.make_rays_incoherent(G.bbox_min - 0.1, G.bbox_max + 0.1, 50001, 6)]).astype(np.float32)
    n = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    api.setup_traversal(grid)
    info = mem.image_format(grid)
    assert info["uniform"] and info["two_layouts"], info
    other = api.MemManager(keep=True)
    try:
        grid.mem = mem
        borrowed = api.share_traversal(other, grid)
        o_tris = other.upload(tris); o_rays = other.upload(rays); o_hits = other.alloc(16 * n)
        for binning in (0, 1):
            mem.set_ray_binning(binning); other.set_ray_binning(binning)
            for flags, oflags in ((0, 0), (api.ANY_HIT, O.ANY_HIT), (api.UVS, O.UVS), (api.ANY_HIT | api.UVS, O.ANY_HIT | O.UVS)):
                want = G.traverse_ex(tris, rays, oflags, nthreads=8)
                api.traverse_grid(grid, d_tris, d_rays, d_hits, n, flags)
                got = mem.download(d_hits, api.HIT_DTYPE, n)
                assert (got["id"] == want["id"]).all(), (binning, flags)
                for f in ("t", "u", "v"):
                    assert (bits(got[f]) == bits(want[f])).all(), (binning, flags, f)
                if flags in (0, api.UVS):                       # the borrower: its own buffers, the owner's image (both layouts)
                    api.traverse_grid(borrowed, o_tris, o_rays, o_hits, n, flags); other.synchronize()
                    got = other.download(o_hits, api.HIT_DTYPE, n)
                    assert (got["id"] == want["id"]).all() and (bits(got["t"]) == bits(want["t"])).all(), ("borrowed", binning, flags)
    finally:
        mem.set_ray_binning(0)
        other.close()
    mem.free(d_rays); mem.free(d_hits); grid.free(); mem.free(d_tris)


def test_any_hit_and_uvs_on_a_deep_clustered_grid(mem):
    """The table-layout image kernels (nested blocks, lists by index, the per-ray nested-block state) in their any-hit and
    barycentric variants, with and without the binning permutation: a small clustered scene with a grid deeper than three levels."""
    from oracle import oracle as O
    from hagrid_amd import api
    tris = scene.make_clustered(3000, 3, 4000)
    G = O.Grid.full(tris)
    assert G.shift > 3
    d_tris = mem.upload(tris)
    grid = upload_oracle_grid(mem, G)
    aimed = scene.make_rays_incoherent(G.bbox_min, G.bbox_max, 60000, 5).copy()
    k = np.arange(aimed.shape[0]) % 3
    centre = np.stack([0.17 + 0.14 * k, 0.32 + 0.08 * k, 0.22 + 0.1 * k], axis=1).astype(np.float32)
    aimed[:, 4:7] = centre - aimed[:, 0:3] + np.float32(0.03) * aimed[:, 4:7]
    rays = np.concatenate([scene.make_rays_primary(G.bbox_min, G.bbox_max, 256, 128), aimed,
                           scene.make_rays_incoherent(G.bbox_min - 0.1, G.bbox_max + 0.1, 30001, 6)]).astype(np.float32)
    n = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    api.setup_traversal(grid)
    try:
        for binning in (0, 1):
            mem.set_ray_binning(binning)
            for flags, oflags in ((0, 0), (api.ANY_HIT, O.ANY_HIT), (api.UVS, O.UVS), (api.ANY_HIT | api.UVS, O.ANY_HIT | O.UVS)):
                api.traverse_grid(grid, d_tris, d_rays, d_hits, n, flags)
                got = mem.download(d_hits, api.HIT_DTYPE, n)
                want = G.traverse_ex(tris, rays, oflags, nthreads=8)
                assert (got["id"] == want["id"]).all(), (binning, flags)
                for f in ("t", "u", "v"):
                    assert (bits(got[f]) == bits(want[f])).all(), (binning, flags, f)
                if flags == 0: assert (got["id"] >= 3000).sum() > 500          # rays that end inside a blob
    finally:
        mem.set_ray_binning(0)
    mem.free(d_rays); mem.free(d_hits); grid.free(); mem.free(d_tris)


def test_automatic_ray_binning(mem):
    """mode 2: identical hits whatever the device decides, and the decision is the expected one (observed through timing-free
    means: the permutation is only used for the unordered batch -- checked with the hits of a deliberately ordered copy)."""
    from oracle import oracle as O
    from hagrid_amd import api
    tris = scene.make_soup(60000, seed=31)
    G = O.Grid.full(tris)
    d_tris = mem.upload(tris)
    grid = upload_oracle_grid(mem, G)
    prim = scene.make_rays_primary(G.bbox_min, G.bbox_max, 256, 256)
    h0, _ = G.traverse(tris, prim, nthreads=8)
    bounce = scene.make_rays_bounce(tris, prim, h0, G.bbox_min, G.bbox_max, 77)
    batches = {"primary": prim, "bounce": bounce, "incoherent": scene.make_rays_incoherent(G.bbox_min, G.bbox_max, 100003, 12),
               "small": scene.make_rays_incoherent(G.bbox_min, G.bbox_max, 1000, 13)}
    try:
        mem.set_ray_binning(2)
        for name, rays in batches.items():
            rays = np.ascontiguousarray(rays, np.float32)
            want, _ = G.traverse(tris, rays, nthreads=8)
            for width in (0, -1):
                mem.set_option("traverse.image_width", width)
                for _ in range(2):          # twice: the coherence counters are reset on the device between batches
                    got = gpu_traverse(mem, grid, d_tris, rays)
                    assert (got["id"] == want["id"]).all() and (bits(got["t"]) == bits(want["t"])).all(), (name, width)
        with pytest.raises(api.HagridError):
            mem.set_ray_binning(3)
    finally:
        mem.set_ray_binning(0); mem.set_option("traverse.image_width", 0)
    grid.free(); mem.free(d_tris)


def test_wave_time_diagnostic_does_not_change_hits(mem):
    """hagrid_kat_traverse_timed (libhagrid_amd_kat.so): the stamped instantiations of the headline kernel and of the plain slim
    kernel give the product's hits, every wavefront has start <= end, and a tile order (here: reversed) only changes which wavefront
    takes which tile."""
    from hagrid_amd import api
    tris = scene.make_soup(30000, seed=3)
    d_tris = mem.upload(tris)
    grid = api.build_all(mem, d_tris, tris.shape[0])
    rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 256, 128)
    n = rays.shape[0]; nw = n // 64
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n); d_times = mem.alloc(16 * nw)
    api.setup_traversal(grid)
    assert mem.image_format(grid) == {"flat": True, "uniform": True, "general": False, "slim_id_bits": 20, "record_bytes": 16, "two_layouts": False}
    api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    ref = mem.download(d_hits, api.HIT_DTYPE, n)
    d_order = mem.upload(np.arange(nw, dtype=np.int32)[::-1].copy())
    for tail in (1, 0):
        for order in (None, d_order):
            mem.zero(d_times, 16 * nw); mem.zero(d_hits, 16 * n)
            api._check(mem, mem._K.hagrid_kat_traverse_timed(mem._ctx, C.byref(grid.pod), d_tris, d_rays, d_hits, n, 256, tail, d_times, order), "kat_traverse_timed")
            got = mem.download(d_hits, api.HIT_DTYPE, n)
            t = mem.download(d_times, np.uint64, 2 * nw).reshape(nw, 2)
            assert (got["id"] == ref["id"]).all() and (bits(got["t"]) == bits(ref["t"])).all()
            assert (t[:, 0] > 0).all() and (t[:, 1] >= t[:, 0]).all()
    mem.free(d_order); mem.free(d_times); mem.free(d_rays); mem.free(d_hits); grid.free(); mem.free(d_tris)


def test_release_for_traversal_keeps_hits_and_frees_the_construction_format():
    """Extension hagrid_grid_release_for_traversal: once setup_traversal has built the image (any of its three layouts), entries and cells
    go back to the pool, traversal (nearest hit, any-hit, barycentrics, binned) gives the same hits from the image alone; what
    needs the construction format is refused without harming the image."""
    from hagrid_amd import api
    mem = api.MemManager(keep=False)
    soup = scene.make_soup(200_000); clustered = scene.make_clustered(20000, 3, 30000)
    # every layout of the image: uniform (the soup), table (the soup at other densities), general (the clustered scene: five levels, wide records)
    for tris, params, compress, layout in ((soup, {}, False, "uniform"), (soup, {}, True, "uniform"), (soup, dict(top_density=0.15, snd_density=3.0), False, "table"),
                                           (clustered, {}, False, "general"), (clustered, {}, True, "general")):
        d_tris = mem.upload(tris)
        grid = api.build_all(mem, d_tris, tris.shape[0], compress=compress, **params)
        rays = np.concatenate([scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 256, 256),
                               scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 150_000, 21)]).astype(np.float32)
        n = rays.shape[0]
        d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
        mem.set_option("traverse.image_uniform", 0 if layout == "table" else 1)       # (a soup this small gets the uniform layout at either density: the table layout is asked for)
        api.setup_traversal(grid)
        mem.set_option("traverse.image_uniform", 1)
        fmt = mem.image_format(grid)
        assert (fmt["uniform"], fmt["general"]) == (layout == "uniform", layout == "general"), (layout, fmt)
        want = {}
        for flags in (0, api.ANY_HIT, api.UVS):
            api.traverse_grid(grid, d_tris, d_rays, d_hits, n, flags)
            want[flags] = mem.download(d_hits, api.HIT_DTYPE, n)
        before = mem.usage()
        held = 4 * grid.num_entries + (16 if compress else 32) * grid.num_cells
        api.release_for_traversal(grid)
        assert not grid.entries and not grid.cells and not grid.small_cells and grid.ref_ids
        assert before - mem.usage() >= held                       # keep = False: the buffers really went back
        for binning in (0, 1):
            mem.set_ray_binning(binning)
            for flags in (0, api.ANY_HIT, api.UVS):
                mem.zero(d_hits, 16 * n)
                api.traverse_grid(grid, d_tris, d_rays, d_hits, n, flags)
                got = mem.download(d_hits, api.HIT_DTYPE, n)
                assert got.tobytes() == want[flags].tobytes(), (layout, compress, binning, flags)
        mem.set_ray_binning(0)
        api.setup_traversal(grid)                                   # nothing to rebuild, nothing lost
        with pytest.raises(api.HagridError):
            api.traverse_grid_stats(grid, d_tris, d_rays, d_hits, n)
        with pytest.raises(api.HagridError):
            api.release_for_traversal(grid)                        # already released
        if not compress:
            with pytest.raises(api.HagridError):
                api.expand_grid(mem, grid, d_tris, 1)
        mem.set_option("traverse.variant", 2)
        with pytest.raises(api.HagridError):
            api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        mem.set_option("traverse.variant", 0)
        api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        assert mem.download(d_hits, api.HIT_DTYPE, n).tobytes() == want[0].tobytes()
        mem.free(d_rays); mem.free(d_hits); grid.free(); mem.free(d_tris)
    # without an image there is nothing that could stand for the construction format
    mem.set_option("traverse.image", 0)
    tris = soup; d_tris = mem.upload(tris)
    grid = api.build_all(mem, d_tris, tris.shape[0])
    api.setup_traversal(grid)
    with pytest.raises(api.HagridError):
        api.release_for_traversal(grid)
    mem.set_option("traverse.image", 2)
    grid.free(); mem.close()


def test_hit_id_can_carry_the_reference_kernels_step_count():
    """traverse.id_is_steps: Hit.id = the step count the reference kernel leaves there (traverse.cu:80,93: one per visited cell plus
    one per reference of its list), t unchanged; the oracle's step counter is the witness."""
    from hagrid_amd import api
    from oracle import oracle as O
    tris = scene.make_soup(20_000)
    mem = api.MemManager(keep=True)
    d_tris = mem.upload(tris)
    for compress in (False, True):
        grid = api.build_all(mem, d_tris, tris.shape[0], compress=compress)
        G = O.Grid.full(tris, compress=compress)
        rays = np.concatenate([scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 128, 128), scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 30_000, 9)]).astype(np.float32)
        n = rays.shape[0]
        d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
        api.setup_traversal(grid)
        api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        plain = mem.download(d_hits, api.HIT_DTYPE, n)
        mem.set_option("traverse.id_is_steps", 1)
        api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        stepped = mem.download(d_hits, api.HIT_DTYPE, n)
        mem.set_option("traverse.id_is_steps", 0)
        oh, _, osteps = G.traverse(tris, rays, want_steps=True)
        assert (stepped["id"] == osteps).all() and (stepped["t"].view(np.uint32) == plain["t"].view(np.uint32)).all()
        assert (plain["id"] == oh["id"]).all() and stepped["id"].max() > 3
        mem.free(d_rays); mem.free(d_hits); grid.free()
    mem.close()


def test_image_of_a_voxel_map_with_non_box_regions(mem):
    """A voxel map whose voxels of ONE cell do not form a box (hand-edited maps, grids uploaded from elsewhere): in many blocks the
    three voxels (0,0,0), (1,0,0), (0,1,0) are redirected to one cell -- an L.  The image (block layouts: a record per voxel; general layout: a record
    per entry) and the construction-format kernels must still agree with the oracle's traversal of the very same arrays."""
    from hagrid_amd import api
    from oracle import oracle as O
    tris = scene.make_soup(30_000)
    G0 = O.Grid.build(tris).flatten()                       # octree leaves, no merging, no expansion: every voxel its own cell
    entries = G0.entries.copy(); cells = G0.cells.copy(); refs = G0.ref_ids.copy()
    num_top = int(np.prod(G0.dims))
    edited = 0
    for T in range(num_top):
        w = int(entries[T]); d = w & 3
        if d == 0: continue
        blk = w >> 2
        block = entries[blk: blk + (1 << (3 * d))]
        if (block & 3).any(): continue                      # only blocks that resolve fully
        a = int(block[0]) >> 2
        k100, k010 = 1, 1 << d
        size = int(cells["max"][a][0] - cells["min"][a][0])
        if any(int(cells["max"][int(block[k]) >> 2][0] - cells["min"][int(block[k]) >> 2][0]) != size for k in (k100, k010)): continue
        # the L: two more voxels name cell a, whose box grows over the 2 x 2 x 1 corner; its list takes their references
        merged = np.unique(np.concatenate([refs[cells["begin"][c]: cells["end"][c]] for c in (a, int(block[k100]) >> 2, int(block[k010]) >> 2)]))
        cells["begin"][a] = refs.size; cells["end"][a] = refs.size + merged.size
        refs = np.concatenate([refs, merged.astype(np.int32)])
        cells["max"][a][0] += size; cells["max"][a][1] += size
        entries[blk + k100] = block[0]; entries[blk + k010] = block[0]
        edited += 1
    assert edited > 100
    G = O.Grid.from_arrays(entries, refs, cells, None, G0.bbox_min, G0.bbox_max, G0.dims, G0.shift, G0.offsets)
    rays = np.concatenate([scene.make_rays_primary(G0.bbox_min, G0.bbox_max, 256, 256), scene.make_rays_incoherent(G0.bbox_min, G0.bbox_max, 100_000, 8)]).astype(np.float32)
    want, _ = G.traverse(tris, rays, nthreads=8)
    d_tris = mem.upload(tris)
    grid = api.Grid.upload(mem, entries, refs, cells, None, G0.bbox_min, G0.bbox_max, G0.dims, G0.shift, G0.offsets)
    try:
        for fmt, variant, general in ((2, 4, 1), (2, 4, 2), (0, 2, 1), (0, 1, 1)):
            mem.set_option("traverse.image", fmt); mem.set_option("traverse.variant", variant); mem.set_option("traverse.image_general", general)
            got = gpu_traverse(mem, grid, d_tris, rays)
            assert (got["id"] == want["id"]).all() and (bits(got["t"]) == bits(want["t"])).all(), (fmt, variant, general)
    finally:
        mem.set_option("traverse.image", 2); mem.set_option("traverse.variant", 0); mem.set_option("traverse.image_general", 1)
        grid.free(); mem.free(d_tris)
