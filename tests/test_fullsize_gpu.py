"""Full BASELINE sizes on the GPU: the 1M-triangle scene (configs[1], [2]).  Structure parity against the
oracle (about 12 s of CPU), hit parity on the whole primary batch, and size-independent properties."""
import numpy as np
import pytest

from hagrid_amd import scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world():
    from hagrid_amd import api
    mem = api.MemManager(keep=True)
    tris = scene.make_soup(1_000_000)
    d_tris = mem.upload(tris)
    yield mem, tris, d_tris
    mem.close()


def traverse(mem, grid, d_tris, rays):
    from hagrid_amd import api
    n = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    api.setup_traversal(grid)
    api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    hits = mem.download(d_hits, api.HIT_DTYPE, n)
    mem.free(d_rays); mem.free(d_hits)
    return hits


def same_hits(a, b):
    return bool((a["id"] == b["id"]).all() and (a["t"].view(np.uint32) == b["t"].view(np.uint32)).all())


def test_config2_structure_and_hits_match_oracle(world):
    """configs[1]: soup-1M, defaults, 1M primary rays: grid arrays and every hit identical to the oracle."""
    from hagrid_amd import api
    from oracle import oracle as O
    mem, tris, d_tris = world
    grid = api.build_all(mem, d_tris, tris.shape[0])
    G = O.Grid.full(tris)
    d = grid.download()
    assert grid.summary() == G.summary()
    assert (d["entries"] == G.entries).all() and (d["ref_ids"] == G.ref_ids).all()
    assert d["cells"].tobytes() == G.cells.tobytes()
    rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 1024)
    hits = traverse(mem, grid, d_tris, rays)
    oh, ost = G.traverse(tris, rays, nthreads=8)
    assert same_hits(hits, oh)
    assert (hits["id"] >= 0).mean() > 0.7
    # the hit triangle really is hit at t (independent of any grid): re-intersect on the host
    idx = np.nonzero(hits["id"] >= 0)[0][::997]
    for i in idx[:200]:
        h = O.brute_force(tris[hits["id"][i]:hits["id"][i] + 1], rays[i:i + 1])
        assert h["id"][0] == 0 and h["t"].view(np.uint32)[0] == hits["t"].view(np.uint32)[i]
    grid.free()


def test_hits_do_not_depend_on_grid_parameters(world):
    """Size-independent property: the nearest hit is a property of (triangles, ray), not of the acceleration
    structure -- defaults, configs[2] parameters (0.15 / 3.0), no merge / no expansion and the compressed form
    must all report the same (id, t) for 4M incoherent rays."""
    from hagrid_amd import api
    mem, tris, d_tris = world
    variants = [dict(), dict(top_density=0.15, snd_density=3.0), dict(alpha=0.0, exp_iters=0), dict(compress=True)]
    rays = None; ref = None
    for kw in variants:
        grid = api.build_all(mem, d_tris, tris.shape[0], **kw)
        if rays is None:
            rays = scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 1 << 22, scene.RAY_SEED_BASE + 4)
        hits = traverse(mem, grid, d_tris, rays)
        if ref is None:
            ref = hits
            assert (hits["id"] >= 0).mean() > 0.5
        else:
            assert same_hits(hits, ref), kw
        grid.free()


def test_ray_order_independence_and_idempotence(world):
    from hagrid_amd import api
    mem, tris, d_tris = world
    grid = api.build_all(mem, d_tris, tris.shape[0])
    rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 512, 512)
    h1 = traverse(mem, grid, d_tris, rays)
    h2 = traverse(mem, grid, d_tris, rays)
    assert same_hits(h1, h2)
    perm = np.argsort(scene.uniform01(3, np.arange(rays.shape[0], dtype=np.uint64)), kind="stable")
    hp = traverse(mem, grid, d_tris, np.ascontiguousarray(rays[perm]))
    assert same_hits(hp, h1[perm])
    # sharding the batch (what a multi-GPU run does) gives the same hits as the whole batch
    for r in range(4):
        b, e = scene.shard_range(rays.shape[0], r, 4)
        assert same_hits(traverse(mem, grid, d_tris, np.ascontiguousarray(rays[b:e])), h1[b:e])
    grid.free()


def test_config5_8M_triangles_compressed_bounce_rays():
    """BASELINE configs[4]: soup-8M, defaults + --compress, diffuse-bounce rays generated from primary hits.
    The oracle cannot afford to BUILD this grid inside a test, so: (1) the compressed and the plain grid give the
    same hits, (2) the oracle's traversal of the GPU-built grid (downloaded) gives the same hits on a 200k-ray
    sample, (3) a brute force over all 8M triangles agrees on 1024 rays, (4) structural invariants hold."""
    import os
    from hagrid_amd import api
    from oracle import oracle as O
    n = 8_000_000
    mem = api.MemManager(keep=True)
    tris = scene.make_soup(n)
    d_tris = mem.upload(tris)
    plain = api.build_all(mem, d_tris, n)
    comp = api.build_all(mem, d_tris, n, compress=True)
    assert comp.small_cells and not comp.cells and comp.num_cells == plain.num_cells
    assert comp.num_refs > plain.num_refs                       # sentinels
    assert max(comp.dims) << comp.shift < 65536
    prim = scene.make_rays_primary(comp.bbox_min, comp.bbox_max, 1024, 1024)
    h0 = traverse(mem, comp, d_tris, prim)
    assert same_hits(h0, traverse(mem, plain, d_tris, prim))
    bounce = scene.make_rays_bounce(tris, prim, h0, comp.bbox_min, comp.bbox_max, scene.RAY_SEED_BASE + 5)
    assert np.isfinite(bounce).all() and (np.abs(bounce[:, 4:7]).sum(axis=1) > 0).all()
    hb = traverse(mem, comp, d_tris, bounce)
    assert same_hits(hb, traverse(mem, plain, d_tris, bounce))
    mem.set_ray_binning(1)
    assert same_hits(hb, traverse(mem, comp, d_tris, bounce))
    mem.set_ray_binning(0)
    assert 0.5 < (hb["id"] >= 0).mean() <= 1.0
    # self-hits are excluded by the 1e-4 offset along the normal: a bounce ray never re-hits its origin triangle at t ~ 0
    d = comp.download()
    G = O.Grid.from_arrays(d["entries"], d["ref_ids"], None, d["small_cells"], d["bbox_min"], d["bbox_max"], d["dims"], d["shift"], d["offsets"])
    rc, msg = G.check(tris, 0)
    assert rc == 0, msg
    cores = os.cpu_count() or 8
    sel = np.arange(0, bounce.shape[0], 5)[:200_000]
    oh, _ = G.traverse(tris, np.ascontiguousarray(bounce[sel]), nthreads=cores)
    assert same_hits(hb[sel], oh)
    sel = np.arange(0, bounce.shape[0], 1021)[:1024]
    bf = O.brute_force(tris, np.ascontiguousarray(bounce[sel]), nthreads=cores)
    assert same_hits(hb[sel], bf)
    plain.free(); comp.free(); mem.close()


def test_clustered_scene_structure_and_hits_match_oracle():
    """A very non-uniform 1M-triangle scene (scene.make_clustered: six dense blobs in a sparse soup; grid shift 6, lists of
    up to ~20 references): grid arrays identical to the oracle's; primary and incoherent hits identical to the oracle's with
    the construction-format kernel and with both image formats (nested blocks, by-index lists)."""
    from hagrid_amd import api
    from oracle import oracle as O
    tris = scene.make_clustered()
    mem = api.MemManager(keep=True)
    try:
        d_tris = mem.upload(tris)
        grid = api.build_all(mem, d_tris, tris.shape[0])
        G = O.Grid.full(tris)
        d = grid.download()
        assert grid.summary() == G.summary() and grid.summary()["shift"] > 3
        assert (d["entries"] == G.entries).all() and (d["ref_ids"] == G.ref_ids).all()
        assert d["cells"].tobytes() == G.cells.tobytes()
        aimed = scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 1 << 17, 12).copy()   # towards the blobs, with some spread
        k = np.arange(aimed.shape[0]) % 6
        centre = np.stack([0.17 + 0.14 * k, 0.32 + 0.08 * k, 0.22 + 0.1 * k], axis=1).astype(np.float32)
        aimed[:, 4:7] = centre - aimed[:, 0:3] + np.float32(0.02) * aimed[:, 4:7]
        rays = np.concatenate([scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 512),
                               scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 1 << 18, 11), aimed]).astype(np.float32)
        oh, _ = G.traverse(tris, rays, nthreads=8)
        assert (oh["id"] >= 1_00_000).sum() > 1000            # rays that end inside a blob
        for image in (2, 1, 0):
            mem.set_option("traverse.image", image)
            assert same_hits(traverse(mem, grid, d_tris, rays), oh), f"traverse.image={image}"
        mem.set_option("traverse.image", 2)
        grid.free()
    finally:
        mem.close()
