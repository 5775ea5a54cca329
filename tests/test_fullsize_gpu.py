"""Full BASELINE sizes on the GPU, every configuration of BASELINE.json at its stated size (multi-GPU configurations with
the per-GPU share of an 8-GPU run): structure parity against the oracle (about 12 s of CPU per 1M-triangle grid), hit
parity against the oracle / a brute force on samples the CPU can afford, identical hits across every traversal path,
and size-independent properties."""
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


_ORACLE_GRIDS = {}


def oracle_grid(tris, *params):
    """the oracle's construction of the 1M-triangle grid (about 12 s of CPU), built once per parameter set"""
    from oracle import oracle as O
    if params not in _ORACLE_GRIDS:
        _ORACLE_GRIDS[params] = O.Grid.full(tris, *params)
    return _ORACLE_GRIDS[params]


def same_hits(a, b):
    return bool((a["id"] == b["id"]).all() and (a["t"].view(np.uint32) == b["t"].view(np.uint32)).all())


def test_config2_structure_and_hits_match_oracle(world):
    """configs[1]: soup-1M, defaults, 1M primary rays: grid arrays and every hit identical to the oracle."""
    from hagrid_amd import api
    from oracle import oracle as O
    mem, tris, d_tris = world
    grid = api.build_all(mem, d_tris, tris.shape[0])
    G = oracle_grid(tris)
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


def test_config2_loop_over_one_buffer_learned_tile_order_matches_oracle(world):
    """configs[1] as the bench line measures it: ONE 1024 x 1024 ray buffer traversed 40 times (main.cpp:398-447, the reference's benchmark
    loop).  Launch 1 runs in the default tile order, launches 2 and 3 in the orders sorted behind launches 1 and 2, launch 33 behind the
    first periodic refresh, launch 40 in the steady state: the hits of each of them are the oracle's, bit for bit.  Then the buffer is
    refilled with another image (flipped top to bottom) -- the launches over it start on the stale order or on the default one, whichever the
    context decides: same requirement."""
    from hagrid_amd import api
    mem, tris, d_tris = world
    grid = api.build_all(mem, d_tris, tris.shape[0])
    G = oracle_grid(tris)
    assert grid.summary() == G.summary()
    w = h = 1024
    rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, w, h)
    n = rays.shape[0]
    oh, _ = G.traverse(tris, rays, nthreads=8)
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    api.setup_traversal(grid)
    fmt = mem.image_format(grid)
    assert fmt.get("slim_id_bits"), fmt                              # the tail kernel with its tile order is what runs here
    checked = []
    for launch in range(1, 41):
        mem.zero(d_hits, 16 * n)
        api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        if launch in (1, 2, 3, 33, 40):
            mem.synchronize()                                        # (the host sees the row length: the order is learned from the next launch on)
            assert same_hits(mem.download(d_hits, api.HIT_DTYPE, n), oh), f"launch {launch}"
            checked.append(launch)
    assert checked == [1, 2, 3, 33, 40]
    flipped = np.ascontiguousarray(rays.reshape(h, w, 8)[::-1].reshape(n, 8))
    oh2 = np.ascontiguousarray(oh.reshape(h, w)[::-1].reshape(n))
    mem.copy_h2d(d_rays, flipped)
    for launch in range(1, 6):
        mem.zero(d_hits, 16 * n)
        api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        mem.synchronize()
        assert same_hits(mem.download(d_hits, api.HIT_DTYPE, n), oh2), f"refilled buffer, launch {launch}"
    mem.free(d_rays); mem.free(d_hits); grid.free()


def test_config3_dense_grid_16M_primary_rays(world):
    """configs[2]: soup-1M, --top-density 0.15 --snd-density 3.0 --expansion 3, 4096 x 4096 primary rays.  Grid arrays identical
    to the oracle's; identical hits from the table layout of the image, the general layout and the construction format; the oracle on a strided
    1M-ray sample; a brute force over all triangles on 1024 rays."""
    import os
    from hagrid_amd import api
    from oracle import oracle as O
    mem, tris, d_tris = world
    cores = os.cpu_count() or 8
    grid = api.build_all(mem, d_tris, tris.shape[0], top_density=0.15, snd_density=3.0, exp_iters=3)
    G = O.Grid.full(tris, 0.15, 3.0, 0.995, 3)
    d = grid.download()
    assert grid.summary() == G.summary()
    assert (d["entries"] == G.entries).all() and (d["ref_ids"] == G.ref_ids).all() and d["cells"].tobytes() == G.cells.tobytes()
    w = h = 4096
    rays = scene.generate_parallel(lambda f, c: scene.make_rays_primary(grid.bbox_min, grid.bbox_max, w, h, first=f, count=c), 0, w * h, chunk=1 << 21)
    n = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    got = {}
    # the uniform layout with the table layout next to it (this grid's default since round 6: rays in image order gather from the uniform one), the table layout alone,
    # the general layout forced, the construction format
    for key, image, general, uniform in (("default", 2, 1, 1), ("table", 2, 1, 0), (1, 1, 2, 1), (0, 0, 1, 1)):
        mem.set_option("traverse.image", image); mem.set_option("traverse.image_general", general); mem.set_option("traverse.image_uniform", uniform)
        api.setup_traversal(grid)
        f = mem.image_format(grid)
        if image: assert f["general"] == (general == 2) and f["uniform"] == (key == "default") and f["two_layouts"] == (key == "default"), (key, f)
        api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        got[key] = mem.download(d_hits, api.HIT_DTYPE, n)
    mem.set_option("traverse.image", 2); mem.set_option("traverse.image_general", 1); mem.set_option("traverse.image_uniform", 1)
    got[2] = got["default"]
    assert same_hits(got[2], got[1]) and same_hits(got[2], got[0]) and same_hits(got[2], got["table"])
    hits = got[2]
    assert (hits["id"] >= 0).mean() > 0.7
    sel = np.arange(7, n, 16)                                   # 1M rays spread over the whole image
    oh, _ = G.traverse(tris, np.ascontiguousarray(rays[sel]), nthreads=cores)
    assert same_hits(hits[sel], oh)
    sel = np.arange(3, n, n // 1024)[:1024]
    assert same_hits(hits[sel], O.brute_force(tris, np.ascontiguousarray(rays[sel]), nthreads=cores))
    mem.free(d_rays); mem.free(d_hits); grid.free()


def test_config4_share_16M_incoherent_rays_binned_and_unbinned(world):
    """configs[3], the per-GPU share of the 8-GPU run: rays [3 * 2^24, 4 * 2^24) of the 128M incoherent batch on the default
    grid.  Ray binning off / on / automatic give identical hits (large-batch kernel choice, binning above 300k rays); the oracle
    on a strided 1M-ray sample."""
    import os
    from hagrid_amd import api
    from oracle import oracle as O
    mem, tris, d_tris = world
    cores = os.cpu_count() or 8
    grid = api.build_all(mem, d_tris, tris.shape[0])
    n = 1 << 24
    first, end = scene.shard_range(1 << 27, 3, 8)
    assert end - first == n
    rays = scene.generate_parallel(lambda f, c: scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, c, scene.RAY_SEED_BASE + 4, first=f), first, n)
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    api.setup_traversal(grid)
    got = {}
    for mode in (0, 1, 2):
        mem.set_ray_binning(mode)
        api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        got[mode] = mem.download(d_hits, api.HIT_DTYPE, n)
    mem.set_ray_binning(0)
    assert same_hits(got[0], got[1]) and same_hits(got[0], got[2])
    # without the traversal image the batch takes the persistent large-batch kernel (unbinned) and v2 (binned)
    mem.set_option("traverse.image", 0)
    api.setup_traversal(grid)
    for mode in (0, 1):
        mem.set_ray_binning(mode)
        api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        assert same_hits(got[0], mem.download(d_hits, api.HIT_DTYPE, n)), f"construction format, binning {mode}"
    mem.set_ray_binning(0); mem.set_option("traverse.image", 2)
    hits = got[0]
    assert 0.5 < (hits["id"] >= 0).mean() <= 1.0
    d = grid.download()
    G = O.Grid.from_arrays(d["entries"], d["ref_ids"], d["cells"], None, d["bbox_min"], d["bbox_max"], d["dims"], d["shift"], d["offsets"])
    sel = np.arange(5, n, 16)
    oh, _ = G.traverse(tris, np.ascontiguousarray(rays[sel]), nthreads=cores)
    assert same_hits(hits[sel], oh)
    mem.free(d_rays); mem.free(d_hits); grid.free()


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
    plain.free()

    # the configuration at its size: the per-GPU share of the 64M-ray batch = rows 3072..4095 of the 8192 x 8192 primary image
    # (8 388 608 rays), bounce rays in the image order of their primary rays.  The >= 4M-ray origin criterion of the row
    # detection and the compressed image kernel run at size; tile packets on (detected), off, and with the width given, ray
    # binning, and the construction format must all give the same hits; the oracle checks a strided 200k-ray sample.
    W = 8192
    first, end = scene.shard_range(W * W, 3, 8)
    n8 = end - first
    assert n8 == 8_388_608 and first % W == 0
    prim = scene.generate_parallel(lambda f, c: scene.make_rays_primary(comp.bbox_min, comp.bbox_max, W, W, first=f, count=c), first, n8, chunk=1 << 21)
    h0 = traverse(mem, comp, d_tris, prim)
    bounce = np.empty_like(prim)
    for off in range(0, n8, 1 << 21):
        sl = slice(off, min(off + (1 << 21), n8))
        bounce[sl] = scene.make_rays_bounce(tris, prim[sl], h0[sl], comp.bbox_min, comp.bbox_max, scene.RAY_SEED_BASE + 5, first=first + off)
    del prim
    d_rays = mem.upload(bounce); d_hits = mem.alloc(16 * n8)
    api.setup_traversal(comp)
    variants = {}
    for name, opts in (("detect", dict(width=0)), ("off", dict(width=-1)), ("given", dict(width=W)), ("binned", dict(width=0, bin=1)),
                       ("construction format", dict(width=0, image=0))):
        mem.set_option("traverse.image_width", opts["width"]); mem.set_ray_binning(opts.get("bin", 0))
        if "image" in opts:
            mem.set_option("traverse.image", opts["image"]); api.setup_traversal(comp)
        api.traverse_grid(comp, d_tris, d_rays, d_hits, n8)
        variants[name] = mem.download(d_hits, api.HIT_DTYPE, n8)
    mem.set_option("traverse.image_width", 0); mem.set_ray_binning(0); mem.set_option("traverse.image", 2)
    for name, v in variants.items():
        assert same_hits(variants["detect"], v), name
    hb8 = variants["detect"]
    assert 0.5 < (hb8["id"] >= 0).mean() <= 1.0
    sel = np.arange(11, n8, 41)[:200_000]
    oh, _ = G.traverse(tris, np.ascontiguousarray(bounce[sel]), nthreads=cores)
    assert same_hits(hb8[sel], oh)
    comp.free(); mem.close()


def test_clustered_scene_structure_and_hits_match_oracle():
    """A very non-uniform 1M-triangle scene (scene.make_clustered: six dense blobs in a sparse soup; grid shift 6, lists of
    up to ~20 references): grid arrays identical to the oracle's; primary and incoherent hits identical to the oracle's with
    the construction-format kernel and with the traversal image (general layout of slim records: links, wide records, by-index lists)."""
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
        for image in (2, 0):
            mem.set_option("traverse.image", image)
            assert same_hits(traverse(mem, grid, d_tris, rays), oh), f"traverse.image={image}"
        # the image of a deep grid is the general layout of slim records, traversed by the tail kernel; the same image by the kernel without the
        # tail mode, with ray binning, in the 26-bit id form
        mem.set_option("traverse.image", 2); api.setup_traversal(grid)
        fmt = mem.image_format(grid)
        assert fmt["general"] and fmt["record_bytes"] == 16 and fmt["slim_id_bits"] == 20, fmt
        for opts in ({"traverse.tail": 0}, {"binning": 1}, {"traverse.image_slim": 2}):
            for k, v in opts.items():
                if k == "binning": mem.set_ray_binning(v)
                else: mem.set_option(k, v)
            assert same_hits(traverse(mem, grid, d_tris, rays), oh), opts
            mem.set_ray_binning(0); mem.set_option("traverse.tail", 1); mem.set_option("traverse.image_slim", 1)
        # ONE 1024 x 1024 buffer traversed 120 times: the tile order is learned, and -- the blobs' tiles cost several times the median tile -- the longest tiles of the
        # order start with four lanes per ray, first ("traverse.quad_head": the share a sort suggests is taken up by a later sort, trav_kernels.h a.quad_head).  Every
        # checked launch gives the oracle's hits; with a lower threshold (more tiles at the head) and without the feature as well.
        prim = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 1024); n = prim.shape[0]
        ohp, _ = G.traverse(tris, prim, nthreads=8)
        d_rays = mem.upload(prim); d_hits = mem.alloc(16 * n)
        for head in (20, 12, 0):
            mem.set_option("traverse.quad_head", head)
            for launch in range(1, 121):
                if launch in (1, 2, 3, 34, 35, 67, 68, 100, 120): mem.zero(d_hits, 16 * n)
                api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
                if launch in (1, 2, 3, 34, 35, 67, 68, 100, 120):
                    mem.synchronize()
                    assert same_hits(mem.download(d_hits, api.HIT_DTYPE, n), ohp), (head, launch)
        mem.set_option("traverse.quad_head", 20)
        # A camera that MOVES (the reference's viewer writes new rays every frame, main.cpp:591-601): eight frames at the viewer's speed into the same buffer, behind the
        # launches above (a learned order that goes stale, the pause from learning, the share trial of launches in the default order -- half of the tiles with four
        # lanes per ray, measured against the rule's share): every frame gives the oracle's hits.
        moving_camera_frames(mem, grid, d_tris, G, tris, d_rays, d_hits, 1024, 1024)
        mem.free(d_rays); mem.free(d_hits)
        grid.free()
    finally:
        mem.close()


def moving_camera_frames(mem, grid, d_tris, G, tris, d_rays, d_hits, width, height, frames=8, extra_launches=3):
    """`frames` frames of a camera that turns and strafes at the reference viewer's speed, written into ONE ray buffer; every frame is traversed a few times (the
    policy's trials take their samples over launches) and its hits compared with the oracle's."""
    from hagrid_amd import api
    n = width * height
    for f in range(frames):
        r = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, width, height, yaw=0.005 * (f + 1), strafe=0.005 * (f + 1))
        want, _ = G.traverse(tris, r, nthreads=8)
        mem.copy_h2d(d_rays, r)
        for k in range(extra_launches):
            mem.zero(d_hits, 16 * n)
            api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
            mem.synchronize()
            assert same_hits(mem.download(d_hits, api.HIT_DTYPE, n), want), (f, k, mem.order_state(d_rays))


def test_stadium_mesh_through_the_front_door(tmp_path):
    """A mesh-shaped scene (scene.make_stadium_mesh: tori and spheres with shared vertices, a grain of dust, inside a hall of ten huge triangles -- edges over four
    orders of magnitude, "teapot in a stadium", the reference README's motivation) through the FRONT DOOR: written as OBJ, read by hagrid_cli with
    include/hagrid/load_obj.h (main.cpp:246-275, load_obj.cpp:78-239), built, traced from a .rays file, saved with --save-grid.  The CLI's triangle, cell and
    reference counts and its intersection count are the oracle's; the grid file it wrote holds the oracle's arrays and the mesh's Tri records bit for bit; the
    Python API builds the same arrays; primary, incoherent and dust-aimed rays give the oracle's hits on the image and on the construction format; eight frames
    of a moving camera as well."""
    import os, subprocess, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _subproc
    from test_cpp_api import _build_cli
    from hagrid_amd import api
    from oracle import oracle as O
    V, F = scene.make_stadium_mesh()
    tris = scene.tris_from_mesh(V, F)
    G = O.Grid.full(tris)
    assert G.shift >= 5 and int((G.cells["end"] - G.cells["begin"]).max()) > 100        # a deep grid with long lists (the dust)
    mem = api.MemManager(keep=True)
    try:
        d_tris = mem.upload(tris)
        grid = api.build_all(mem, d_tris, tris.shape[0])
        d = grid.download()
        assert grid.summary() == G.summary()
        assert (d["entries"] == G.entries).all() and (d["ref_ids"] == G.ref_ids).all() and d["cells"].tobytes() == G.cells.tobytes()
        lo, hi = grid.bbox_min, grid.bbox_max
        dust = scene.make_rays_incoherent(lo, hi, 1 << 16, 21).copy()
        dust[:, 4:7] = np.float32([0.50, 0.02, 0.30]) - dust[:, 0:3] + np.float32(0.004) * dust[:, 4:7]       # towards the grain of dust (lists of hundreds of ids)
        rays = np.ascontiguousarray(np.concatenate([scene.make_rays_primary(lo, hi, 1024, 512), scene.make_rays_incoherent(lo, hi, 1 << 18, 11), dust]).astype(np.float32))
        rays[:, 3] = 0.0; rays[:, 7] = scene.FLT_MAX
        want, _ = G.traverse(tris, rays, nthreads=8)
        assert (want["id"] >= 0).mean() > 0.5                                # (a 2:1 image from outside sees past the hall; from inside only the open front lets a ray out)
        n_dust = tris.shape[0] - 3 * (2 * 16 * 6 + 2 * 16)                   # (the three coarse lamps are the last objects, the dust is just before them)
        assert ((want["id"] < n_dust) & (want["id"] >= n_dust - 39000)).sum() > 100, "no ray ends on the grain of dust"
        for image in (2, 0):
            mem.set_option("traverse.image", image)
            assert same_hits(traverse(mem, grid, d_tris, rays), want), f"traverse.image={image}"
        mem.set_option("traverse.image", 2); api.setup_traversal(grid)
        assert mem.image_format(grid)["general"]
        # the front door
        obj = str(tmp_path / "stadium.obj"); rfile = str(tmp_path / "stadium.rays"); gfile = str(tmp_path / "stadium.grid")
        scene.write_obj(obj, V, F)
        np.ascontiguousarray(rays[:, [0, 1, 2, 4, 5, 6]]).tofile(rfile)
        exe = _build_cli(str(tmp_path))
        r = _subproc.check([exe, obj, "-r", rfile, "-n", "3", "-w", "1", "-k", "-nb", "2", "--save-grid", gfile], timeout=300)
        assert f"{tris.shape[0]} triangle(s)" in r.stdout and f"{G.num_cells} cells, {G.num_refs} references)" in r.stdout, r.stdout
        assert f"{int((want['id'] >= 0).sum())} intersection(s)." in r.stdout, r.stdout
        g2, d_tris2, n2 = api.Grid.load(mem, gfile)
        assert n2 == tris.shape[0] and mem.download(d_tris2, np.float32, 12 * n2).tobytes() == tris.tobytes()
        d2 = g2.download()
        assert g2.summary() == G.summary() and (d2["entries"] == G.entries).all() and (d2["ref_ids"] == G.ref_ids).all() and d2["cells"].tobytes() == G.cells.tobytes()
        g2.free(); mem.free(d_tris2)
        # a loop over one buffer, then a camera that moves
        prim = scene.make_rays_primary(lo, hi, 1024, 1024); n = prim.shape[0]
        wp, _ = G.traverse(tris, prim, nthreads=8)
        d_rays = mem.upload(prim); d_hits = mem.alloc(16 * n)
        for launch in range(1, 80):
            if launch in (1, 2, 3, 34, 35, 79): mem.zero(d_hits, 16 * n)
            api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
            if launch in (1, 2, 3, 34, 35, 79):
                mem.synchronize()
                assert same_hits(mem.download(d_hits, api.HIT_DTYPE, n), wp), launch
        moving_camera_frames(mem, grid, d_tris, G, tris, d_rays, d_hits, 1024, 1024)
        mem.free(d_rays); mem.free(d_hits); grid.free()
    finally:
        mem.close()
