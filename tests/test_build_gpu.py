"""GPU parity of the construction passes against the CPU oracle: after every pass the device grid
(entries, cells, ref_ids, dims, bbox, shift, offsets, counts) must equal the oracle's bit for bit."""
import numpy as np
import pytest

from hagrid_amd import scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mem():
    from hagrid_amd import api
    m = api.MemManager(keep=True)
    yield m
    m.close()


def assert_same_grid(dev: dict, G, stage: str):
    assert tuple(dev["dims"]) == tuple(G.dims), stage
    assert dev["shift"] == G.shift, stage
    assert list(dev["offsets"]) == list(G.offsets), (stage, dev["offsets"], G.offsets)
    assert (dev["bbox_min"].view(np.uint32) == G.bbox_min.view(np.uint32)).all(), stage
    assert (dev["bbox_max"].view(np.uint32) == G.bbox_max.view(np.uint32)).all(), stage
    assert dev["entries"].shape == G.entries.shape and (dev["entries"] == G.entries).all(), stage
    if G.cells is not None:
        assert dev["cells"] is not None and dev["cells"].shape == G.cells.shape, stage
        for f in ("min", "max", "begin", "end"):
            assert (dev["cells"][f] == G.cells[f]).all(), (stage, f)
    else:
        assert dev["small_cells"] is not None and dev["small_cells"].shape == G.small_cells.shape, stage
        for f in ("min", "max", "begin"):
            assert (dev["small_cells"][f] == G.small_cells[f]).all(), (stage, f)
    assert dev["ref_ids"].shape == G.ref_ids.shape and (dev["ref_ids"] == G.ref_ids).all(), stage


def run_stages(mem, tris, td=0.12, sd=2.4, alpha=0.995, exp=3, compress=True):
    from hagrid_amd import api
    from oracle import oracle as O
    d_tris = mem.upload(tris)
    n = tris.shape[0]
    grid = api.Grid()
    api.build_grid(mem, d_tris, n, grid, td, sd)
    G = O.Grid.build(tris, td, sd)
    assert_same_grid(grid.download(), G, "build")
    api.merge_grid(mem, grid, alpha); G.merge(alpha)
    assert_same_grid(grid.download(), G, "merge")
    api.flatten_grid(mem, grid); G.flatten()
    assert_same_grid(grid.download(), G, "flatten")
    api.expand_grid(mem, grid, d_tris, exp); G.expand(tris, exp)
    assert_same_grid(grid.download(), G, "expand")
    if compress:
        ok = api.compress_grid(mem, grid)
        assert ok == G.compress()
        assert_same_grid(grid.download(), G, "compress")
    return grid, G, d_tris


def test_build_config1_every_stage(mem):
    """BASELINE config 1: soup-10k, defaults."""
    tris = scene.make_soup(10000)
    grid, G, d_tris = run_stages(mem, tris)
    assert grid.summary() == G.summary()
    grid.free(); mem.free(d_tris)


def test_merge_with_wide_working_cells(mem):
    """merge.narrow_cells = 0 keeps the 32-byte record between the passes (the path of virtual resolutions >= 65536)."""
    from hagrid_amd import api
    from oracle import oracle as O
    tris = scene.make_soup(30000, seed=91)
    d_tris = mem.upload(tris)
    G = O.Grid.build(tris).merge(0.995)
    try:
        for narrow in (0, 1):
            mem.set_option("merge.narrow_cells", narrow)
            grid = api.Grid()
            api.build_grid(mem, d_tris, tris.shape[0], grid, 0.12, 2.4)
            api.merge_grid(mem, grid, 0.995)
            assert_same_grid(grid.download(), G, ("merge", narrow))
            grid.free()
    finally:
        mem.set_option("merge.narrow_cells", 1)
    mem.free(d_tris)


@pytest.mark.parametrize("n,seed,td,sd,alpha", [(30000, 91, 0.12, 2.4, 0.995), (60000, 5, 0.12, 2.4, 0.9999), (20000, 6, 0.5, 8.0, 0.9999), (4000, 7, 0.3, 1.0, 0.99999),
                                                (200000, 8, 0.12, 2.4, 0.999)])
def test_merge_iterations_in_place_and_compacting_agree_with_the_oracle(mem, n, seed, td, sd, alpha):
    """merge_grid runs its late iterations in place (dirty cells only, tombstones, one compaction at the end; merge.hip).  The arrays must be
    the oracle's whatever the mode: in place (default), compacting throughout (merge.inplace = 0), and leaving the mode after every in-place
    iteration so that compacting and in-place iterations alternate (merge.inplace_iters = 1).  alpha close to 1 runs many iterations:
    beyond the fourth the mask of merge.cu:361 drops to 0 and every cell has to look again."""
    from hagrid_amd import api
    from oracle import oracle as O
    tris = scene.make_soup(n, seed=seed)
    d_tris = mem.upload(tris)
    G = O.Grid.build(tris, td, sd).merge(alpha)
    try:
        # room: the in-place mode appends its merged lists behind the live references of the same buffer; a pass whose lists do not fit is not
        # applied and the rest of its iteration compacts -- forced here by capping the buffer (1: the very first pass; final size + a little: a later one)
        # div: the mode is entered once an iteration merges less than 1 / div of its cells (default 8); 1 enters behind the first iteration, where the
        # stamps of the compacting passes make most cells dirty, 2 / 4 somewhere in between
        for inplace, iters, room, div in ((1, 0, 0, 0), (0, 0, 0, 0), (1, 0, 0, 1), (1, 0, 0, 2), (1, 1, 0, 1), (1, 2, 0, 4), (1, 0, 1, 1), (1, 0, G.num_refs + 64, 1),
                                          (1, 0, G.num_refs + G.num_refs // 16, 1)):
            mem.set_option("merge.inplace", inplace); mem.set_option("merge.inplace_iters", iters); mem.set_option("merge.inplace_room", room)
            mem.set_option("merge.inplace_div", div)
            grid = api.Grid()
            api.build_grid(mem, d_tris, n, grid, td, sd)
            api.merge_grid(mem, grid, alpha)
            assert_same_grid(grid.download(), G, ("merge", inplace, iters, room, div))
            bc = mem.build_counts()
            assert bc["merged_cells"] == G.num_cells and bc["merged_refs"] == G.num_refs
            grid.free()
    finally:
        mem.set_option("merge.inplace", 1); mem.set_option("merge.inplace_iters", 0); mem.set_option("merge.inplace_room", 0); mem.set_option("merge.inplace_div", 0)
    mem.free(d_tris)


@pytest.mark.parametrize("n,td,sd", [(1, 0.12, 2.4), (2, 0.12, 2.4), (37, 0.12, 2.4), (3000, 0.15, 3.0), (50000, 0.12, 2.4), (20000, 0.5, 8.0)])
def test_build_sizes_and_densities(mem, n, td, sd):
    tris = scene.make_soup(n, seed=1234 + n)
    grid, G, d_tris = run_stages(mem, tris, td, sd)
    grid.free(); mem.free(d_tris)


def test_build_clustered_scene_long_lists(mem):
    """Teapot-in-a-stadium: a dense cluster (long per-cell lists, deep levels) inside a sparse soup."""
    big = scene.make_soup(2000, seed=7)
    small = scene.make_soup(6000, seed=8).copy()
    small[:, 0:3] = small[:, 0:3] * np.float32(0.01) + np.float32(0.4)      # v0 into a 1 % box
    small[:, 4:7] *= np.float32(0.002); small[:, 8:11] *= np.float32(0.002)   # tiny edges
    e1, e2 = small[:, 4:7], small[:, 8:11]
    nrm = np.stack([e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1], e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2], e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]], axis=1).astype(np.float32)
    small[:, 3] = nrm[:, 0]; small[:, 7] = nrm[:, 1]; small[:, 11] = nrm[:, 2]
    tris = np.concatenate([big, small]).astype(np.float32)
    grid, G, d_tris = run_stages(mem, tris)
    grid.free(); mem.free(d_tris)


def test_build_then_traverse_matches_reference_bruteforce(mem, golden_dir):
    """End to end on the GPU only (build + traverse), against the reference-arithmetic brute force."""
    import os
    from hagrid_amd import api
    g = np.load(os.path.join(golden_dir, "config1_hits.npz"))
    tris = scene.make_soup(10000)
    lo, hi = scene.tris_bbox(tris)
    rays = scene.make_rays_incoherent(lo, hi, 65536, scene.RAY_SEED_BASE + 1)
    d_tris = mem.upload(tris)
    for compress in (False, True):
        grid = api.build_all(mem, d_tris, tris.shape[0], compress=compress)
        d_rays = mem.upload(rays); d_hits = mem.alloc(16 * rays.shape[0])
        api.setup_traversal(grid)
        api.traverse_grid(grid, d_tris, d_rays, d_hits, rays.shape[0])
        hits = mem.download(d_hits, api.HIT_DTYPE, rays.shape[0])
        assert (hits["id"] == g["id"]).all()
        assert (hits["t"].view(np.uint32) == g["t"].view(np.uint32)).all()
        mem.free(d_rays); mem.free(d_hits); grid.free()
    mem.free(d_tris)


def test_rebuild_in_keep_mode_is_stable(mem):
    """main.cpp:481-508: repeated builds with freed grid arrays give identical grids; pool does not grow."""
    from hagrid_amd import api
    tris = scene.make_soup(20000)
    d_tris = mem.upload(tris)
    ref = None; usage = []
    for it in range(3):
        grid = api.build_all(mem, d_tris, tris.shape[0])
        d = grid.download()
        if ref is None:
            ref = d
        else:
            assert (d["entries"] == ref["entries"]).all() and (d["ref_ids"] == ref["ref_ids"]).all()
            assert d["cells"].tobytes() == ref["cells"].tobytes()
        grid.free()
        usage.append(mem.usage())
    assert usage[2] <= usage[1] * 1.05
    mem.free(d_tris)


def test_compress_refuses_fine_grids(mem):
    from hagrid_amd import api
    tris = scene.make_soup(64)
    d_tris = mem.upload(tris)
    grid = api.build_all(mem, d_tris, tris.shape[0])
    grid.pod.shift = 15                                       # pretend: dims << 15 >= 65536
    cells_before = grid.cells
    assert api.compress_grid(mem, grid) is False and grid.cells == cells_before and not grid.small_cells
    grid.pod.shift = 0
    mem.free(d_tris)


def test_build_rejects_bad_input(mem):
    from hagrid_amd import api
    with pytest.raises(api.HagridError):
        api.build_grid(mem, 0, 0, api.Grid(), 0.12, 2.4)


def test_build_scene_with_huge_triangles(mem):
    """Teapot-in-a-stadium: a soup standing on a ground plane and inside four walls whose triangles cover thousands of
    top-level cells each (the wave-cooperative emission path, build.cu:106-135 in the reference)."""
    soup = scene.make_soup(30000, seed=99)
    lo, hi = np.float32([-2, -0.05, -2]), np.float32([3, 1.5, 3])
    def quad(a, b, c, d):
        a, b, c, d = (np.float32(v)[None, :] for v in (a, b, c, d))
        return np.concatenate([scene.tris_from_vertices(a, b, c), scene.tris_from_vertices(a, c, d)])
    walls = [quad([lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], lo[1], hi[2]], [lo[0], lo[1], hi[2]]),       # ground
             quad([lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]]),
             quad([lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]),
             quad([lo[0], lo[1], lo[2]], [lo[0], lo[1], hi[2]], [lo[0], hi[1], hi[2]], [lo[0], hi[1], lo[2]]),
             quad([hi[0], lo[1], lo[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [hi[0], hi[1], lo[2]]),
             quad([lo[0], 0.4, lo[2]], [hi[0], 0.9, lo[2]], [hi[0], 0.2, hi[2]], [lo[0], 0.7, hi[2]])]               # a slanted sheet
    tris = np.concatenate([soup] + walls).astype(np.float32)
    grid, G, d_tris = run_stages(mem, tris)
    rays = scene.make_rays_incoherent(G.bbox_min, G.bbox_max, 100000, 3)
    from hagrid_amd import api
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * rays.shape[0])
    api.traverse_grid(grid, d_tris, d_rays, d_hits, rays.shape[0])
    hits = mem.download(d_hits, api.HIT_DTYPE, rays.shape[0])
    want, _ = G.traverse(tris, rays, nthreads=8)
    assert (hits["id"] == want["id"]).all() and (hits["t"].view(np.uint32) == want["t"].view(np.uint32)).all()
    assert (hits["id"] >= soup.shape[0]).mean() > 0.3          # the big triangles are what most rays hit
    mem.free(d_rays); mem.free(d_hits); grid.free(); mem.free(d_tris)


def test_build_degenerate_inputs(mem):
    """Coincident triangles (every cell they touch holds a list of thousands of references: the Shell-sort path of
    sort_cell_refs and long union / subset loops) and zero-area triangles (no normal)."""
    base = scene.make_soup(300, seed=5)
    same = np.repeat(scene.make_soup(1, seed=6), 3000, axis=0)
    flat = scene.tris_from_vertices(np.float32([[0.2, 0.2, 0.2], [0.5, 0.5, 0.5]]), np.float32([[0.4, 0.4, 0.4], [0.5, 0.5, 0.5]]),
                                    np.float32([[0.3, 0.3, 0.3], [0.5, 0.5, 0.5]]))        # a segment and a point
    tris = np.concatenate([base, same, flat]).astype(np.float32)
    grid, G, d_tris = run_stages(mem, tris)
    assert max(np.diff(np.stack([G.small_cells["begin"], np.roll(G.small_cells["begin"], -1)]), axis=0).max(), 1) > 0
    from hagrid_amd import api
    rays = scene.make_rays_incoherent(G.bbox_min, G.bbox_max, 50000, 8)
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * rays.shape[0])
    api.traverse_grid(grid, d_tris, d_rays, d_hits, rays.shape[0])
    hits = mem.download(d_hits, api.HIT_DTYPE, rays.shape[0])
    want, _ = G.traverse(tris, rays, nthreads=8)
    assert (hits["id"] == want["id"]).all() and (hits["t"].view(np.uint32) == want["t"].view(np.uint32)).all()
    # of 3000 coincident triangles either the first tested wins (ascending lists -> the smallest id) or, for rays where
    # abs_det * fl(t / abs_det) rounds above t, every later copy passes `abs_det * tmax > t` again and the last one wins
    # (prims.h:281-283) -- the reference's arithmetic, reproduced bit for bit
    dup = hits["id"][(hits["id"] >= 300) & (hits["id"] < 3300)]
    assert dup.size > 0 and set(np.unique(dup)) <= {300, 3299}
    mem.free(d_rays); mem.free(d_hits); grid.free(); mem.free(d_tris)


def test_precise_expansion_matches_oracle(mem):
    """expand with subset_only = false (compute_overlap, expand.cu:39-57,96-127), an option here."""
    from hagrid_amd import api
    from oracle import oracle as O
    tris = scene.make_soup(40000, seed=77)
    d_tris = mem.upload(tris)
    G = O.Grid.build(tris).merge(0.995).flatten().expand(tris, 3, subset_only=False)
    Gd = O.Grid.build(tris).merge(0.995).flatten().expand(tris, 3)
    assert G.cells.tobytes() != Gd.cells.tobytes()                 # the two modes really differ
    try:
        mem.set_option("expand.subset_only", 0)
        grid = api.Grid()
        api.build_grid(mem, d_tris, tris.shape[0], grid, 0.12, 2.4)
        api.merge_grid(mem, grid, 0.995); api.flatten_grid(mem, grid); api.expand_grid(mem, grid, d_tris, 3)
    finally:
        mem.set_option("expand.subset_only", 1)
    assert_same_grid(grid.download(), G, "precise expand")
    rays = scene.make_rays_incoherent(G.bbox_min, G.bbox_max, 100000, 6)
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * rays.shape[0])
    api.traverse_grid(grid, d_tris, d_rays, d_hits, rays.shape[0])
    hits = mem.download(d_hits, api.HIT_DTYPE, rays.shape[0])
    bf = O.brute_force(tris, rays[:20000], nthreads=8)
    assert (hits["id"][:20000] == bf["id"]).all() and (hits["t"][:20000].view(np.uint32) == bf["t"].view(np.uint32)).all()
    mem.free(d_rays); mem.free(d_hits); grid.free(); mem.free(d_tris)


@pytest.mark.parametrize("flatten", [True, False])
@pytest.mark.parametrize("subset_only", [1, 0])
def test_expansion_with_and_without_the_resolved_voxel_map(mem, flatten, subset_only):
    """expand_grid resolves the voxel map into one word per voxel for its look-ups (expand.hip, `expand.voxel_map`): with it and with the chain through
    the map's levels -- on a flattened map and on the construction's own (deeper chains) -- the expansion is the oracle's."""
    from hagrid_amd import api
    from oracle import oracle as O
    tris = scene.make_soup(30000, seed=5 + subset_only)
    d_tris = mem.upload(tris)
    G = O.Grid.build(tris, 0.12, 3.5).merge(0.995)
    if flatten: G = G.flatten()
    G = G.expand(tris, 3, subset_only=bool(subset_only))
    try:
        mem.set_option("expand.subset_only", subset_only)
        for vm in (1, 0):
            mem.set_option("expand.voxel_map", vm)
            grid = api.Grid()
            api.build_grid(mem, d_tris, tris.shape[0], grid, 0.12, 3.5); api.merge_grid(mem, grid, 0.995)
            if flatten: api.flatten_grid(mem, grid)
            api.expand_grid(mem, grid, d_tris, 3)
            assert_same_grid(grid.download(), G, f"expand, voxel_map {vm}")
            grid.free()
    finally:
        mem.set_option("expand.subset_only", 1); mem.set_option("expand.voxel_map", 1)
    mem.free(d_tris)


@pytest.mark.parametrize("seed", range(20))
def test_random_scenes_and_parameters(mem, seed):
    """Randomised scenes (sizes 1..4000, clustered / stretched / degenerate variants) with random densities, merge
    thresholds and expansion counts: every stage bit-identical to the oracle, then hits bit-identical on the finished grid
    (uncompressed and compressed), with the expansion in both modes."""
    from hagrid_amd import api
    from oracle import oracle as O
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 3, 7, 40, 333, 1500, 4000]))
    tris = scene.make_soup(n, seed=500 + seed).copy()
    kind = seed % 5
    if kind == 1:                       # anisotropic scene: a thin slab
        tris[:, [1, 5, 9]] *= np.float32(0.02)
    elif kind == 2 and n > 10:          # a dense cluster inside the soup
        m = n // 2
        tris[:m, 0:3] = tris[:m, 0:3] * np.float32(0.03) + np.float32(0.6); tris[:m, 4:7] *= np.float32(0.03); tris[:m, 8:11] *= np.float32(0.03)
    elif kind == 3 and n > 3:           # duplicates and zero-area triangles
        tris[1] = tris[0]; tris[2, 4:7] = 0; tris[2, 8:11] = 0
    elif kind == 4:                     # large triangles that span the scene
        tris[: max(1, n // 50), 4:7] *= np.float32(30); tris[: max(1, n // 50), 8:11] *= np.float32(30)
    e1, e2 = tris[:, 4:7], tris[:, 8:11]                      # normals as the front-end computes them: cross(e1, e2)
    nrm = np.stack([e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1], e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2], e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]], axis=1).astype(np.float32)
    tris[:, 3] = nrm[:, 0]; tris[:, 7] = nrm[:, 1]; tris[:, 11] = nrm[:, 2]
    tris = np.ascontiguousarray(tris, np.float32)
    td = float(rng.choice([0.02, 0.12, 0.5, 1.5])); sd = float(rng.choice([0.5, 2.4, 6.0, 12.0]))
    alpha = float(rng.choice([0.0, 0.9, 0.995, 0.9999])); exp = int(rng.choice([0, 1, 3, 5]))
    try:
        mem.set_option("expand.subset_only", seed % 2)
        O.lib()                                   # the oracle follows the same option
        d_tris = mem.upload(tris)
        grid = api.Grid()
        api.build_grid(mem, d_tris, n, grid, td, sd); G = O.Grid.build(tris, td, sd)
        assert_same_grid(grid.download(), G, "build")
        api.merge_grid(mem, grid, alpha); G.merge(alpha)
        assert_same_grid(grid.download(), G, "merge")
        api.flatten_grid(mem, grid); G.flatten()
        assert_same_grid(grid.download(), G, "flatten")
        api.expand_grid(mem, grid, d_tris, exp); G.expand(tris, exp, subset_only=bool(seed % 2))
        assert_same_grid(grid.download(), G, "expand")
        rays = np.concatenate([scene.make_rays_incoherent(G.bbox_min - 0.3, G.bbox_max + 0.3, 20000, 40 + seed),
                               scene.make_rays_primary(G.bbox_min, G.bbox_max, 128, 64)]).astype(np.float32)
        want, _ = G.traverse(tris, rays, nthreads=4)
        brute = O.brute_force(tris, rays[:4000], nthreads=8)
        assert (want["id"][:4000] == brute["id"]).all() and (want["t"][:4000].view(np.uint32) == brute["t"].view(np.uint32)).all()
        d_rays = mem.upload(rays); d_hits = mem.alloc(16 * rays.shape[0])
        for compressed in (False, True):
            if compressed:
                ok = api.compress_grid(mem, grid)
                assert ok == G.compress()
                if not ok:
                    break
                assert_same_grid(grid.download(), G, "compress")
            api.setup_traversal(grid)
            api.traverse_grid(grid, d_tris, d_rays, d_hits, rays.shape[0])
            got = mem.download(d_hits, api.HIT_DTYPE, rays.shape[0])
            assert (got["id"] == want["id"]).all() and (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all(), (seed, compressed)
        mem.free(d_rays); mem.free(d_hits); grid.free(); mem.free(d_tris)
    finally:
        mem.set_option("expand.subset_only", 1)


def test_debug_sync_build_runs_the_whole_path(tmp_path):
    """The HAGRID_DEBUG_SYNC build (per-kernel stream synchronisation + error check, the reference's DEBUG_SYNC of common.h:95-108)
    compiles, reports itself and carries construction + traversal with unchanged results."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys, shutil, importlib
sys.path.insert(0, ROOT)
import hagrid_amd.build as B
B.OBJ = os.path.join(OUT, "obj"); B.LIB = os.path.join(OUT, "libhagrid_amd_debug.so")
B.FLAGS.append("-DHAGRID_DEBUG_SYNC")
B.build(force=True)
import hagrid_amd.lib as L
L.LIB_PATH = B.LIB
import numpy as np
from hagrid_amd import api, scene
assert L.load().hagrid_debug_sync_enabled() == 1
mem = api.MemManager(keep=True)
tris = scene.make_soup(30000); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, 30000, compress=True)
rays = scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 70000, 3)
d_rays = mem.upload(rays); d_hits = mem.alloc(16 * 70000)
api.setup_traversal(grid); mem.set_ray_binning(1)
api.traverse_grid(grid, d_tris, d_rays, d_hits, 70000)
h = mem.download(d_hits, api.HIT_DTYPE, 70000)
print("SUMMARY", grid.num_cells, grid.num_refs, int((h["id"] >= 0).sum()), int(h["id"].astype(np.int64).sum()))
'''.replace("ROOT", repr(root)).replace("OUT", repr(str(tmp_path)))
    import _subproc
    r = _subproc.run([sys.executable, "-c", code], timeout=600)
    assert not r.timed_out, r.diagnosis
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    got = [l for l in r.stdout.splitlines() if l.startswith("SUMMARY")][-1].split()[1:]
    from hagrid_amd import api
    mem = api.MemManager(keep=True)
    assert mem._L.hagrid_debug_sync_enabled() == 0
    tris = scene.make_soup(30000); d_tris = mem.upload(tris)
    grid = api.build_all(mem, d_tris, 30000, compress=True)
    rays = scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 70000, 3)
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * 70000)
    api.setup_traversal(grid); mem.set_ray_binning(1)
    api.traverse_grid(grid, d_tris, d_rays, d_hits, 70000)
    h = mem.download(d_hits, api.HIT_DTYPE, 70000)
    assert got == [str(grid.num_cells), str(grid.num_refs), str(int((h["id"] >= 0).sum())), str(int(h["id"].astype(np.int64).sum()))]
    mem.close()
