"""Runs the BASELINE.md configs that fit one GPU through the C ABI and prints one JSON line per config
(build ms, Mrays/s, algorithmic GB/s).  Multi-GPU configs are run with their per-GPU share of the rays."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene

mem = api.MemManager(keep=True)
CONFIGS = [
    dict(name="1: soup-10k, 64k incoherent", tris=10_000, rays=("incoherent", 65536), params={}),
    dict(name="2: soup-1M, 1M primary (1024^2)", tris=1_000_000, rays=("primary", 1024), params={}),
    dict(name="3: soup-1M td .15 sd 3.0, 16M primary (4096^2)", tris=1_000_000, rays=("primary", 4096), params=dict(top_density=0.15, snd_density=3.0)),
    dict(name="4: soup-1M, 16M incoherent (per-GPU share of 128M), binned", tris=1_000_000, rays=("incoherent", 1 << 24), params={}, bin=1),
    dict(name="4: soup-1M, 16M incoherent, not binned", tris=1_000_000, rays=("incoherent", 1 << 24), params={}),
    dict(name="5: soup-8M --compress, 8M bounce (per-GPU share of 64M), binned", tris=8_000_000, rays=("bounce", 2896), params=dict(compress=True), bin=1),
    dict(name="5: soup-8M --compress, 8M bounce, not binned", tris=8_000_000, rays=("bounce", 2896), params=dict(compress=True)),
    dict(name="5: soup-8M --compress, 8M bounce, image width given (tile packets)", tris=8_000_000, rays=("bounce", 2896), params=dict(compress=True), width=2896),
    dict(name="2: soup-1M, 1M primary, occlusion rays (any-hit)", tris=1_000_000, rays=("primary", 1024), params={}, flags=1),
    dict(name="4: soup-1M, 16M incoherent, binned, construction format (no image)", tris=1_000_000, rays=("incoherent", 1 << 24), params={}, bin=1, image=0),
    dict(name="2: soup-1M, 1M primary, construction format (no image)", tris=1_000_000, rays=("primary", 1024), params={}, image=0),
    dict(name="6 (extra): clustered-1M (six dense blobs in a sparse soup, shift 6), 1M primary", tris=1_000_000, scene="clustered", rays=("primary", 1024), params={}),
    dict(name="6 (extra): clustered-1M, 1M primary, construction format (no image)", tris=1_000_000, scene="clustered", rays=("primary", 1024), params={}, image=0),
    dict(name="6 (extra): clustered-1M, 1M incoherent", tris=1_000_000, scene="clustered", rays=("incoherent", 1 << 20), params={}),
    dict(name="6 (extra): clustered-1M, 1M incoherent, construction format (no image)", tris=1_000_000, scene="clustered", rays=("incoherent", 1 << 20), params={}, image=0),
]
cache = {}
for c in CONFIGS:
    n = c["tris"]
    if cache.get("n") != (n, c.get("scene")):
        if cache.get("d_tris"): mem.free(cache["d_tris"])
        cache = {"n": (n, c.get("scene")), "tris": scene.make_clustered() if c.get("scene") == "clustered" else scene.make_soup(n)}
        cache["d_tris"] = mem.upload(cache["tris"])
    tris, d_tris = cache["tris"], cache["d_tris"]
    grid = api.build_all(mem, d_tris, n, **c["params"])
    bt = []
    for _ in range(3):
        grid.free(); bt.append(api.profile(lambda: api.build_all(mem, d_tris, n, grid=grid, **c["params"]), mem))
    kind, size = c["rays"]
    if kind == "incoherent":
        rays = scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, size, scene.RAY_SEED_BASE + 4)
    else:
        rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, size, size)
    nr = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * nr)
    if kind == "bounce":
        api.traverse_grid(grid, d_tris, d_rays, d_hits, nr)
        h = mem.download(d_hits, api.HIT_DTYPE, nr)
        rays = scene.make_rays_bounce(tris, rays, h, grid.bbox_min, grid.bbox_max, scene.RAY_SEED_BASE + 5)
        mem.copy_h2d(d_rays, rays)
    mem.set_ray_binning(c.get("bin", 0))
    mem.set_option("traverse.image_width", c.get("width", 0))
    mem.set_option("traverse.image", c.get("image", 2))
    api.setup_traversal(grid)
    flags = c.get("flags", 0)
    st = api.traverse_grid_stats(grid, d_tris, d_rays, d_hits, nr)
    ab = api.algorithmic_bytes(st, bool(grid.small_cells))
    for _ in range(2): api.traverse_grid(grid, d_tris, d_rays, d_hits, nr, flags)
    t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, nr, flags)) for _ in range(7))
    mem.set_ray_binning(0); mem.set_option("traverse.image_width", 0); mem.set_option("traverse.image", 2)
    print(json.dumps({"config": c["name"], "grid": grid.summary(), "build_ms": round(float(np.mean(bt)), 2), "rays": nr,
                      "traverse_ms_median": round(t[3], 3), "mrays": round(nr / t[3] / 1e3, 1), "hit_fraction": round(st["hits"] / nr, 3),
                      "bytes_per_ray": round(ab["B_ray"] / nr, 1), "alg_GBps": round(ab["B_ray"] / t[3] / 1e6, 1)}), flush=True)
    mem.free(d_rays); mem.free(d_hits); grid.free()
