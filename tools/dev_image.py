"""DEV TOOL: traversal image vs construction format, per batch shape.  usage: python tools/dev_image.py"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene

N = int(os.environ.get("N", 1000000))
mem = api.MemManager(keep=True)
mem.set_option("traverse.image", int(os.environ.get("FMT", 1)))
tris = scene.make_soup(N); d_tris = mem.upload(tris)
params = dict(top_density=float(os.environ.get("TD", 0.12)), snd_density=float(os.environ.get("SD", 2.4)))
grid = api.build_all(mem, d_tris, N, **params)
t_setup = sorted(api.profile(lambda: api.setup_traversal(grid)) for _ in range(5))[2]
print(json.dumps({"grid": grid.summary(), "setup_traversal_ms": round(t_setup, 3)}), flush=True)

def bench(d_rays, d_hits, n, rounds=11):
    for _ in range(2): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)) for _ in range(rounds))
    return round(t[len(t) // 2], 4)

sets = {"primary 1024^2": scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 1024),
        "primary 1920x1080": scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1920, 1080),
        "incoherent 1M": scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 1 << 20, scene.RAY_SEED_BASE + 4),
        "primary 4096^2": scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 4096, 4096),
        "incoherent 16M": scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 1 << 24, scene.RAY_SEED_BASE + 4)}
for name, rays in sets.items():
    n = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    res = {"rays": name}
    ref = None
    for label, variant, binning in (("v2", 2, 0), ("v3", 3, 0), ("image", 4, 0), ("v2+bin", 2, 1), ("image+bin", 4, 1)):
        if binning and "incoherent" not in name: continue
        mem.set_option("traverse.variant", variant); mem.set_ray_binning(binning)
        res[label] = bench(d_rays, d_hits, n)
        h = mem.download(d_hits, api.HIT_DTYPE, n)
        if ref is None: ref = h
        else: assert (h["id"] == ref["id"]).all() and (h["t"].view(np.uint32) == ref["t"].view(np.uint32)).all(), label
    res["mrays_image"] = round(n / res["image"] / 1e3)
    print(json.dumps(res), flush=True)
    mem.free(d_rays); mem.free(d_hits)
