"""DEV TOOL: automatic ray binning (mode 2) against off (0) and always (1) on primary, bounce and random batches."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
N = 1000000
tris = scene.make_soup(N); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, N)
def run(rays, mode, rounds=9):
    n = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    mem.set_ray_binning(mode)
    for _ in range(2): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)) for _ in range(rounds))
    h = mem.download(d_hits, api.HIT_DTYPE, n)
    mem.free(d_rays); mem.free(d_hits); mem.set_ray_binning(0)
    return round(t[len(t) // 2], 4), h
for W in (1024, 2048):
    prim = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, W, W)
    t0, h0 = run(prim, 0)
    bounce = scene.make_rays_bounce(tris, prim, h0, grid.bbox_min, grid.bbox_max, 5)
    inc = scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, W * W, 9)
    for name, rays in (("primary", prim), ("bounce", bounce), ("incoherent", inc)):
        res = {"rays": f"{name} {W}x{W}"}
        ref = None
        for mode in (0, 1, 2):
            res[f"mode{mode}"], h = run(rays, mode)
            if ref is None: ref = h
            else: assert (h["id"] == ref["id"]).all() and (h["t"].view(np.uint32) == ref["t"].view(np.uint32)).all()
        print(json.dumps(res), flush=True)
