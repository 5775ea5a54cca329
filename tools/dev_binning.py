"""DEV TOOL: GPU ray binning on/off for incoherent and primary batches."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
N = 1000000
tris = scene.make_soup(N); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, N)
for kind, n in (("incoherent", 1 << 20), ("incoherent", 1 << 22), ("incoherent", 1 << 24), ("primary", 1 << 20), ("primary", 1 << 24)):
    w = int(n ** 0.5)
    rays = scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, n, scene.RAY_SEED_BASE + 4) if kind == "incoherent" else scene.make_rays_primary(grid.bbox_min, grid.bbox_max, w, w)
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    row = {"rays": kind, "n": n}; ref = None
    for mode in (0, 1):
        mem.set_ray_binning(mode)
        for _ in range(2): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)) for _ in range(7))
        row[f"bin{mode}_mrays"] = round(n / t[3] / 1e3); row[f"bin{mode}_ms"] = round(t[3], 3)
        h = mem.download(d_hits, api.HIT_DTYPE, n)
        if ref is None: ref = h
        else: assert (h["id"] == ref["id"]).all() and (h["t"].view(np.uint32) == ref["t"].view(np.uint32)).all()
    print(json.dumps(row), flush=True)
    mem.free(d_rays); mem.free(d_hits)
