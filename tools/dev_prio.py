"""DEV TOOL: issue priority by age in the one-pass image kernel (traverse.prio_step) on the headline batch and others."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
tris = scene.make_soup(1_000_000); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, tris.shape[0])
api.setup_traversal(grid)
mem.set_option("traverse.generations", 0)
batches = {"primary 1024^2": scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 1024),
           "primary 2048^2": scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 2048, 2048),
           "incoherent 1M": scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 1 << 20, scene.RAY_SEED_BASE + 4)}
for name, rays in batches.items():
    n = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    ref = None
    for p in [int(x) for x in os.environ.get("PRIO", "0,4,6,8,12,16,24,32,48").split(",")]:
        mem.set_option("traverse.prio_step", p)
        for _ in range(3): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n), mem) for _ in range(21))
        h = mem.download(d_hits, api.HIT_DTYPE, n)
        if ref is None: ref = h
        same = bool((h["id"] == ref["id"]).all() and (h["t"].view(np.uint32) == ref["t"].view(np.uint32)).all())
        print(json.dumps({"batch": name, "prio_step": p, "ms_median": round(t[10], 4), "ms_min": round(t[0], 4), "Grays/s": round(n / t[10] / 1e6, 2), "identical": same}), flush=True)
    mem.free(d_rays); mem.free(d_hits)
