"""DEV TOOL: the 1M-primary step with the host transfers included (rays up, hits down; pageable numpy buffers through hagrid_mem_copy)."""
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
N = 1000000
tris = scene.make_soup(N); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, N)
rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 1024); n = rays.shape[0]
d_rays = mem.alloc(32 * n); d_hits = mem.alloc(16 * n)
api.setup_traversal(grid)
out = []
for _ in range(12):
    t0 = time.perf_counter()
    mem.copy_h2d(d_rays, rays)
    api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    h = mem.download(d_hits, api.HIT_DTYPE, n)
    out.append((time.perf_counter() - t0) * 1e3)
out.sort()
print(json.dumps({"ms per step incl. 32 MB up + 16 MB down (median)": round(out[len(out) // 2], 3), "min": round(out[0], 3), "Mrays/s": round(n / out[len(out) // 2] / 1e3, 1)}))
