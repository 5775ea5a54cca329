"""DEV TOOL: v3 parameter sweep at 16M primary rays."""
import os, sys, json, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
N = 1000000
tris = scene.make_soup(N); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, N)
rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 4096, 4096)
n = rays.shape[0]; d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
mem.set_option("traverse.variant", 3)
res = []
for refill, chunk, waves in itertools.product((8, 16, 24, 32, 48), (0, 256, 1024), (24, 32)):
    mem.set_option("traverse.refill_at", refill); mem.set_option("traverse.chunk", chunk); mem.set_option("traverse.waves_per_cu", waves)
    for _ in range(2): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)) for _ in range(5))
    res.append((round(n / t[2] / 1e3), refill, chunk, waves))
res.sort(reverse=True)
for r in res[:8]: print(r)
print("worst", res[-1])
