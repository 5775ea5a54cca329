import json, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
tris = scene.make_soup(1_000_000); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, tris.shape[0]); api.setup_traversal(grid)
W=1024
rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, W, W); n = rays.shape[0]
d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n); d_steps = mem.alloc(4 * n)
api.traverse_grid_stats(grid, d_tris, d_rays, d_hits, n, d_steps)
s = mem.download(d_steps, np.int32, n).reshape(W, W)
b = s.reshape(16, 64, 16, 64)
print("mean steps per 64x64 block (rows = image rows top to bottom):")
for r in b.mean(axis=(1,3)): print(" ".join(f"{v:5.1f}" for v in r))
print("max steps per block:")
for r in b.max(axis=(1,3)): print(" ".join(f"{v:5d}" for v in r))
t = s.reshape(128, 8, 128, 8).max(axis=(1,3))   # longest ray per 8x8 tile
print("longest ray per tile: mean", t.mean(), "p90", np.percentile(t,90), "max", t.max())
print("row-band (64 px) mean of per-tile longest:", " ".join(f"{v:5.1f}" for v in t.reshape(16,8,128).mean(axis=(1,2))))
