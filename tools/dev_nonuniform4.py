"""DEV TOOL: clustered scene, rays aimed at the blobs (every ray ends in dense geometry, as primary rays of real scenes do):
traversal image formats and ray binning."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
tris = scene.make_clustered(); N = tris.shape[0]
d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, N)
n = 1 << 20
aimed = scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, n, 12).copy()
k = np.arange(n) % 6
centre = np.stack([0.17 + 0.14 * k, 0.32 + 0.08 * k, 0.22 + 0.1 * k], axis=1).astype(np.float32)
aimed[:, 4:7] = centre - aimed[:, 0:3] + np.float32(0.02) * aimed[:, 4:7]
# a camera close to one blob: coherent rays that all end in dense geometry
yy, xx = np.meshgrid(np.arange(1024, dtype=np.float32), np.arange(1024, dtype=np.float32), indexing="ij")
close = np.zeros((n, 8), np.float32)
close[:, 0:3] = np.float32([0.17, 0.32, 0.22 - 0.05]); close[:, 3] = 0.0
close[:, 4] = (xx.reshape(-1) / np.float32(1024.0) - np.float32(0.5)); close[:, 5] = (yy.reshape(-1) / np.float32(1024.0) - np.float32(0.5)); close[:, 6] = 1.0
close[:, 7] = np.float32(10.0)
for label, rays in (("aimed at the blobs, incoherent origins", aimed), ("camera close to one blob", close)):
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n); d_steps = mem.alloc(4 * n)
    api.traverse_grid_stats(grid, d_tris, d_rays, d_hits, n, d_steps); s = mem.download(d_steps, np.int32, n)
    h = mem.download(d_hits, api.HIT_DTYPE, n)
    res = {"rays": label, "steps_mean": round(float(s.mean()), 1), "hit a blob": float((h["id"] >= 100000).mean())}
    for img in (2, 0, 1, 2, 0):
        mem.set_option("traverse.image", img); api.setup_traversal(grid)
        for _ in range(2): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)) for _ in range(9))
        res[f"image{img}" + ("b" if f"image{img}" in res else "")] = round(t[4], 4)
    print(json.dumps(res), flush=True)
    mem.free(d_rays); mem.free(d_hits); mem.free(d_steps)
