"""DEV TOOL: non-uniform scene of dev_nonuniform.py, frame split into 8 bands of 128 rows and growing prefixes: image 2 vs 0."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.argv = ["x"]
import numpy as np
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev_nonuniform.py")).read().split("prim = scene.make_rays_primary")[0])
prim = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 1024).reshape(1024, 1024, 8)
def run(label, rays):
    rays = np.ascontiguousarray(rays.reshape(-1, 8)); n = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n); d_steps = mem.alloc(4 * n)
    api.traverse_grid_stats(grid, d_tris, d_rays, d_hits, n, d_steps); s = mem.download(d_steps, np.int32, n)
    res = {"rays": label, "n": n, "steps_mean": round(float(s.mean()), 1), "steps_max": int(s.max())}
    for img in (2, 0, 2, 0):
        mem.set_option("traverse.image", img); api.setup_traversal(grid)
        for _ in range(2): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)) for _ in range(9))
        res[f"image{img}" + ("b" if f"image{img}" in res else "")] = round(t[4], 4)
    print(json.dumps(res), flush=True)
    mem.free(d_rays); mem.free(d_hits); mem.free(d_steps)
for b in range(8): run(f"rows {128*b}..{128*b+127}", prim[128 * b: 128 * b + 128])
for k in (256, 512, 768, 1024): run(f"rows 0..{k-1}", prim[:k])
