"""DEV TOOL: non-uniform scene of dev_nonuniform.py: the heaviest 8x8 pixel tiles alone (one wavefront each), image 2/1/0."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.argv = ["x"]
import numpy as np
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev_nonuniform.py")).read().split("prim = scene.make_rays_primary")[0])
prim = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 1024)
n = prim.shape[0]
d_rays = mem.upload(prim); d_hits = mem.alloc(16 * n); d_steps = mem.alloc(4 * n)
api.traverse_grid_stats(grid, d_tris, d_rays, d_hits, n, d_steps); s = mem.download(d_steps, np.int32, n).reshape(128, 8, 128, 8)
mem.free(d_rays); mem.free(d_hits); mem.free(d_steps)
tile_max = s.max(axis=(1, 3)); order = np.argsort(-tile_max.reshape(-1))
P = prim.reshape(128, 8, 128, 8, 8)
def run(label, rays):
    rays = np.ascontiguousarray(rays.reshape(-1, 8)); m = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * m)
    res = {"rays": label, "n": m}
    for img in (2, 0, 1, 2, 0):
        mem.set_option("traverse.image", img); api.setup_traversal(grid)
        for _ in range(2): api.traverse_grid(grid, d_tris, d_rays, d_hits, m)
        t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, m)) for _ in range(9))
        res[f"image{img}" + ("b" if f"image{img}" in res else "")] = round(t[4], 4)
    print(json.dumps(res), flush=True)
    mem.free(d_rays); mem.free(d_hits)
for k in range(6):
    ty, tx = divmod(int(order[k]), 128)
    run(f"tile ({ty},{tx}) steps max {int(tile_max[ty, tx])} mean {float(s[ty, :, tx, :].mean()):.1f}", P[ty, :, tx, :].transpose(0, 1, 2))
# all tiles but the 64 heaviest
keep = np.ones(128 * 128, bool); keep[order[:64]] = False
rest = P.transpose(0, 2, 1, 3, 4).reshape(128 * 128, 64, 8)[keep]
run("frame without the 64 heaviest tiles (tile order, no packets)", rest)
