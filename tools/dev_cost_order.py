"""DEV TOOL: does ordering the 1M primary batch by (predicted) ray cost shorten the launch?  Host-side reorder,
unchanged kernels.  Orders: buffer order; true step count descending (upper bound on the benefit); box path length
descending, quantised to q buckets with pixel order kept inside a bucket."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene

N = int(os.environ.get("N", 1000000))
W = int(os.environ.get("W", 1024))
mem = api.MemManager(keep=True)
tris = scene.make_soup(N); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, N, compress=bool(int(os.environ.get("COMPRESS", "0"))))
rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, W, W)
n = rays.shape[0]

def bench(r, variant=2, rounds=9):
    mem.set_option("traverse.variant", variant)
    d_rays = mem.upload(np.ascontiguousarray(r)); d_hits = mem.alloc(16 * n)
    for _ in range(2): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)) for _ in range(rounds))
    mem.free(d_rays); mem.free(d_hits)
    return round(t[len(t) // 2], 4), round(t[0], 4)

# true steps
d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n); d_steps = mem.alloc(4 * n)
api.traverse_grid_stats(grid, d_tris, d_rays, d_hits, n, d_steps)
steps = mem.download(d_steps, np.int32, n)
mem.free(d_rays); mem.free(d_hits); mem.free(d_steps)
print(json.dumps({"steps_mean": float(steps.mean()), "steps_p50": int(np.percentile(steps, 50)), "steps_p99": int(np.percentile(steps, 99)), "steps_max": int(steps.max())}))

# slab path length inside the grid box
org = rays[:, 0:3].astype(np.float64); d = rays[:, 4:7].astype(np.float64)
with np.errstate(divide="ignore", invalid="ignore"):
    inv = 1.0 / d
    t0 = (np.asarray(grid.bbox_min, np.float64) - org) * inv; t1 = (np.asarray(grid.bbox_max, np.float64) - org) * inv
tn = np.maximum(np.minimum(t0, t1).max(axis=1), rays[:, 3]); tf = np.minimum(np.maximum(t0, t1).min(axis=1), rays[:, 7])
plen = np.where(tf > tn, (tf - tn) * np.linalg.norm(d, axis=1), 0.0)
print(json.dumps({"corr_len_steps": float(np.corrcoef(plen, steps)[0, 1])}))

res = {"buffer": bench(rays)}
res["steps_desc"] = bench(rays[np.argsort(-steps, kind="stable")])
res["steps_asc"] = bench(rays[np.argsort(steps, kind="stable")])
for q in (4, 8, 16, 64):
    b = np.minimum((plen / (plen.max() + 1e-9) * q).astype(np.int32), q - 1)
    res[f"len_desc_q{q}"] = bench(rays[np.argsort(-b, kind="stable")])
    sq = np.minimum((steps.astype(np.float64) / steps.max() * q).astype(np.int32), q - 1)
    res[f"steps_desc_q{q}"] = bench(rays[np.argsort(-sq, kind="stable")])
# tile order: 8x8 pixel tiles per wave instead of 64x1 rows
idx = np.arange(n).reshape(W, W)
tiles = idx.reshape(W // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
res["tiles8x8"] = bench(rays[tiles])
res["tiles8x8_v3"] = bench(rays[tiles], 3)
res["steps_desc_v3"] = bench(rays[np.argsort(-steps, kind="stable")], 3)
print(json.dumps(res))
