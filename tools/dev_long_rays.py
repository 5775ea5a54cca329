"""DEV TOOL: what the longest rays of the headline batch cost.  Rays that visit more than T cells are switched off (tmax = tmin: they miss
the grid at once) and the launch is timed again: if the launch is as long as its longest dependent chains, removing one per cent of the
rays removes far more than one per cent of the time."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
for kv in filter(None, os.environ.get("OPTS", "").split(",")):
    k, v = kv.split("="); mem.set_option(k, int(v))
W = int(os.environ.get("W", 1024))
tris = scene.make_soup(1_000_000); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, tris.shape[0]); api.setup_traversal(grid)
rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, W, W); n = rays.shape[0]
d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n); d_steps = mem.alloc(4 * n)
st = api.traverse_grid_stats(grid, d_tris, d_rays, d_hits, n, d_steps)
steps = mem.download(d_steps, np.int32, n)            # the reference's count: cells + tests
cells = None
def timed():
    t0 = time.time()
    while time.time() - t0 < 0.1:
        for _ in range(20): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        mem.synchronize()
    return sorted(api.profile(lambda: [api.traverse_grid(grid, d_tris, d_rays, d_hits, n) for _ in range(20)], mem) / 20 for _ in range(10))[5]
print(json.dumps({"rays": n, "steps (cells + tests) mean/p50/p90/p99/p99.9/max": [float(np.round(x, 1)) for x in (steps.mean(), *np.percentile(steps, [50, 90, 99, 99.9]), steps.max())]}), flush=True)
for pct in (100, 99.9, 99.5, 99, 98, 95, 90, 75, 50):
    T = np.percentile(steps, pct) if pct < 100 else steps.max()
    r = rays.copy(); off = steps > T
    r[off, 7] = r[off, 3] - 1.0                           # tmax < tmin: the ray never enters the grid
    mem.copy_h2d(d_rays, r)
    print(json.dumps({"rays kept %": pct, "steps threshold": float(T), "rays off": int(off.sum()), "work kept %": round(100.0 * steps[~off].sum() / steps.sum(), 1), "launch ms": round(timed(), 5)}), flush=True)
