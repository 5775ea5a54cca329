"""DEV TOOL: what bounds a small batch?  (a) time vs batch size, (b) effect of ray order (longest first)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
N = 1000000
tris = scene.make_soup(N); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, N)
rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 1024)
n = rays.shape[0]
d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n); d_steps = mem.alloc(4 * n)
api.traverse_grid_stats(grid, d_tris, d_rays, d_hits, n, d_steps)
steps = mem.download(d_steps, np.int32, n)
print("steps: mean", steps.mean(), "max", steps.max(), "p99.9", np.percentile(steps, 99.9))
def timeit(ptr, cnt, variant, reps=7):
    mem.set_option("traverse.variant", variant)
    for _ in range(2): api.traverse_grid(grid, d_tris, ptr, d_hits, cnt)
    t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, ptr, d_hits, cnt)) for _ in range(reps))
    return t[len(t) // 2]
order = np.argsort(-steps, kind="stable")
longest = np.ascontiguousarray(rays[order])
d_long = mem.upload(longest)
for variant in (1, 2, 3):
    row = {"variant": variant}
    for cnt in (64, 4096, 65536, 524288, 1048576):
        row[f"orig_{cnt}"] = round(timeit(d_rays + 0, cnt, variant), 4)
    row["sorted_desc_1M"] = round(timeit(d_long, n, variant), 4)
    row["top64_longest"] = round(timeit(d_long, 64, variant), 4)
    row["top4096_longest"] = round(timeit(d_long, 4096, variant), 4)
    print(json.dumps(row), flush=True)
