"""DEV TOOL: how much would spatial binning of incoherent rays buy?  Host-side sort, unchanged kernels."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
N = 1000000
tris = scene.make_soup(N); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, N)
n = 1 << 22
rays = scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, n, scene.RAY_SEED_BASE + 4)
lo, hi = grid.bbox_min, grid.bbox_max
def part1by2(x):
    x = x.astype(np.uint32) & 0x3ff
    x = (x | (x << 16)) & 0x30000ff; x = (x | (x << 8)) & 0x300f00f; x = (x | (x << 4)) & 0x30c30c3; x = (x | (x << 2)) & 0x9249249
    return x
def keys(res_bits):
    res = 1 << res_bits
    c = np.clip(((rays[:, 0:3] - lo) / (hi - lo) * res).astype(np.int32), 0, res - 1)
    m = part1by2(c[:, 0]) | (part1by2(c[:, 1]) << 1) | (part1by2(c[:, 2]) << 2)
    octant = (rays[:, 4] < 0).astype(np.uint32) | ((rays[:, 5] < 0).astype(np.uint32) << 1) | ((rays[:, 6] < 0).astype(np.uint32) << 2)
    return m, octant
def bench(r, variant):
    mem.set_option("traverse.variant", variant)
    d_rays = mem.upload(r); d_hits = mem.alloc(16 * n)
    for _ in range(2): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)) for _ in range(5))
    mem.free(d_rays); mem.free(d_hits)
    return round(n / t[2] / 1e3)
row = {"unsorted": {v: bench(rays, v) for v in (2, 3)}}
for bits in (3, 4, 5, 6, 8):
    m, o = keys(bits)
    for name, key in (("morton", m), ("oct+morton", (o.astype(np.uint64) << 32) | m), ("morton+oct", (m.astype(np.uint64) << 3) | o)):
        perm = np.argsort(key, kind="stable")
        row[f"{name}_{bits}b"] = {v: bench(np.ascontiguousarray(rays[perm]), v) for v in (2, 3)}
    print(json.dumps(row), flush=True); row = {}
