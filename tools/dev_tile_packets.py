"""DEV TOOL: tile packets in the kernel -- off / detected / given row length, super-tile size sweep, kernel choice."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene

N = int(os.environ.get("N", 1000000))
mem = api.MemManager(keep=True)
tris = scene.make_soup(N); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, N, compress=bool(int(os.environ.get("COMPRESS", "0"))))
api.setup_traversal(grid)

def bench(d_rays, d_hits, n, rounds=11):
    for _ in range(2): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)) for _ in range(rounds))
    return round(t[len(t) // 2], 4)

for W, H in ((1024, 1024), (1920, 1080), (1280, 720), (4096, 4096)):
    rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, W, H)
    n = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    res = {"W": W, "H": H}
    for variant in (0,):
        mem.set_option("traverse.variant", variant)
        for width in (-1, 0):
            mem.set_option("traverse.image_width", width)
            for sl, ch in (((5, 0),) if width < 0 else [(s, c) for s in (4, 5, 7) for c in (-1, 2, 4, 6, 8) if c <= 2 * s]):
                mem.set_option("traverse.xcd_chunk", ch)
                mem.set_option("traverse.super_tile", sl)
                res[f"w{width}_s{sl}_c{ch}"] = bench(d_rays, d_hits, n)
    print(json.dumps(res), flush=True)
    mem.free(d_rays); mem.free(d_hits)
