"""DEV TOOL: lane/wave assignment of a W x W primary batch -- rows of 64 (buffer order) vs pixel tiles vs Z-curve.
Host-side reorder, unchanged kernels."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene

N = int(os.environ.get("N", 1000000))
mem = api.MemManager(keep=True)
tris = scene.make_soup(N); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, N, compress=bool(int(os.environ.get("COMPRESS", "0"))))

def bench(r, variant, rounds=9):
    n = r.shape[0]
    mem.set_option("traverse.variant", variant)
    d_rays = mem.upload(np.ascontiguousarray(r)); d_hits = mem.alloc(16 * n)
    for _ in range(2): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)) for _ in range(rounds))
    mem.free(d_rays); mem.free(d_hits)
    return round(t[len(t) // 2], 4)

def tile(W, th, tw):
    idx = np.arange(W * W).reshape(W, W)
    return idx.reshape(W // th, th, W // tw, tw).transpose(0, 2, 1, 3).reshape(-1)

def zcurve(W):
    y, x = np.divmod(np.arange(W * W, dtype=np.uint64), W)
    def spread(v):
        v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555
        return v
    return np.argsort(spread(x) | (spread(y) << 1), kind="stable")

def tiles_in_tiles(W, outer, th=8, tw=8):
    """8x8 tiles, themselves listed row-major inside outer x outer pixel super-tiles"""
    idx = np.arange(W * W).reshape(W, W)
    a = idx.reshape(W // outer, outer // th, th, W // outer, outer // tw, tw).transpose(0, 3, 1, 4, 2, 5)
    return a.reshape(-1)

for W, variants in ((1024, (2,)), (4096, (2, 3))):
    rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, W, W)
    orders = {"rows": None, "t8x8": tile(W, 8, 8), "t4x16": tile(W, 4, 16), "t16x4": tile(W, 16, 4), "t2x32": tile(W, 2, 32),
              "z": zcurve(W), "t8x8_in64": tiles_in_tiles(W, 64), "t8x8_in128": tiles_in_tiles(W, 128), "t8x8_in256": tiles_in_tiles(W, 256)}
    for v in variants:
        res = {k: bench(rays if p is None else rays[p], v) for k, p in orders.items()}
        print(json.dumps({"W": W, "variant": v, **res}), flush=True)
