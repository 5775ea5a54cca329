"""DEV TOOL: launch time over (image size x traverse.quad_tail): which share of the tiles should start with four lanes per ray."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
tris = scene.make_soup(1_000_000); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, tris.shape[0]); api.setup_traversal(grid)
sizes = [(256, 256), (384, 384), (512, 384), (512, 512), (640, 480), (800, 600), (960, 540), (1024, 768), (1280, 720), (1024, 1024), (1280, 1024), (1536, 1024), (1600, 1200), (1920, 1080)]
pcts = [0, 12, 25, 37, 50, 75, 100]
import os
for kv in filter(None, os.environ.get('OPTS', '').split(',')):
    k, v = kv.split('='); mem.set_option(k, int(v))
for w, h in sizes:
    rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, w, h); n = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    go = lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    row = {}
    for rep in range(2):
        for p in pcts:
            mem.set_option("traverse.quad_tail", p)
            for _ in range(40): go()
            mem.synchronize()
            ms = sorted(api.profile(lambda: [go() for _ in range(10)], mem) / 10 for _ in range(8))[3]
            row[p] = min(row.get(p, 9e9), ms)
    best = min(row, key=row.get)
    print(json.dumps({"size": f"{w}x{h}", "tiles": (n + 63) // 64, "tiles/slots": round((n + 63) // 64 / 8192, 2), "best %": best,
                      "gain % vs 0": round(100 * (1 - row[best] / row[0]), 1), "ms": {str(p): round(v, 4) for p, v in row.items()}}), flush=True)
    mem.free(d_rays); mem.free(d_hits)
