"""DEV TOOL: clustered scene built with compress=True (SmallCells, shift 6): traversal image (blocks + nested blocks) vs construction format."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
tris = scene.make_clustered(); N = tris.shape[0]
d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, N, compress=True)
print(json.dumps(grid.summary()))
for label, rays in (("1024^2 primary", scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 1024)), ("1M incoherent", scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 1 << 20, 9))):
    n = rays.shape[0]; d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    res = {"rays": label}; ref = None
    for img in (2, 0, 2, 0):
        mem.set_option("traverse.image", img); api.setup_traversal(grid)
        for _ in range(2): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)) for _ in range(9))
        res.setdefault(f"image{img}", []).append(round(t[4], 4))
        h = mem.download(d_hits, api.HIT_DTYPE, n)
        if ref is None: ref = h
        else: assert (h["id"] == ref["id"]).all() and (h["t"].view(np.uint32) == ref["t"].view(np.uint32)).all()
    print(json.dumps(res), flush=True)
    mem.free(d_rays); mem.free(d_hits)
