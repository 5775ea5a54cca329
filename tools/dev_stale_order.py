"""DEV TOOL: what a STALE tile order costs.  The order learned on image A is applied to image B (the same camera image flipped top to bottom, or a
camera from the other side) written into the same buffer; launches are timed before the next refresh re-learns it (31 launches), after it, and
in the default order."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
tris = scene.make_soup(1_000_000); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, tris.shape[0]); api.setup_traversal(grid)
W = 1024
A = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, W, W)
B_flip = np.ascontiguousarray(A.reshape(W, W, 8)[::-1].reshape(-1, 8))
B_mirror = np.ascontiguousarray(A.reshape(W, W, 8)[:, ::-1].reshape(-1, 8))
B_shift = np.ascontiguousarray(np.roll(A.reshape(W, W, 8), 64, axis=1).reshape(-1, 8))       # (a pan by 64 pixels, wrapped)
n = A.shape[0]
d_rays = mem.upload(A); d_hits = mem.alloc(16 * n)
go = lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
def timed(k): return api.profile(lambda: [go() for _ in range(k)], mem) / k
for name, B in (("flipped top to bottom", B_flip), ("mirrored left to right", B_mirror), ("panned by 64 pixels", B_shift)):
    mem.set_option("traverse.tile_order", -1)
    mem.copy_h2d(d_rays, A)
    for _ in range(80): go()
    mem.synchronize()
    a_ms = timed(20)
    for _ in range(40): go()                      # (put the refresh counter somewhere in the middle is not possible from here: time short groups instead)
    mem.copy_h2d(d_rays, B)
    stale = [timed(4) for _ in range(6)]          # 24 launches: stale until the refresh falls into one of them
    for _ in range(80): go()
    mem.synchronize()
    b_ms = timed(20)
    mem.set_option("traverse.tile_order", 0)
    for _ in range(10): go()
    b0_ms = timed(20)
    print(json.dumps({"image B": name, "A in its own order": round(a_ms, 4), "B in A's order, groups of 4 launches": [round(x, 4) for x in stale],
                      "B in its own order": round(b_ms, 4), "B in the default order": round(b0_ms, 4)}), flush=True)
