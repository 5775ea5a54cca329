"""DEV TOOL: row-length detection from the origins (bounce rays) -- cost on primary rays, gain on bounce rays."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
N = int(os.environ.get("N", 1000000)); W = int(os.environ.get("W", 1024)); COMP = bool(int(os.environ.get("COMPRESS", "0")))
tris = scene.make_soup(N); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, N, compress=COMP); api.setup_traversal(grid)
prim = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, W, W); n = prim.shape[0]
d_rays = mem.upload(prim); d_hits = mem.alloc(16 * n)
def run(rounds=11):
    for _ in range(2): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)) for _ in range(rounds))
    return round(t[len(t) // 2], 4)
res = {"N": N, "W": W, "compressed": COMP}
for v in (0, 1): mem.set_option("traverse.detect_origins", v); res[f"primary_detect{v}"] = run()
api.traverse_grid(grid, d_tris, d_rays, d_hits, n); h = mem.download(d_hits, api.HIT_DTYPE, n)
bounce = scene.make_rays_bounce(tris, prim, h, grid.bbox_min, grid.bbox_max, scene.RAY_SEED_BASE + 5)
mem.copy_h2d(d_rays, bounce)
for v in (0, 1): mem.set_option("traverse.detect_origins", v); res[f"bounce_detect{v}"] = run()
mem.set_option("traverse.image_width", W); res["bounce_width_given"] = run(); mem.set_option("traverse.image_width", 0)
w = C = None
import ctypes as C
w = C.c_int32(-1)
mem._L.hagrid_kat_detect_ray_rows(mem._ctx, C.c_void_p(d_rays), n, C.c_float(float(np.linalg.norm(np.array(grid.bbox_max) - np.array(grid.bbox_min)))), C.byref(w))
res["detected_row_length"] = w.value
print(json.dumps(res))
