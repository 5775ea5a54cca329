"""DEV TOOL: host time of ONE hagrid_traverse_grid call on an idle stream (what the policy's bookkeeping costs a frame loop before the launch is queued): 4 us, round 6."""
import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
tris = scene.make_clustered(); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, tris.shape[0]); api.setup_traversal(grid)
W = 1024; n = W * W
rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, W, W)
d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
for mode in ("policy", "default_order"):
    mem.set_option("traverse.tile_order", -1 if mode == "policy" else 0)
    for _ in range(300): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    mem.synchronize()
    # host time of a call on an idle stream (the call returns once the launch is queued)
    ts = []
    for _ in range(200):
        mem.synchronize()
        t0 = time.perf_counter(); api.traverse_grid(grid, d_tris, d_rays, d_hits, n); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(mode, "host us per call: median %.1f  min %.1f  p90 %.1f" % (1e6 * ts[100], 1e6 * ts[0], 1e6 * ts[180]))
t0 = time.perf_counter()
for _ in range(2000): mem.usage()
print("ctypes call overhead us: %.2f" % ((time.perf_counter() - t0) / 2000 * 1e6))
