"""DEV TOOL: one 8x8 pixel tile of the non-uniform scene alone (for PMC runs): python tools/dev_nonuniform_tile.py TY TX"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ty, tx = int(sys.argv[1]), int(sys.argv[2]); sys.argv = ["x"]
import numpy as np
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev_nonuniform.py")).read().split("prim = scene.make_rays_primary")[0])
P = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 1024).reshape(128, 8, 128, 8, 8)
rays = np.ascontiguousarray(P[ty, :, tx, :].reshape(64, 8))
d_rays = mem.upload(rays); d_hits = mem.alloc(16 * 64)
for img in (2, 0):
    mem.set_option("traverse.image", img); api.setup_traversal(grid)
    for _ in range(6): api.traverse_grid(grid, d_tris, d_rays, d_hits, 64)
mem.synchronize() if hasattr(mem, "synchronize") else mem.download(d_hits, api.HIT_DTYPE, 64)
