"""DEV TOOL: how much does the memory order of the triangles matter?  The same soup with its triangles in random (generator)
order and sorted along a Morton curve of their first vertex (ids differ, hits equal up to the renaming)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
N = 1000000
tris0 = scene.make_soup(N)
def morton(p):
    q = np.clip((p * 1024).astype(np.int64), 0, 1023)
    def spread(v):
        v = (v | (v << 16)) & 0x30000ff; v = (v | (v << 8)) & 0x300f00f; v = (v | (v << 4)) & 0x30c30c3; v = (v | (v << 2)) & 0x9249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
order = np.argsort(morton(tris0[:, 0:3]), kind="stable")
for name, tris in (("generator order", tris0), ("morton order", np.ascontiguousarray(tris0[order]))):
    d_tris = mem.upload(tris)
    grid = api.build_all(mem, d_tris, N); api.setup_traversal(grid)
    res = {"triangles": name}
    for label, rays in (("primary1M", scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 1024)),
                        ("incoh1M", scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 1 << 20, 9)),
                        ("primary16M", scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 4096, 4096))):
        n = rays.shape[0]
        d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
        for _ in range(2): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)) for _ in range(9))
        res[label] = round(t[4], 4)
        mem.free(d_rays); mem.free(d_hits)
    print(json.dumps(res), flush=True)
    grid.free(); mem.free(d_tris)
