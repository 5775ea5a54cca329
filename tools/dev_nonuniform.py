"""DEV TOOL: traversal image vs construction format on a very non-uniform scene (dense clusters inside a sparse soup, walls)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
tris = scene.make_clustered(); N = tris.shape[0]
d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, N)
print(json.dumps({"triangles": N, "grid": grid.summary()}), flush=True)
prim = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 1024)
d_r = mem.upload(prim); d_h = mem.alloc(16 * prim.shape[0])
api.traverse_grid(grid, d_tris, d_r, d_h, prim.shape[0]); hp = mem.download(d_h, api.HIT_DTYPE, prim.shape[0]); mem.free(d_r); mem.free(d_h)
in_cluster = hp["id"] >= 100000
print(json.dumps({"primary rays hitting a cluster": int(in_cluster.sum()), "hitting the sparse soup": int(((hp["id"] >= 0) & ~in_cluster).sum()), "missing": int((hp["id"] < 0).sum())}), flush=True)
rows_cluster = np.flatnonzero(in_cluster.reshape(1024, 1024).any(axis=1))
band = prim.reshape(1024, 1024, 8)[rows_cluster.min(): rows_cluster.min() + (len(rows_cluster) // 8) * 8].reshape(-1, 8) if len(rows_cluster) >= 8 else prim[:8192]
other = prim.reshape(1024, 1024, 8)[: max(8, (rows_cluster.min() // 8) * 8)].reshape(-1, 8) if len(rows_cluster) else prim
for label, rays in (("primary1M", prim), ("image rows that see clusters", np.ascontiguousarray(band)), ("image rows above them", np.ascontiguousarray(other)),
                    ("incoh1M", scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 1 << 20, 9))):
    n = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    res = {"rays": label}; ref = None
    for img in (2, 0, 1, 2, 0):
        mem.set_option("traverse.image", img); api.setup_traversal(grid)
        import ctypes as C
        nb = C.c_int64(0)
        if img and mem._L.hagrid_kat_image_records(mem._ctx, C.byref(grid.pod), None, 0, None, C.byref(nb)) == 0: res[f"image{img}_MB"] = round(nb.value / 1e6, 1)
        for _ in range(2): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)) for _ in range(9))
        res[f"image{img}" + ("b" if f"image{img}" in res else "")] = round(t[4], 4)
        h = mem.download(d_hits, api.HIT_DTYPE, n)
        if ref is None: ref = h
        else: assert (h["id"] == ref["id"]).all() and (h["t"].view(np.uint32) == ref["t"].view(np.uint32)).all()
    print(json.dumps(res), flush=True)
    mem.free(d_rays); mem.free(d_hits)
