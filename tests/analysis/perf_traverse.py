"""DEV TOOL (not the bench): times the traversal kernel on an ORACLE-built grid uploaded to the GPU.
Used before the GPU construction passes existed and to A/B kernel variants on a fixed grid."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hagrid_amd import api, scene
from oracle import oracle as O

N = int(os.environ.get("N", 1000000)); iters = int(os.environ.get("ITERS", 20))
tris = scene.make_soup(N)
t = time.time(); G = O.Grid.full(tris); print("oracle build s", time.time() - t, G.summary(), flush=True)
mem = api.MemManager(keep=True)
print(mem.device_info())
d_tris = mem.upload(tris)
for compress in (False, True):
    if compress: G.compress()
    grid = api.Grid.upload(mem, G.entries, G.ref_ids, G.cells, G.small_cells, G.bbox_min, G.bbox_max, G.dims, G.shift, G.offsets)
    for kind in ("primary", "incoherent"):
        if kind == "primary": rays = scene.make_rays_primary(G.bbox_min, G.bbox_max, 1024, 1024)
        else: rays = scene.make_rays_incoherent(G.bbox_min, G.bbox_max, 1 << 20, scene.RAY_SEED_BASE + 4)
        n = rays.shape[0]
        d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
        for _ in range(3): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        ms = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)) for _ in range(iters))
        st = api.traverse_grid_stats(grid, d_tris, d_rays, d_hits, n)
        ab = api.algorithmic_bytes(st, compress)
        hits = mem.download(d_hits, api.HIT_DTYPE, n)
        oh, _ = G.traverse(tris, rays[:65536], nthreads=8)
        ok = bool((hits["id"][:65536] == oh["id"]).all() and (hits["t"][:65536].view(np.uint32) == oh["t"].view(np.uint32)).all())
        print(json.dumps({"compress": compress, "rays": kind, "n": n, "ms_min": ms[0], "ms_med": ms[len(ms) // 2],
                          "mrays_med": n / ms[len(ms) // 2] / 1e3, "GBps_alg": ab["B_ray"] / ms[len(ms) // 2] / 1e6,
                          "GBps_walk": ab["B_walk"] / ms[len(ms) // 2] / 1e6, "parity64k": ok, "stats": st}), flush=True)
        mem.free(d_rays); mem.free(d_hits)
    grid.free()
