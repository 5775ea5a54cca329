"""DEV TOOL (GPU): differential fuzz of the traversal kernels.  Random scenes (soups of several densities, coincident and sliver
triangles) and random ray batches (primary tiles, unordered rays, rays that start inside the grid, short tmax, tmin > 0) are traversed
by the tail kernel, by the image kernel without the tail mode and by v2 on the construction format; the three must agree bit for bit
(ids and t).  usage: python tools/dev_fuzz_kernels.py [rounds]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(12345)
mem = api.MemManager(keep=True)
checked = 0
for r in range(rounds):
    n_tris = int(rng.choice([300, 3000, 30000, 120000]))
    tris = scene.make_soup(n_tris, seed=1000 + r).copy()
    if r % 3 == 1:                                   # coincident triangles: equal t, long lists
        tris = np.concatenate([np.repeat(tris[: max(8, n_tris // 50)], 6, axis=0), tris])
    if r % 4 == 2:                                   # slivers
        tris[::7, 8:11] = tris[::7, 4:7] * np.float32(1.0001) + np.float32(1e-6)
        e1, e2 = tris[::7, 4:7].astype(np.float64), tris[::7, 8:11].astype(np.float64)
        nrm = np.cross(e1, e2).astype(np.float32); tris[::7, 3] = nrm[:, 0]; tris[::7, 7] = nrm[:, 1]; tris[::7, 11] = nrm[:, 2]
    tris = np.ascontiguousarray(tris, np.float32)
    params = [dict(), dict(top_density=0.15, snd_density=3.0), dict(top_density=0.3, snd_density=1.0), dict(top_density=0.05, snd_density=6.0)][r % 4]
    d_tris = mem.upload(tris)
    grid = api.build_all(mem, d_tris, tris.shape[0], compress=(r % 5 == 4), **params)
    lo, hi = np.asarray(grid.bbox_min), np.asarray(grid.bbox_max)
    w = int(rng.choice([64, 200, 512]))
    prim = scene.make_rays_primary(lo, hi, w, w)
    inco = scene.make_rays_incoherent(lo - 0.3, hi + 0.3, 150001, 7000 + r)
    inside = scene.make_rays_incoherent(lo + 0.3 * (hi - lo), hi - 0.3 * (hi - lo), 50000, 8000 + r)
    short = inco[:40000].copy(); short[:, 7] = np.float32(0.05) * np.float32(np.linalg.norm(hi - lo)); short[:, 3] = np.float32(0.01)
    rays = np.ascontiguousarray(np.concatenate([prim, inco, inside, short]), np.float32)
    n = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    out = {}
    for name, opts in (("tail", {"traverse.image": 2, "traverse.tail": 1}), ("tail quad 0", {"traverse.quad_tail": 0}), ("tail one id per round", {"traverse.tail_dual": 0}), ("tail two ids per round", {"traverse.tail_dual": 1}), ("tail mailbox", {"traverse.mailbox": 1}), ("tail, no tile order", {"traverse.tile_order": 0}), ("tail quad 40", {"traverse.quad_tail": 40}), ("tail quad 100", {"traverse.quad_tail": 100}), ("img", {"traverse.image": 2, "traverse.tail": 0}),
                       ("wide ids", {"traverse.image_slim": 2}), ("general layout", {"traverse.image_general": 2}), ("v2", {"traverse.image": 0})):
        for k, v in {"traverse.image": 2, "traverse.tail": 1, "traverse.image_slim": 1, "traverse.image_general": 1, "traverse.quad_tail": -1, "traverse.tail_dual": -1, "traverse.tile_order": -1, "traverse.mailbox": -1, **opts}.items(): mem.set_option(k, v)
        for binning in (0, 1):
            mem.set_ray_binning(binning)
            api.setup_traversal(grid)
            if name == "tail" and binning == 0: fmt = mem.image_format(grid)
            for _ in range(14 if name == "tail" else 3):    # (a learned tile order applies from the second launch over a buffer on; the default policy's trials take a dozen)
                api.traverse_grid(grid, d_tris, d_rays, d_hits, n); mem.synchronize()
            out[(name, binning)] = mem.download(d_hits, api.HIT_DTYPE, n).copy()
    ref = out[("v2", 0)]
    for key, h in out.items():
        bad = (h["id"] != ref["id"]) | (h["t"].view(np.uint32) != ref["t"].view(np.uint32))
        if bad.any():
            i = int(np.flatnonzero(bad)[0])
            print(json.dumps({"round": r, "kernel": key, "mismatches": int(bad.sum()), "first": i, "got": [int(h["id"][i]), float(h["t"][i])], "want": [int(ref["id"][i]), float(ref["t"][i])]}))
            sys.exit(1)
    checked += n
    print(json.dumps({"round": r, "tris": int(tris.shape[0]), "shift": grid.shift, "compressed": r % 5 == 4, "rays": n, "hits": int((ref["id"] >= 0).sum()), "image": fmt}), flush=True)
    mem.set_ray_binning(0); mem.free(d_rays); mem.free(d_hits); grid.free(); mem.free(d_tris)
print(json.dumps({"rounds": rounds, "rays checked per kernel": checked, "result": "all kernels agree bit for bit"}))
