"""DEV TOOL: interleaved A/B of one integer option on the GPU-built 1M grid.  usage: python tools/dev_opt_ab.py traverse.narrow 0,1 [rounds]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
key = sys.argv[1]; values = [int(v) for v in sys.argv[2].split(",")]; rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 15
N = int(os.environ.get("N", 1000000))
mem = api.MemManager(keep=True)
tris = scene.make_soup(N); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, N, compress=bool(int(os.environ.get("COMPRESS", "0"))))
api.setup_traversal(grid)
sets = {"primary1M": scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 1024),
        "incoh1M": scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 1 << 20, scene.RAY_SEED_BASE + 4),
        "primary16M": scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 4096, 4096)}
if os.environ.get("SETS"):
    sets = {k: v for k, v in sets.items() if k in os.environ["SETS"].split(",")}
for k, v in (kv.split("=") for kv in os.environ.get("OPTS", "").split(",") if kv):
    mem.set_option(k, int(v))
for name, rays in sets.items():
    n = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    ref = None; times = {v: [] for v in values}
    for r in range(rounds + 1):
        for v in values:
            mem.set_option(key, v)
            ms = api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n))
            if r: times[v].append(ms)
            if r == 1:
                h = mem.download(d_hits, api.HIT_DTYPE, n)
                if ref is None: ref = h
                else: assert (h["id"] == ref["id"]).all() and (h["t"].view(np.uint32) == ref["t"].view(np.uint32)).all(), f"{key}={v} differs"
    for v in values:
        t = sorted(times[v])
        print(json.dumps({"rays": name, key: v, "ms_med": round(t[len(t) // 2], 4), "ms_min": round(t[0], 4), "mrays_med": round(n / t[len(t) // 2] / 1e3, 1)}), flush=True)
    mem.free(d_rays); mem.free(d_hits)
