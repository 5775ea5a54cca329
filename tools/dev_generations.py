"""DEV TOOL: the generational traversal (traverse.hip) against the one-pass image kernel, and a sweep of generation
schedules, on the headline batch and on the other ray kinds.  Prints ms (median of 15) and checks that hits are identical."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene

def sched(*l):
    v = 0
    for i, x in enumerate(l): v |= x << (6 * i)
    return v

mem = api.MemManager(keep=True)
tris = scene.make_soup(1_000_000); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, tris.shape[0])
api.setup_traversal(grid)
batches = {"primary 1024^2": scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 1024),
           "primary 2048^2": scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 2048, 2048),
           "incoherent 1M": scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 1 << 20, scene.RAY_SEED_BASE + 4)}
scheds = [("one pass", None), ("8,8,16,32", sched(8, 8, 16, 32)), ("8,16,32", sched(8, 16, 32)), ("6,6,12,24,48", sched(6, 6, 12, 24, 48)),
          ("4,4,8,16,32", sched(4, 4, 8, 16, 32)), ("12,12,24", sched(12, 12, 24)), ("10,20", sched(10, 20)), ("16,32", sched(16, 32)), ("8", sched(8)), ("16", sched(16)),
          ("5,5,10,20,40", sched(5, 5, 10, 20, 40)), ("8,8,8,16,32", sched(8, 8, 8, 16, 32))]
only = os.environ.get("SCHED")
if only: scheds = [x for x in scheds if x[0] in only.split(";")]
if os.environ.get("BATCH"): batches = {k: v for k, v in batches.items() if k in os.environ["BATCH"].split(";")}
for name, rays in batches.items():
    n = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    for binning in ((0, 1) if name.startswith("incoherent") else (0,)):
        mem.set_ray_binning(binning)
        ref = None
        for sname, sv in scheds:
            mem.set_option("traverse.generations", 0 if sv is None else 1)
            if sv is not None: mem.set_option("traverse.gen_schedule", sv)
            for _ in range(3): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
            t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n), mem) for _ in range(15))
            h = mem.download(d_hits, api.HIT_DTYPE, n)
            if ref is None: ref = h
            same = bool((h["id"] == ref["id"]).all() and (h["t"].view(np.uint32) == ref["t"].view(np.uint32)).all())
            print(json.dumps({"batch": name, "binning": binning, "schedule": sname, "ms_median": round(t[7], 4), "ms_min": round(t[0], 4), "Grays/s": round(n / t[7] / 1e6, 2), "identical": same}), flush=True)
    mem.set_ray_binning(0)
    mem.free(d_rays); mem.free(d_hits)
