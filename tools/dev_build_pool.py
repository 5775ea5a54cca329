"""DEV TOOL: does the pool settle?  N constructions in keep mode: time of build_grid alone and of the whole construction, pool bytes after each (a build whose
requests the cached slots do not fit pays hipFree + hipMalloc: the bytes then change from build to build)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
n = int(os.environ.get("TRIS", 1_000_000)); iters = int(os.environ.get("ITERS", 24))
mem = api.MemManager(keep=True)
for kv in filter(None, os.environ.get("OPTS", "").split(",")):
    k, v = kv.split("="); mem.set_option(k, int(v))
tris = scene.make_soup(n); d_tris = mem.upload(tris)
grid = api.Grid(); rows = []
for it in range(iters):
    grid.free(); grid = api.Grid()
    if it % 2 == 0:
        tb = api.profile(lambda: api.build_grid(mem, d_tris, n, grid, 0.12, 2.4), mem)
        api.merge_grid(mem, grid, 0.995); api.flatten_grid(mem, grid); api.expand_grid(mem, grid, d_tris, 3)
        rows.append(("build_grid", round(tb, 3), mem.usage() >> 20))
    else:
        ta = api.profile(lambda: api.build_all(mem, d_tris, n, grid=grid), mem)
        rows.append(("all", round(ta, 3), mem.usage() >> 20))
print(" ".join(f"{k}:{t}/{u}MB" for k, t, u in rows))
