"""DEV TOOL: construction time of the 1M-triangle grid (mean of N event-timed builds, keep mode), optionally with
build options; run under tools/gpu_prof_cmd.sh for the per-kernel table."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
n = int(os.environ.get("TRIS", 1_000_000)); iters = int(os.environ.get("ITERS", 5))
mem = api.MemManager(keep=True)
for kv in filter(None, os.environ.get("OPTS", "").split(",")):
    k, v = kv.split("="); mem.set_option(k, int(v))
SCENE = os.environ.get("SCENE", "")          # "" = soup-N, or clustered | gradient | shell | stadium
tris = scene.make_soup(n) if not SCENE else getattr(scene, "make_" + SCENE)(); n = tris.shape[0]; d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, n)
t = []
stages = {"build": [], "merge": [], "flatten": [], "expand": []}
for _ in range(iters):
    grid.free()
    g = api.Grid()
    stages["build"].append(api.profile(lambda: api.build_grid(mem, d_tris, n, g, 0.12, 2.4), mem))
    stages["merge"].append(api.profile(lambda: api.merge_grid(mem, g, 0.995), mem))
    stages["flatten"].append(api.profile(lambda: api.flatten_grid(mem, g), mem))
    stages["expand"].append(api.profile(lambda: api.expand_grid(mem, g, d_tris, 3), mem))
    grid = g
for _ in range(iters):
    grid.free(); t.append(api.profile(lambda: api.build_all(mem, d_tris, n, grid=grid), mem))
print(json.dumps({"tris": n, "build_ms_mean": round(float(np.mean(t)), 3), "median": round(float(np.median(t)), 3), "min": round(min(t), 3),
                  "stages_ms": {k: round(float(np.mean(v)), 3) for k, v in stages.items()}, "grid": grid.summary()}))
