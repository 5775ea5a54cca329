"""DEV TOOL: traversal time of the standard batches on the 1M-triangle scene (median of 21 event-timed launches each), with a
checksum of the hits so that two builds can be compared line by line."""
import json, os, sys, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
for kv in filter(None, os.environ.get("OPTS", "").split(",")):
    k, v = kv.split("="); mem.set_option(k, int(v))
tris = scene.make_soup(1_000_000); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, tris.shape[0])
api.setup_traversal(grid)
batches = [("primary 1024^2", lambda: scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 1024), 0),
           ("primary 2048^2", lambda: scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 2048, 2048), 0),
           ("primary 4096^2", lambda: scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 4096, 4096), 0),
           ("incoherent 1M", lambda: scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 1 << 20, scene.RAY_SEED_BASE + 4), 0),
           ("incoherent 4M binned", lambda: scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 1 << 22, scene.RAY_SEED_BASE + 4), 1)]
only = os.environ.get("BATCH")
flags = int(os.environ.get("FLAGS", "0"))          # api.ANY_HIT = 1, api.UVS = 2
for name, gen, binning in batches:
    if only and name not in only.split(";"): continue
    rays = gen(); n = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    mem.set_ray_binning(binning)
    for _ in range(3): api.traverse_grid(grid, d_tris, d_rays, d_hits, n, flags)
    t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n, flags), mem) for _ in range(21))
    h = mem.download(d_hits, api.HIT_DTYPE, n)
    print(json.dumps({"batch": name, "flags": flags, "ms_median": round(t[10], 4), "ms_min": round(t[0], 4), "Grays/s": round(n / t[10] / 1e6, 2),
                      "hits_crc": zlib.crc32(h.tobytes())}), flush=True)
    mem.set_ray_binning(0); mem.free(d_rays); mem.free(d_hits)
