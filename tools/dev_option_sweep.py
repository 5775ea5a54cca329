"""DEV TOOL: one traversal option swept in ONE process on ONE grid (same box, same clocks): per value the median / min of event-timed
launches in steady state (launches back to back for `settle` ms first) and a checksum of the hits, which must not move.

usage: python tools/dev_option_sweep.py KEY v0,v1,... [--batch "primary 1024^2"] [--reps 3] [OPTS=k=v,... in the environment]
"""
import json, os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene

key = sys.argv[1]; values = [int(v) for v in sys.argv[2].split(",")]
arg = lambda name, default: (sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default)
batch = arg("--batch", "primary 1024^2"); reps = int(arg("--reps", "3")); launches = int(arg("--launches", "200"))
mem = api.MemManager(keep=True)
for kv in filter(None, os.environ.get("OPTS", "").split(",")):
    k, v = kv.split("="); mem.set_option(k, int(v))
td, sd = (0.15, 3.0) if "config3" in batch else (0.12, 2.4)
td, sd = float(os.environ.get("TD", td)), float(os.environ.get("SD", sd))          # (other grids of the same scene: TD=0.15 SD=3.0 primary 1024^2 ...)
def other_scene(name):
    """scene families beside the uniform soup and the six blobs (what do rules fitted on those two do elsewhere?)"""
    if name == "clustered": return scene.make_clustered()
    if name == "clustered2": return scene.make_clustered(400000, 2, 300000)               # two large blobs in a denser soup
    if name == "gradient": return scene.make_gradient()
    if name == "shell": return scene.make_shell()
    if name == "stadium": return scene.make_stadium()
    if name == "soup8m": return scene.make_soup(8_000_000)                               # the scene of configurations 4 and 5
    raise SystemExit("unknown SCENE " + name)
tris = other_scene(os.environ["SCENE"]) if os.environ.get("SCENE") else scene.make_soup(1_000_000); d_tris = mem.upload(tris)          # (SCENE=clustered: the non-uniform scene of bench.py --config clustered)
grid = api.build_all(mem, d_tris, tris.shape[0], top_density=td, snd_density=sd)
api.setup_traversal(grid)
print(json.dumps({"grid": grid.summary(), "image": mem.image_format(grid), "image_mb": round(mem.image_bytes(grid) / 2**20, 1)}), flush=True)
gens = {"primary 1024^2": lambda: scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 1024),
        "primary 2048^2": lambda: scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 2048, 2048),
        "primary 4096^2": lambda: scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 4096, 4096),
        "config3 1024^2": lambda: scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 1024),
        "config3 4096^2": lambda: scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 4096, 4096),
        "incoherent 4M binned": lambda: scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 1 << 22, scene.RAY_SEED_BASE + 4),
        "incoherent 16M binned": lambda: scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 1 << 24, scene.RAY_SEED_BASE + 4),       # the per-GPU share of configuration 4
        "incoherent 64M binned": lambda: scene.generate_parallel(lambda f, c: scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, c, scene.RAY_SEED_BASE + 4, first=f), 0, 1 << 26, chunk=1 << 22),
        "aimed 1M": lambda: scene.make_rays_aimed(grid.bbox_min, grid.bbox_max, 1 << 20, 5),
        "incoherent 1M": lambda: scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 1 << 20, scene.RAY_SEED_BASE + 4)}
import re
m = re.match(r"primary (\d+)x(\d+)$", batch)
rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, int(m.group(1)), int(m.group(2))) if m else gens[batch](); n = rays.shape[0]
d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
if "binned" in batch: mem.set_ray_binning(1)
go = lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
for rep in range(reps):
    for v in values:
        mem.set_option(key, v)
        mem.forget_hints()
        t0 = time.time()
        while time.time() - t0 < 0.15:
            for _ in range(20): go()
            mem.synchronize()
        mem.zero(d_hits, 16 * n)
        # steady state: K launches between one pair of events
        ms = sorted(api.profile(lambda: [go() for _ in range(launches // 10)], mem) / (launches // 10) for _ in range(10))
        h = mem.download(d_hits, api.HIT_DTYPE, n)
        print(json.dumps({"rep": rep, key: v, "batch": batch, "ms_median": round(ms[5], 5), "ms_min": round(ms[0], 5),
                          "hits_crc": zlib.crc32(h.tobytes())}), flush=True)
