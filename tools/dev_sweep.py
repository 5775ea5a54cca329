"""DEV TOOL: traversal variants x batch sizes (primary rays of growing images, incoherent)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
N = 1000000
tris = scene.make_soup(N); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, N)
configs = [dict(v=1), dict(v=2), dict(v=3), dict(v=3, BOTH=1), dict(v=3, BOTH=1, WAVES=16), dict(v=3, WAVES=16)]
if len(sys.argv) > 1:
    configs = [json.loads(a) for a in sys.argv[1:]]
SIZES = (("primary", [(1024, 1024), (2048, 2048), (4096, 4096)]), ("incoherent", [(1 << 20, 1), (1 << 22, 1), (1 << 24, 1)]))
if os.environ.get("SWEEP") == "small":
    SIZES = (("primary", [(1024, 1024), (2048, 2048)]),)
for kind, sizes in SIZES:
    for w, h in sizes:
        rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, w, h) if kind == "primary" else scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, w, scene.RAY_SEED_BASE + 4)
        n = rays.shape[0]; d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
        row = {"rays": kind, "n": n}
        for c in configs:
            mem.set_option("traverse.variant", c["v"])
            mem.set_option("traverse.both_phases", c.get("BOTH", 0)); mem.set_option("traverse.waves_per_cu", c.get("WAVES", 32))
            mem.set_option("traverse.refill_at", c.get("REFILL", 12))
            mem.set_option("traverse.chunk", c.get("CHUNK", 0))
            for _ in range(2): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
            t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)) for _ in range(7))
            row["|".join(f"{k}{v}" for k, v in c.items())] = round(n / t[3] / 1e3, 0)
        print(json.dumps(row), flush=True)
        mem.free(d_rays); mem.free(d_hits)
