"""DEV TOOL: 16M incoherent rays with ray binning -- run under rocprofv3 --kernel-trace --stats to see what the binning passes cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
N = 1000000
tris = scene.make_soup(N); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, N)
n = 1 << 24
rays = scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, n, scene.RAY_SEED_BASE + 4)
d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
api.setup_traversal(grid); mem.set_ray_binning(1)
for _ in range(6): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
mem.download(d_hits, api.HIT_DTYPE, 16)
