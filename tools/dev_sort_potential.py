"""DEV TOOL: what finer ray binning could give.  Incoherent rays (random origin + direction, BASELINE config 4) are put in the order of a
Morton key of their origin with BITS bits per axis ON THE HOST and traversed unbinned; compared with the device's 512-bin counting sort."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
N = int(os.environ.get("RAYS", 1 << 24))
mem = api.MemManager(keep=True)
tris = scene.make_soup(1_000_000); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, tris.shape[0])
api.setup_traversal(grid)
rays = scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, N, scene.RAY_SEED_BASE + 4)
d_rays = mem.alloc(32 * N); d_hits = mem.alloc(16 * N)

def spread(v, bits):
    out = np.zeros_like(v, dtype=np.uint64)
    for b in range(bits):
        out |= ((v >> b) & 1).astype(np.uint64) << np.uint64(3 * b)
    return out

def run(label, r, binning):
    mem.copy_h2d(d_rays, np.ascontiguousarray(r))
    mem.set_ray_binning(binning)
    for _ in range(2): api.traverse_grid(grid, d_tris, d_rays, d_hits, N)
    t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, N), mem) for _ in range(7))
    print(json.dumps({"order": label, "rays": N, "ms_median": round(t[3], 3), "Grays/s": round(N / t[3] / 1e6, 2)}), flush=True)

run("as generated, device binning (512 bins)", rays, 1)
run("as generated, no binning", rays, 0)
ext = (grid.bbox_max - grid.bbox_min).astype(np.float32)
for bits in (3, 4, 5, 6, 7):
    q = np.clip(((rays[:, 0:3] - grid.bbox_min) / ext * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    key = spread(q[:, 0], bits) | (spread(q[:, 1], bits) << np.uint64(1)) | (spread(q[:, 2], bits) << np.uint64(2))
    order = np.argsort(key, kind="stable")
    run(f"host-sorted, Morton {bits} bits per axis, no binning", rays[order], 0)
