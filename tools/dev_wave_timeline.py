"""DEV TOOL: timeline of the wavefronts of one traversal launch (table-free image kernel, hagrid_kat_traverse_timed of libhagrid_amd_kat.so): when every
wavefront starts and ends (100 MHz wall clock), how many are resident over time, which ones end last."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
N = int(os.environ.get("N", 1000000)); W = int(os.environ.get("W", 1024))
mem = api.MemManager(keep=True)
for kv in filter(None, os.environ.get("OPTS", "").split(",")):
    k, v = kv.split("="); mem.set_option(k, int(v))
tris = scene.make_soup(N); d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, N)
rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, W, W); n = rays.shape[0]
d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n); d_steps = mem.alloc(4 * n)
api.setup_traversal(grid)
api.traverse_grid_stats(grid, d_tris, d_rays, d_hits, n, d_steps); steps = mem.download(d_steps, np.int32, n)
for _ in range(3): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
print(json.dumps({"plain launch ms": round(sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)) for _ in range(9))[4], 4)}))
nw = (n + 63) // 64
d_times = mem.alloc(16 * nw)
def run(order=None, label=""):
    d_order = mem.upload(np.ascontiguousarray(order, np.int32)) if order is not None else None
    out = None
    for rep in range(3):
        mem.zero(d_times, 16 * nw)
        import ctypes as C
        tail = 0 if "traverse.tail=0" in os.environ.get("OPTS", "") else 1
        ms = api.profile(lambda: api._check(mem, mem._K.hagrid_kat_traverse_timed(mem._ctx, C.byref(grid.pod), d_tris, d_rays, d_hits, n, W, tail, d_times, d_order), "kat_traverse_timed"))
        t = mem.download(d_times, np.uint64, 2 * nw).reshape(nw, 2).astype(np.int64)
        t0 = t[:, 0].min(); s = (t[:, 0] - t0) / 100.0; e = (t[:, 1] - t0) / 100.0        # microseconds
        dur = e - s
        grid_t = np.arange(0, e.max() + 10, 10.0)
        resident = [(int(((s <= x) & (e > x)).sum())) for x in grid_t]
        print(json.dumps({"order": label, "rep": rep, "launch ms (event)": round(ms, 4), "span us": round(float(e.max()), 1), "last start us": round(float(s.max()), 1),
                          "duration us mean/p50/p90/p99/max": [round(float(x), 1) for x in (dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max())],
                          "mean occupancy (of 8192 slots)": round(float(dur.sum() / e.max() / 8192), 3), "resident every 10 us": resident}), flush=True)
        out = dur
    if d_order: mem.free(d_order)
    h = mem.download(d_hits, api.HIT_DTYPE, n)
    return out, h
dur0, h0 = run(None, "default")
# the packet's position b -> duration of the tile it processed; LPT with the measured durations (an upper bound on what ordering can give)
lpt = np.argsort(-dur0, kind="stable")
dur1, h1 = run(lpt, "longest measured duration first")
assert (h1["id"] == h0["id"]).all() and (h1["t"].view(np.uint32) == h0["t"].view(np.uint32)).all()
# coarse: 8 classes of the measured duration, default order inside a class
cls = np.minimum((dur0 / dur0.max() * 8).astype(np.int32), 7)
dur2, h2 = run(np.argsort(-cls, kind="stable"), "8 duration classes, default order inside")
print(json.dumps({"corr(duration default, duration LPT run) by tile": float(np.corrcoef(dur0[lpt], dur1)[0, 1])}))
# LPT by the INTRINSIC work of a tile (reference step counts of its rays: cells + tests), not by a measured duration
S = 8; tx_n = W // 8
b = np.arange(nw); band = b // (tx_n * S); in_band = b - band * tx_n * S; col = in_band // (S * S); in_super = in_band - col * S * S
def compact(v):
    v = v & 0x55555555; v = (v | (v >> 1)) & 0x33333333; v = (v | (v >> 2)) & 0x0f0f0f0f; v = (v | (v >> 4)) & 0x00ff00ff; v = (v | (v >> 8)) & 0x0000ffff
    return v
tx = compact(in_super); ty = compact(in_super >> 1)
px = ((col * S + tx) << 3)[:, None] + (np.arange(64) & 7)[None, :]; py = ((band * S + ty) << 3)[:, None] + (np.arange(64) >> 3)[None, :]
tile_steps = steps[py * W + px]
for label, key in (("longest ray of the tile first", tile_steps.max(axis=1)), ("most work (sum of steps) first", tile_steps.sum(axis=1)),
                   ("16th longest ray first (what the one-ray-per-lane phase leaves)", np.sort(tile_steps, axis=1)[:, -16])):
    d, h = run(np.argsort(-key, kind="stable"), label)
    assert (h["id"] == h0["id"]).all()
    print(json.dumps({"order": label, "corr(key, duration default order)": float(np.corrcoef(key, dur0)[0, 1])}))
