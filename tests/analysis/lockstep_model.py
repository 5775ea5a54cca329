"""DEV TOOL (CPU, oracle): how much of a wavefront's lock-step iterations are its rays' own cells, per BASELINE configuration.
`traverse_kernel_tail` walks 64 rays in lock step while more than 16 are alive (phase 1: one iteration = one cell step of every live lane), then the
survivors with four lanes each (phase 2).  An iteration ends when the slowest gather of its lanes is back, so a launch that waits for memory
(configuration 5: profiles/NOTES.md "Round 4") costs about  iterations x latency / resident wavefronts  whatever it fetches.  From the oracle's
per-ray traces this script counts, for the wavefronts the kernel forms (8x8 tiles of the image for image-ordered batches, consecutive rays of the
sorted bins for the incoherent one):
  cells/ray            mean cells a ray visits
  phase-1 iterations   the (64 - 16)-th smallest cell count of the wavefront (the iteration at which 16 rays are left)
  phase-2 iterations   longest ray - that
  lock-step efficiency sum of the rays' cells / (64 x phase-1 iterations + 16 x phase-2 iterations)
and the same for wavefronts regrouped IDEALLY: the rays of a super-tile (64 tiles) sorted by their cell counts (a bound on what "rays of similar
length together" can buy -- it ignores that regrouped rays share fewer lines).
usage: python tests/analysis/lockstep_model.py [2|4|5] ; environment N (triangles; default the configuration's) ROWS (image rows of the sample)"""
import os, sys, json, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from hagrid_amd import scene

config = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = int(os.environ.get("N", {2: 1_000_000, 4: 1_000_000, 5: 8_000_000}[config]))
threads = len(os.sched_getaffinity(0))
tris = scene.make_soup(N)
t0 = time.time(); G = O.Grid.full(tris, compress=(config == 5)); print("oracle build", round(time.time() - t0, 1), "s", G.summary(), flush=True)
if config == 2:
    W = 1024; rows = int(os.environ.get("ROWS", 1024)); first_row = (W - rows) // 2
    rays = scene.make_rays_primary(G.bbox_min, G.bbox_max, W, W, first=first_row * W, count=rows * W)
elif config == 5:
    W = 8192; rows = int(os.environ.get("ROWS", 64)); first_row = 3072 + 256          # inside the share of rank 3 of 8
    prim = scene.make_rays_primary(G.bbox_min, G.bbox_max, W, W, first=first_row * W, count=rows * W)
    hits, _ = G.traverse(tris, prim, nthreads=threads)
    rays = scene.make_rays_bounce(tris, prim, hits, G.bbox_min, G.bbox_max, 0x52415953 + 5, first=first_row * W)
else:
    W = 0; n4 = int(os.environ.get("RAYS", 1 << 20))
    rays = scene.make_rays_incoherent(G.bbox_min, G.bbox_max, n4, 0x52415953 + 4)
n = rays.shape[0]
CAP = 255
lens = np.zeros((n, CAP), np.uint8); nc = np.zeros(n, np.int32)
L = O.lib(); L.orc_traverse_trace.argtypes = [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]; L.orc_traverse_trace.restype = None
t0 = time.time()
r = np.ascontiguousarray(rays, dtype=np.float32)
L.orc_traverse_trace(C.byref(G.g), tris.ctypes.data, r.ctypes.data, n, CAP, lens.ctypes.data, nc.ctypes.data, 0, None, None)
print("trace", round(time.time() - t0, 1), "s; rays", n, flush=True)
cells = np.minimum(nc, CAP).astype(np.int64)
tests = lens.astype(np.int64).sum(axis=1)

if W:   # 8x8 tiles of the image, row-major over the tiles (the order inside a launch does not matter here)
    order = np.arange(n).reshape(rows // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
    super_tiles = order.reshape(rows // 8, W // 64, 8 * 64) if W % 64 == 0 else None            # 8 tiles side by side: 512 rays
else:   # the kernel's bins: Morton key of the entry voxel at 512 bins; inside a bin the input order
    g = np.float32(1.0) / (G.bbox_max - G.bbox_min)
    q = np.clip(((r[:, 0:3] - G.bbox_min) * g * 8).astype(np.int64), 0, 7)
    key = np.zeros(n, np.int64)
    for b in range(3):
        for a in range(3): key |= ((q[:, a] >> b) & 1) << (3 * b + a)
    order = np.argsort(key, kind="stable"); order = order[: n // 64 * 64]
    super_tiles = None


def stats(groups):     # groups: [waves, 64] ray indices
    c = np.sort(cells[groups], axis=1)
    p1 = c[:, 64 - 16 - 1]                      # iterations with more than 16 rays alive
    p2 = c[:, -1] - p1
    own = c.sum(axis=1)
    key = 2 * p1 + p2                            # what the kernel leaves at its tile for the learned order (iterations, those of phase 1 twice)
    q = np.percentile(key, [5, 25, 50, 75, 95, 99, 100])
    spread = {"tile cost percentiles 5/25/50/75/95/99/max": [round(float(v), 1) for v in q], "p95 / median": round(float(q[4] / max(q[2], 1e-9)), 2),
              "tiles that cost nothing": round(float((key == 0).mean()), 3), "coefficient of variation": round(float(key.std() / max(key.mean(), 1e-9)), 3)}
    return {"tile costs": spread, "wavefronts": int(groups.shape[0]), "cells/ray": round(float(own.sum() / groups.size), 2), "tests/ray": round(float(tests[groups].sum() / groups.size), 2),
            "phase-1 iterations": round(float(p1.mean()), 2), "phase-2 iterations": round(float(p2.mean()), 2), "longest ray of a wavefront": round(float(c[:, -1].mean()), 2),
            "lock-step efficiency": round(float(own.sum() / (64 * p1.sum() + 16 * p2.sum())), 3),
            "iterations per wavefront (p1 + p2)": round(float((p1 + p2).mean()), 2)}


def refill_sim(K, refill_min=16, tail=16):
    """The prototype of tools/proto/refill.patch on the traces: a wavefront owns K consecutive tiles; while the pool is not empty and at least `refill_min`
    lanes are idle they take its next rays, which JOIN one iteration later; with the pool empty and at most `tail` rays alive the rest runs with four lanes
    per ray.  Returns lock-step iterations per ray-tile (so that K = 1 is the kernel as it is) and the lane efficiency of phase 1."""
    tiles = order.reshape(-1, 64)
    nt = tiles.shape[0] // K * K
    c = cells[tiles[:nt]].reshape(-1, K * 64)
    it1 = it2 = 0; busy = 0
    for wave in c:
        left = wave[:64].copy(); nxt = 64; joining = None
        while True:
            alive = left > 0
            na = int(alive.sum())
            if nxt < wave.size and 64 - na >= refill_min:
                idle = np.flatnonzero(~alive)
                take = idle[: min(idle.size, wave.size - nxt)]
                joining = (take, wave[nxt: nxt + take.size].copy()); nxt += take.size
            if na <= tail and nxt >= wave.size and joining is None:
                break
            left[alive] -= 1; busy += na; it1 += 1
            if joining is not None:
                left[joining[0]] = joining[1]; joining = None
        it2 += int(left.max()) if left.size else 0
    return {"tiles per wavefront": K, "phase-1 iterations per tile": round(it1 / nt, 2), "phase-2 iterations per tile": round(it2 / nt, 2),
            "iterations per tile": round((it1 + it2) / nt, 2), "phase-1 lane efficiency": round(busy / (64 * max(it1, 1)), 3)}


out = {"config": config, "triangles": N, "rays": int(n), "as the kernel groups them": stats(order.reshape(-1, 64))}
if os.environ.get("REFILL", "1") != "0":
    sample = int(os.environ.get("SIM_TILES", 4096))       # (a python loop per wavefront: a sample of the tiles is enough)
    keep_order = order
    order = order[: min(order.size, sample * 64)]
    out["refill prototype, simulated on the traces"] = [refill_sim(K) for K in (1, 2, 3, 4, 8)]
    out["refill prototype, 32 idle lanes before a refill"] = [refill_sim(K, refill_min=32) for K in (2, 8)]
    order = keep_order
if W and super_tiles is not None:
    st = super_tiles.reshape(-1, 512)
    regrouped = np.take_along_axis(st, np.argsort(cells[st], axis=1, kind="stable"), axis=1).reshape(-1, 64)
    out["rays of 8 neighbouring tiles sorted by length (bound)"] = stats(regrouped)
glob = order[np.argsort(cells[order], kind="stable")]
out["all rays sorted by length (bound)"] = stats(glob[: glob.size // 64 * 64].reshape(-1, 64))
print(json.dumps(out, indent=1))
