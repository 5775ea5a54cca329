"""DEV TOOL (CPU, oracle): lock-step structure of the headline batch as the tail kernel runs it (8x8 tiles, one ray per lane while more
than 16 rays of a wavefront are live): wave-iterations, triangle rounds per iteration (= longest list among the live lanes), and what
two ids per round trip would leave."""
import os, sys, json, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from hagrid_amd import scene

N = int(os.environ.get("N", 1000000)); W = int(os.environ.get("W", 1024)); SUB = int(os.environ.get("SUB", 8))
tris = scene.make_soup(N)
G = O.Grid.full(tris)
rays = scene.make_rays_primary(G.bbox_min, G.bbox_max, W, W)
rows = np.arange(W).reshape(-1, 8)[::SUB].reshape(-1)
idx = (rows[:, None] * W + np.arange(W)[None, :]).reshape(-1)
r = np.ascontiguousarray(rays[idx]); n = r.shape[0]
CAP, ICAP = 320, 640
lens = np.zeros((n, CAP), np.uint8); nc = np.zeros(n, np.int32)
ids = np.full((n, ICAP), -1, np.int32); nids = np.zeros(n, np.int32)
L = O.lib(); L.orc_traverse_trace.argtypes = [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]; L.orc_traverse_trace.restype = None
L.orc_traverse_trace(C.byref(G.g), tris.ctypes.data, r.ctypes.data, n, CAP, lens.ctypes.data, nc.ctypes.data, ICAP, ids.ctypes.data, nids.ctypes.data)
bands = n // (8 * W)
tile = np.arange(n).reshape(bands, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
ln = lens[tile].astype(np.int32)                     # wave, lane, cell
cnt = nc[tile]                                       # wave, lane
steps = np.arange(CAP)[None, None, :]
live = steps < cnt[:, :, None]
nlive = live.sum(axis=1)                             # wave, cell
p1 = nlive > 16                                      # iterations of phase 1
mx = np.where(live, ln, 0).max(axis=1)               # longest list among the live lanes
waves = tile.shape[0]
out = {"waves": waves, "iterations/wave": float((nlive > 0).sum() / waves), "phase-1 iterations/wave": float(p1.sum() / waves),
       "phase-2 iterations/wave": float(((nlive > 0) & ~p1).sum() / waves)}
hist = np.bincount(np.minimum(mx[p1], 6), minlength=7)
out["phase 1: longest list of an iteration (0..5, 6+), share"] = [round(float(h) / p1.sum(), 3) for h in hist]
one = np.where(mx > 4, mx, mx)[p1].sum()
two = np.where(mx > 4, mx, (mx + 1) // 2)[p1].sum()   # lists by index (> 4 ids) keep one id per round
out["phase 1: triangle rounds/wave, one id per round"] = float(one / waves)
out["phase 1: triangle rounds/wave, two ids per round"] = float(two / waves)
out["phase 1: live lanes per iteration"] = float(nlive[p1].mean())
lanes_testing = (np.where(live, ln, 0) > 0).sum(axis=1)
out["phase 1: lanes with a non-empty list per iteration"] = float(lanes_testing[p1].mean())
print(json.dumps(out, indent=1))
