"""DEV TOOL (CPU, oracle): the non-uniform scene of tools/dev_nonuniform.py -- list lengths along the rays of the heaviest
8x8 pixel tiles (what a lone wavefront has to do there)."""
import os, sys, json, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from hagrid_amd import scene
tris = scene.make_clustered()
t0 = time.time(); G = O.Grid.full(tris); print("oracle build", round(time.time() - t0, 1), "s", G.summary(), flush=True)
rays = scene.make_rays_primary(G.bbox_min, G.bbox_max, 1024, 1024).reshape(128, 8, 128, 8, 8)
L = O.lib(); L.orc_traverse_trace.argtypes = [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]; L.orc_traverse_trace.restype = None
CAP = 512
for ty, tx in ((81, 94), (82, 93), (89, 93), (31, 32)):
    r = np.ascontiguousarray(rays[ty, :, tx, :].reshape(64, 8)); n = 64
    lens = np.zeros((n, CAP), np.uint8); nc = np.zeros(n, np.int32); ids = np.full((n, 8), -1, np.int32); nids = np.zeros(n, np.int32)
    L.orc_traverse_trace(C.byref(G.g), tris.ctypes.data, r.ctypes.data, n, CAP, lens.ctypes.data, nc.ctypes.data, 8, ids.ctypes.data, nids.ctypes.data)
    live = np.arange(CAP)[None, :] < nc[:, None]
    l = np.where(live, lens, 0).astype(np.int64)
    lock = l.max(axis=0)                     # lock-step: every step costs the longest list among the lanes
    print(json.dumps({"tile": [ty, tx], "cells max": int(nc.max()), "cells mean": float(nc.mean()), "refs per ray max": int(l.sum(axis=1).max()),
                      "lock-step triangle iterations": int(lock.sum()), "steps with a list > 4 (any lane)": int((lock > 4).sum()),
                      "longest list": int(l.max()), "iterations in lists > 4": int(lock[lock > 4].sum()),
                      "hist of per-step max list": np.bincount(np.minimum(lock[: nc.max()], 16), minlength=17).tolist()}), flush=True)
