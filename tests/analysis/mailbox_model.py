"""DEV TOOL (CPU, oracle): how many triangle tests of the headline batch repeat a test the same ray made a few cells earlier?
A triangle spans several cells (5.6 references per triangle in the 1M soup) and a ray that passes it crosses several of them.  Repeating
a test cannot change (id, t): a rejected triangle stays rejected (tmax only falls) and an accepted one writes the same values again.
Model: per ray an LRU "mailbox" of the last K distinct ids tested; lock-step rounds of a wave-step = max over live lanes of the list
length after the mailbox filter (8x8 tiles as the kernel forms them)."""
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
assert nids.max() < ICAP and nc.max() < CAP
total = int(nids.sum())
print(json.dumps({"rays": n, "cells/ray": float(nc.mean()), "tests/ray": total / n}), flush=True)
bands = n // (8 * W)
tile = np.arange(n).reshape(bands, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
for K in (0, 1, 2, 3, 4, 6, 8):
    kept = np.zeros((n, CAP), np.uint8)                       # list length per visited cell after the filter
    saved = 0
    for i in range(n):
        box = []
        at = 0
        row = lens[i]
        for c in range(nc[i]):
            ln = int(row[c]); k = 0
            for j in range(ln):
                t = int(ids[i, at + j])
                if t in box:
                    box.remove(t); box.append(t)
                else:
                    k += 1
                    if K:
                        box.append(t)
                        if len(box) > K: box.pop(0)
            at += ln
            kept[i, c] = k
        saved += int(nids[i]) - int(kept[i].sum())
    w = tile.reshape(-1, 64)
    rounds = kept[w].max(axis=1).sum() / w.shape[0]
    lane_tests = kept.sum() / n
    print(json.dumps({"mailbox entries": K, "tests/ray": round(float(lane_tests), 2), "repeated": round(saved / total, 3), "tri_rounds/wave": round(float(rounds), 1)}), flush=True)
