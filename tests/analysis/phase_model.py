"""DEV TOOL (CPU, oracle): how much lock-step work a wavefront of the 1M-primary batch could shed if its lanes did not wait
for each other at cell boundaries (decoupled cell-step / triangle-test phases, greedy majority vote per iteration)."""
import os, sys, json, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from hagrid_amd import scene
N = 1000000; W = 1024; SUB = int(os.environ.get("SUB", 8))
tris = scene.make_soup(N)
G = O.Grid.full(tris)
rays = scene.make_rays_primary(G.bbox_min, G.bbox_max, W, W)
rows = np.arange(W).reshape(-1, 8)[::SUB].reshape(-1)
idx = (rows[:, None] * W + np.arange(W)[None, :]).reshape(-1)
r = np.ascontiguousarray(rays[idx]); n = r.shape[0]
CAP = 320
lens = np.zeros((n, CAP), np.uint8); nc = np.zeros(n, np.int32)
L = O.lib(); L.orc_traverse_trace.argtypes = [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]; L.orc_traverse_trace.restype = None
ids = np.full((n, 1), -1, np.int32); nids = np.zeros(n, np.int32)
L.orc_traverse_trace(C.byref(G.g), tris.ctypes.data, r.ctypes.data, n, CAP, lens.ctypes.data, nc.ctypes.data, 1, ids.ctypes.data, nids.ctypes.data)
bands = n // (8 * W)
tile = np.arange(n).reshape(bands, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1).reshape(-1, 64)
C_CELL, C_TRI = 90, 60
now = []; bound = []; vote = []; both = []
for w in tile[:: max(1, len(tile) // 600)]:
    ln = lens[w].astype(np.int64); c = nc[w]
    live = np.arange(CAP)[None, :] < c[:, None]
    cell_it = int(live.any(axis=0).sum()); tri_it = int(np.where(live, ln, 0).max(axis=0).sum())
    now.append(C_CELL * cell_it + C_TRI * tri_it)
    bound.append(int((C_CELL * c + C_TRI * np.where(live, ln, 0).sum(axis=1)).max()))
    # per-lane programs: sequence of ops: for each cell: one S then n T's
    prog = []
    for l in range(64):
        p = []
        for s in range(c[l]): p.append(0); p.extend([1] * int(ln[l, s]))
        prog.append(p)
    pc = [0] * 64
    cost_v = 0; cost_b = 0
    while True:
        want = [prog[l][pc[l]] for l in range(64) if pc[l] < len(prog[l])]
        if not want: break
        nS = want.count(0); nT = len(want) - nS
        # majority vote: run one phase
        ph = 0 if nS >= nT else 1
        cost_v += C_CELL if ph == 0 else C_TRI
        for l in range(64):
            if pc[l] < len(prog[l]) and prog[l][pc[l]] == ph: pc[l] += 1
    vote.append(cost_v)
    pc = [0] * 64
    while True:
        want = [prog[l][pc[l]] for l in range(64) if pc[l] < len(prog[l])]
        if not want: break
        for ph in (0, 1):                                 # both phases every iteration, if anyone wants them
            if any(pc[l] < len(prog[l]) and prog[l][pc[l]] == ph for l in range(64)):
                cost_b += C_CELL if ph == 0 else C_TRI
                for l in range(64):
                    if pc[l] < len(prog[l]) and prog[l][pc[l]] == ph: pc[l] += 1
    both.append(cost_b)
now, bound, vote, both = map(np.array, (now, bound, vote, both))
print(json.dumps({"waves sampled": len(now), "VALU/wave now (barrier at every cell)": float(now.mean()), "longest lane alone (lower bound)": float(bound.mean()),
                  "majority vote, one phase per iteration": float(vote.mean()), "both phases per iteration": float(both.mean())}))
