"""DEV TOOL (CPU, oracle): would a FINISHED lane helping a fixed partner lane shorten the lock-step triangle rounds of the image kernel?
Model: 8x8 tiles (lane = y * 8 + x); a lane whose own ray is done adopts the ray of lane ^ P and tests every second id of that
lane's inline list, so the partner needs ceil(L / 2) rounds instead of L.  Rounds of a wave-step = max over its live lanes."""
import os, sys, json, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from hagrid_amd import scene

N = int(os.environ.get("N", 1000000)); W = int(os.environ.get("W", 1024)); SUB = int(os.environ.get("SUB", 4))
tris = scene.make_soup(N)
G = O.Grid.full(tris)
rays = scene.make_rays_primary(G.bbox_min, G.bbox_max, W, W)
rows = np.arange(W).reshape(-1, 8)[::SUB].reshape(-1)
idx = (rows[:, None] * W + np.arange(W)[None, :]).reshape(-1)
r = np.ascontiguousarray(rays[idx]); n = r.shape[0]
CAP = 320
lens = np.zeros((n, CAP), np.uint8); nc = np.zeros(n, np.int32)
L = O.lib(); L.orc_traverse_trace.argtypes = [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]; L.orc_traverse_trace.restype = None
ids = np.full((n, 8), -1, np.int32); nids = np.zeros(n, np.int32)
L.orc_traverse_trace(C.byref(G.g), tris.ctypes.data, r.ctypes.data, n, CAP, lens.ctypes.data, nc.ctypes.data, 8, ids.ctypes.data, nids.ctypes.data)
bands = n // (8 * W)
tile = np.arange(n).reshape(bands, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
w = tile.reshape(-1, 64)
lw = lens[w].astype(np.int64)                                  # [waves, 64, CAP]
live = (np.arange(CAP)[None, None, :] < nc[w][:, :, None])
cell_iters = live.any(axis=1).sum()
base = lw.max(axis=1).sum()
print(json.dumps({"waves": int(w.shape[0]), "cell_iters/wave": cell_iters / w.shape[0], "tri_rounds/wave": base / w.shape[0]}), flush=True)
lanes = np.arange(64)
for P in (1, 8, 2, 16, 4, 32, 9, 63):
    partner_dead = ~live[:, lanes ^ P, :]
    rounds = np.where(partner_dead, (lw + 1) // 2, lw)
    helped = rounds.max(axis=1).sum()
    # helpers that are idle THIS step (dead, or in an empty cell) -- needs the partner's ray every step (more registers)
    idle = partner_dead | (lw[:, lanes ^ P, :] == 0)
    rounds2 = np.where(idle, (lw + 1) // 2, lw)
    print(json.dumps({"partner": f"lane ^ {P}", "tri_rounds/wave dead helper": helped / w.shape[0], "ratio": round(helped / base, 3),
                      "idle helper (dead or empty cell)": rounds2.max(axis=1).sum() / w.shape[0], "ratio2": round(rounds2.max(axis=1).sum() / base, 3)}), flush=True)
# upper bound: every lane with a list of two or more always has a helper
print(json.dumps({"every list halved": ((lw + 1) // 2).max(axis=1).sum() / w.shape[0], "ratio": round(((lw + 1) // 2).max(axis=1).sum() / base, 3)}))

# ---- tail mode: once a wavefront holds at most 64 / G live rays, every live ray is spread over G lanes and its inline list is tested
# G ids per round (rays only finish, so the switch is one-way); rounds of a step = max over the live rays of ceil(L / G)
n_live = live.sum(axis=1)                                        # [waves, CAP]
any_live = n_live > 0
for thresholds in ((16,), (32, 16), (32, 16, 8), (32,), (8,)):
    rounds = lw.max(axis=1)
    for t in sorted(thresholds, reverse=True):
        G = 64 // t
        rounds = np.where(n_live <= t, ((lw + G - 1) // G).max(axis=1), rounds)
    tail_steps = (any_live & (n_live <= max(thresholds))).sum()
    print(json.dumps({"tail mode at live rays <=": thresholds, "tri_rounds/wave": rounds.sum() / w.shape[0], "ratio": round(rounds.sum() / base, 3),
                      "cell steps in tail mode": round(float(tail_steps / any_live.sum()), 3),
                      "iterations/wave": (cell_iters + rounds.sum()) / w.shape[0]}), flush=True)
