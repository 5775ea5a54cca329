"""DEV TOOL (CPU, oracle): per-wavefront instruction model of the traversal kernels on the 1M-primary batch.
Traces the list length of every cell each ray visits, groups rays into wavefronts (strips of 64 or 8x8 tiles) and counts
lock-step iterations: cell steps = max cells over the wave, triangle iterations = sum over steps of the max list length
among live lanes; alternatives: pairs packed across lanes, consolidation of thin waves."""
import os, sys, json, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from hagrid_amd import scene

N = int(os.environ.get("N", 1000000)); W = int(os.environ.get("W", 1024)); SUB = int(os.environ.get("SUB", 4))
tris = scene.make_soup(N)
t0 = time.time(); G = O.Grid.full(tris); print("oracle build", round(time.time() - t0, 1), "s", G.summary(), flush=True)
rays = scene.make_rays_primary(G.bbox_min, G.bbox_max, W, W)
# sample: every SUB-th 8-row band of the image (keeps whole tiles and whole strips)
rows = np.arange(W).reshape(-1, 8)[::SUB].reshape(-1)
idx = (rows[:, None] * W + np.arange(W)[None, :]).reshape(-1)
r = np.ascontiguousarray(rays[idx]); n = r.shape[0]
CAP = 320
lens = np.zeros((n, CAP), np.uint8); nc = np.zeros(n, np.int32)
L = O.lib(); L.orc_traverse_trace.argtypes = [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]; L.orc_traverse_trace.restype = None
IDCAP = 512
ids = np.full((n, IDCAP), -1, np.int32); nids = np.zeros(n, np.int32)
t0 = time.time()
L.orc_traverse_trace(C.byref(G.g), tris.ctypes.data, r.ctypes.data, n, CAP, lens.ctypes.data, nc.ctypes.data, IDCAP, ids.ctypes.data, nids.ctypes.data)
print("trace", round(time.time() - t0, 1), "s; rays", n, "cells/ray", nc.mean(), "refs/ray", lens.sum() / n, flush=True)
live = np.arange(CAP)[None, :] < nc[:, None]
# repeated tests: a reference already tested among the last k tests of the same ray
valid = np.arange(IDCAP)[None, :] < np.minimum(nids, IDCAP)[:, None]
rep = {}
for kwin in (1, 2, 4, 8, 16):
    seen = np.zeros_like(valid)
    for d in range(1, kwin + 1):
        seen[:, d:] |= (ids[:, d:] == ids[:, :-d]) & valid[:, d:]
    rep[kwin] = float(seen.sum() / valid.sum())
first = np.zeros_like(valid)
srt = np.sort(np.where(valid, ids, -1), axis=1)
distinct = ((srt[:, 1:] != srt[:, :-1]) & (srt[:, 1:] >= 0)).sum() + (srt[:, 0] >= 0).sum()
print(json.dumps({"tests": int(valid.sum()), "distinct (ray, triangle) pairs": int(distinct), "repeat fraction within last k tests": rep}), flush=True)
bands = n // (8 * W)
def waves(order):
    return order.reshape(-1, 64)
strip = np.arange(n)
tile = np.arange(n).reshape(bands, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
C_CELL, C_TRI = 95, 55            # VALU instructions per cell step / per triangle test (ISA of v2)
for name, order in (("strips 64x1", strip), ("tiles 8x8", tile)):
    w = waves(order)
    lw, live_w = lens[w], live[w]                     # [waves, 64, CAP]
    cell_iters = live_w.any(axis=1).sum(axis=1)       # per wave
    tri_iters = lw.max(axis=1).astype(np.int64).sum(axis=1)
    pairs = lw.astype(np.int64).sum(axis=1)           # [waves, CAP] pairs per step
    passes = ((pairs + 63) // 64).sum(axis=1)
    fold = tri_iters
    lanes_cell = live_w.sum(axis=1).sum(axis=1)
    now = C_CELL * cell_iters + C_TRI * tri_iters
    packed = C_CELL * cell_iters + 25 * (pairs > 0).sum(axis=1) + 54 * passes + 8 * fold
    out = {"order": name, "waves": int(w.shape[0]), "cell_iters/wave": float(cell_iters.mean()), "tri_iters/wave": float(tri_iters.mean()),
           "lane util cell": float(lanes_cell.sum() / (64 * cell_iters.sum())), "lane util tri": float(lens[w].sum() / (64 * tri_iters.sum())),
           "VALU/wave now": float(now.mean()), "VALU/wave packed pairs": float(packed.mean()),
           "max list per step (mean over live steps)": float(lw.max(axis=1)[live_w.any(axis=1)].mean())}
    # consolidation model: G waves share their rays after step S (perfect repacking inside a group of G waves)
    for Gn in (4, 16):
        g = w[: (w.shape[0] // Gn) * Gn].reshape(-1, Gn * 64)
        lg, live_g = lens[g], live[g]
        alive = live_g.sum(axis=1)                    # [groups, CAP] live rays per step
        waves_needed = (alive + 63) // 64
        mx = lg.max(axis=1).astype(np.int64)
        out[f"VALU/wave consolidated x{Gn}"] = float(((C_CELL + C_TRI * mx) * waves_needed).sum() / (g.shape[0] * Gn))
        pg = lg.astype(np.int64).sum(axis=1)
        out[f"VALU/wave consolidated x{Gn} + packed"] = float((C_CELL * waves_needed + 25 * (pg > 0) + 54 * ((pg + 63) // 64) + 8 * mx * waves_needed).sum() / (g.shape[0] * Gn))
    print(json.dumps(out), flush=True)

# periodic consolidation: the waves of a group repack their live rays (order kept) at given steps
def periodic(order, Gn, repack_steps, overhead=70):
    g = order[: (order.shape[0] // (Gn * 64)) * Gn * 64].reshape(-1, Gn * 64)
    lg, live_g = lens[g], live[g]                     # [groups, Gn*64, CAP]
    total = np.zeros(g.shape[0], np.int64)
    slot = np.tile(np.arange(Gn * 64), (g.shape[0], 1))          # current lane slot of every ray
    bounds = [0] + [s for s in repack_steps if s < CAP] + [CAP]
    for a, b in zip(bounds[:-1], bounds[1:]):
        if a > 0:                                     # repack at step a: live rays compacted in order
            alive = live_g[:, :, a]
            slot = np.where(alive, np.cumsum(alive, axis=1) - 1, Gn * 64 - 1)
            total += overhead * ((alive.sum(axis=1) + 63) // 64 + 0)        # surviving waves pay the repack (exiting ones pay ~half)
        wave_of = slot // 64
        for wv in range(Gn):
            m = (wave_of == wv)[:, :, None] & live_g[:, :, a:b]
            mx = np.where(m, lg[:, :, a:b], 0).max(axis=1).astype(np.int64)
            total += (C_CELL * m.any(axis=1) + C_TRI * mx).sum(axis=1)
    return float(total.sum() / (g.shape[0] * Gn))

for Gn in (4, 8, 16):
    for steps in ((8, 16, 24, 32, 48, 64, 96, 128), (4, 8, 12, 16, 20, 24, 32, 40, 48, 64, 96, 128), (6, 12, 18, 24, 30, 36, 48, 64, 96, 128), tuple(range(2, 200, 2))):
        print(json.dumps({"G": Gn, "repack_at": steps[:6], "VALU/wave": periodic(tile, Gn, steps)}), flush=True)

# k rays per lane, processed one after the other (static assignment: lane l of wave w gets the l-th ray of k consecutive tiles)
def sequential(order, k):
    w = order[: (order.shape[0] // (64 * k)) * 64 * k].reshape(-1, k, 64)          # [waves, k, 64]
    total = np.zeros(w.shape[0]); cells_it = np.zeros(w.shape[0]); tri_it = np.zeros(w.shape[0])
    for wi in range(0, w.shape[0], 256):
        ww = w[wi:wi + 256]
        B = ww.shape[0]
        seq = np.zeros((B, 64, k * CAP), np.uint8); alive = np.zeros((B, 64, k * CAP), bool)
        pos = np.zeros((B, 64), np.int64)
        for j in range(k):
            r = ww[:, j, :]                                    # [B, 64] ray ids
            ln = nc[r]                                         # cells of that ray
            for s in range(CAP):
                m = s < ln
                if not m.any(): break
                bi, li = np.nonzero(m)
                seq[bi, li, pos[bi, li] + s] = lens[r[bi, li], s]
                alive[bi, li, pos[bi, li] + s] = True
            pos += ln
        ci = alive.any(axis=1).sum(axis=1); ti = seq.max(axis=1).astype(np.int64).sum(axis=1)
        cells_it[wi:wi + B] = ci; tri_it[wi:wi + B] = ti
    return float((C_CELL * cells_it + C_TRI * tri_it).mean() / k), float(cells_it.mean() / k), float(tri_it.mean() / k)

for k in (1, 2, 4):
    v, ci, ti = sequential(tile, k)
    print(json.dumps({"rays per lane (sequential)": k, "VALU per 64 rays": v, "cell iters per 64 rays": ci, "tri iters per 64 rays": ti}), flush=True)
