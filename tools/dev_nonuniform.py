"""DEV TOOL: the very non-uniform scene (scene.make_clustered: six dense blobs in a sparse soup, grid shift 6) -- traversal image
(2: the traversal image -- the general layout of slim records on this grid; 0: the construction format) on different batches.  Prints JSON lines; hits are compared across formats.

    python tools/dev_nonuniform.py frames            whole 1024^2 frame, the image rows that see blobs / do not, 1M incoherent rays
    python tools/dev_nonuniform.py bands             the frame in 8 bands of 128 rows and growing prefixes
    python tools/dev_nonuniform.py tiles             the six heaviest 8x8 pixel tiles alone (one wavefront each), the frame without the 64 heaviest
    python tools/dev_nonuniform.py dense             rays aimed at the blobs; a camera close to one blob
    python tools/dev_nonuniform.py compressed        the same scene built with compress (SmallCells)
    python tools/dev_nonuniform.py tile TY TX        one 8x8 tile, a few launches per format and nothing else (the command PMC runs profile)
"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene

mode = sys.argv[1] if len(sys.argv) > 1 else "frames"
mem = api.MemManager(keep=True)
tris = scene.make_clustered(); N = tris.shape[0]
d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, N, compress=(mode == "compressed"))
print(json.dumps({"mode": mode, "triangles": N, "grid": grid.summary()}), flush=True)


def steps_of(rays):
    n = rays.shape[0]
    d_r = mem.upload(rays); d_h = mem.alloc(16 * n); d_s = mem.alloc(4 * n)
    api.traverse_grid_stats(grid, d_tris, d_r, d_h, n, d_s)
    s = mem.download(d_s, np.int32, n); h = mem.download(d_h, api.HIT_DTYPE, n)
    mem.free(d_r); mem.free(d_h); mem.free(d_s)
    return s, h


def run(label, rays, images=(2, 0, 2, 0), extra=None, repeats=9):
    rays = np.ascontiguousarray(rays.reshape(-1, 8)); n = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    res = {"rays": label, "n": n}; res.update(extra or {}); ref = None
    for img in images:
        mem.set_option("traverse.image", img); api.setup_traversal(grid)
        if img: res[f"image{img}_MB"] = round(mem.image_bytes(grid) / 1e6, 1)
        for _ in range(2): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        t = sorted(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)) for _ in range(repeats))
        res.setdefault(f"image{img}_ms", []).append(round(t[len(t) // 2], 4))
        h = mem.download(d_hits, api.HIT_DTYPE, n)
        if ref is None: ref = h
        else: assert (h["id"] == ref["id"]).all() and (h["t"].view(np.uint32) == ref["t"].view(np.uint32)).all()
    print(json.dumps(res), flush=True)
    mem.free(d_rays); mem.free(d_hits)


prim = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 1024)
incoh = lambda seed: scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, 1 << 20, seed)
if mode in ("frames", "compressed"):
    _, hp = steps_of(prim)
    in_blob = hp["id"] >= 100000
    print(json.dumps({"primary rays hitting a blob": int(in_blob.sum()), "hitting the sparse soup": int(((hp["id"] >= 0) & ~in_blob).sum()), "missing": int((hp["id"] < 0).sum())}), flush=True)
    run("primary 1024^2", prim); run("incoherent 1M", incoh(9))
    if mode == "frames":
        rows = np.flatnonzero(in_blob.reshape(1024, 1024).any(axis=1))
        frame = prim.reshape(1024, 1024, 8)
        run("image rows that see blobs", frame[rows.min(): rows.min() + (len(rows) // 8) * 8]); run("image rows above them", frame[: max(8, (rows.min() // 8) * 8)])
elif mode == "bands":
    frame = prim.reshape(1024, 1024, 8)
    for b in range(8): run(f"rows {128 * b}..{128 * b + 127}", frame[128 * b: 128 * b + 128], images=(2, 0, 2, 0))
    for k in (256, 512, 768, 1024): run(f"rows 0..{k - 1}", frame[:k], images=(2, 0, 2, 0))
elif mode in ("tiles", "tile"):
    P = prim.reshape(128, 8, 128, 8, 8)
    if mode == "tile":
        ty, tx = int(sys.argv[2]), int(sys.argv[3])
        run(f"tile ({ty},{tx})", P[ty, :, tx, :], images=(2, 0), repeats=5)
    else:
        s, _ = steps_of(prim); s = s.reshape(128, 8, 128, 8)
        tile_max = s.max(axis=(1, 3)); order = np.argsort(-tile_max.reshape(-1))
        for k in range(6):
            ty, tx = divmod(int(order[k]), 128)
            run(f"tile ({ty},{tx})", P[ty, :, tx, :], extra={"steps_max": int(tile_max[ty, tx]), "steps_mean": round(float(s[ty, :, tx, :].mean()), 1)})
        keep = np.ones(128 * 128, bool); keep[order[:64]] = False
        run("frame without the 64 heaviest tiles (tile order)", P.transpose(0, 2, 1, 3, 4).reshape(128 * 128, 64, 8)[keep])
elif mode == "dense":
    n = 1 << 20
    aimed = incoh(12).copy()
    k = np.arange(n) % 6
    centre = np.stack([0.17 + 0.14 * k, 0.32 + 0.08 * k, 0.22 + 0.1 * k], axis=1).astype(np.float32)
    aimed[:, 4:7] = centre - aimed[:, 0:3] + np.float32(0.02) * aimed[:, 4:7]
    yy, xx = np.meshgrid(np.arange(1024, dtype=np.float32), np.arange(1024, dtype=np.float32), indexing="ij")
    close = np.zeros((n, 8), np.float32)
    close[:, 0:3] = np.float32([0.17, 0.32, 0.22 - 0.05])
    close[:, 4] = xx.reshape(-1) / np.float32(1024.0) - np.float32(0.5); close[:, 5] = yy.reshape(-1) / np.float32(1024.0) - np.float32(0.5); close[:, 6] = 1.0
    close[:, 7] = np.float32(10.0)
    for label, rays in (("aimed at the blobs, incoherent origins", aimed), ("camera close to one blob", close)):
        s, h = steps_of(rays)
        run(label, rays, extra={"steps_mean": round(float(s.mean()), 1), "hit a blob": round(float((h["id"] >= 100000).mean()), 3)})
else:
    raise SystemExit(__doc__)
