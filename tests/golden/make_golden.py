#!/usr/bin/env python3
"""Generates the committed golden vectors from the REFERENCE's own headers.

Run in the authoring container only (needs /root/reference and therefore oracle/_ref):

    python tests/golden/make_golden.py

Outputs (data only: inputs + the reference's outputs):
  tests/golden/l0_kat.npz        known-answer vectors for the L0 inline functions
                                 (common.h, grid.h, prims.h of the reference, compiled by
                                 oracle/Makefile into oracle/_ref/libhagrid_ref.so)
  tests/golden/config1_hits.npz  BASELINE config 1 (soup-10k, 64k incoherent rays): nearest hit
                                 (id, t) per ray by brute force with the reference's
                                 intersect_prim_ray -- grid-independent ground truth
  tests/golden/l0_kat_uvs.npz    intersect_prim_ray compiled with -DCOMPUTE_UVS (prims.h:285-288) on the inputs of
                                 l0_kat.npz: (ret, id, t, u, v)      [python tests/golden/make_golden.py uvs]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hagrid_amd import scene  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def main():
    R = O.ref_lib()
    if R is None:
        raise SystemExit("reference harness not available (needs /root/reference)")
    seed = 0x474F4C44454E
    u = lambda k, shape: scene.uniform01(seed + k, np.arange(int(np.prod(shape)), dtype=np.uint64)).reshape(shape)
    out = {}

    # ---- scalars --------------------------------------------------------------------------
    x = np.concatenate([np.float32([0.0, -0.0, 1.0, -1.0, 1e-30, -1e-30, 3e38, -3e38, 0.5, 1e-45]),
                        (u(1, (246,)) * 2 - 1).astype(np.float32) * np.float32(10)]).astype(np.float32)
    y = np.concatenate([np.float32([1.0, -1.0, -0.0, 0.0, -5.0, 5.0, 1.0, -1.0, -0.0, -1.0]),
                        (u(2, (246,)) * 2 - 1).astype(np.float32)]).astype(np.float32)
    out["rcp_in"] = x
    out["rcp_out"] = np.float32([R.ref_safe_rcp(float(v)) for v in x])
    out["prodsign_x"] = x; out["prodsign_y"] = y
    out["prodsign_out"] = np.float32([R.ref_prodsign(float(a), float(b)) for a, b in zip(x, y)])
    il = np.concatenate([np.arange(0, 70), 2 ** np.arange(1, 31), 2 ** np.arange(1, 31) - 1, 2 ** np.arange(1, 31) + 1]).astype(np.int32)
    out["ilog2_in"] = il
    out["ilog2_out"] = np.int32([R.ref_ilog2_i32(int(v)) for v in il])
    ld = (u(3, (64,)) * 4).astype(np.uint32); bg = (u(4, (64,)) * (2 ** 30)).astype(np.uint32)
    out["entry_log_dim"] = ld; out["entry_begin"] = bg
    out["entry_out"] = np.uint32([R.ref_make_entry(int(a), int(b)) for a, b in zip(ld, bg)])

    # ---- triangles: soup-4096 plus degenerate / axis-aligned cases ---------------------------
    tris = scene.make_soup(4096, seed=seed + 10)
    special = scene.tris_from_vertices(
        np.float32([[0, 0, 0], [0, 0, 0], [0.5, 0.5, 0.5], [0, 0, 0.25], [0.1, 0.1, 0.1]]),
        np.float32([[1, 0, 0], [0, 0, 0], [0.5, 0.5, 0.5], [1, 0, 0.25], [0.2, 0.1, 0.1]]),
        np.float32([[0, 1, 0], [0, 1, 0], [0.5, 0.5, 0.5], [0, 1, 0.25], [0.3, 0.1, 0.1]]))
    tris = np.concatenate([special, tris]).astype(np.float32)
    nt = tris.shape[0]
    out["tris"] = tris
    bb = np.zeros((nt, 8), dtype=np.float32)
    for i in range(nt):
        R.ref_tri_bbox(p(tris[i:i + 1]), p(bb[i:i + 1]))
    out["tri_bbox"] = bb

    # ---- intersect_prim_ray: rays aimed at a jittered point of the triangle -------------------
    n = 12288
    tid = (u(20, (n,)) * nt).astype(np.int64)
    t = tris[tid]
    b = u(21, (n, 2)); b1 = b[:, 0] * np.float32(1.4) - np.float32(0.2); b2 = b[:, 1] * np.float32(1.4) - np.float32(0.2)
    target = t[:, 0:3] - t[:, 4:7] * b1[:, None] + t[:, 8:11] * b2[:, None]
    org = (u(22, (n, 3)) * np.float32(1.2) - np.float32(0.1)).astype(np.float32)
    d = (target - org).astype(np.float32)
    sc = (u(23, (n,)) * np.float32(2.0) + np.float32(0.05)).astype(np.float32)
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3] = org; rays[:, 4:7] = d * sc[:, None]
    rays[:, 3] = np.where(u(24, (n,)) < 0.1, np.float32(0.7), np.float32(0.0))
    rays[:, 7] = np.where(u(25, (n,)) < 0.2, np.float32(0.9), scene.FLT_MAX)
    rays[0:64, 4] = 0.0   # exact zero direction components
    rays[64:128, 5] = -0.0
    hit = np.zeros(n, dtype=O.HIT_DTYPE); ret = np.zeros(n, dtype=np.int32)
    for i in range(n):
        h = np.array([(-1, rays[i, 7], 0, 0)], dtype=O.HIT_DTYPE)
        ret[i] = R.ref_intersect_prim_ray(p(tris[tid[i]:tid[i] + 1]), p(rays[i:i + 1]), int(tid[i]), p(h))
        hit[i] = h[0]
    out["ipr_tid"] = tid.astype(np.int32); out["ipr_rays"] = rays; out["ipr_ret"] = ret
    out["ipr_hit_id"] = hit["id"].copy(); out["ipr_hit_t"] = hit["t"].copy()

    # ---- intersect_prim_cell: boxes around / near the triangle --------------------------------
    n = 12288
    tid = (u(30, (n,)) * nt).astype(np.int64)
    tb = bb[tid]
    ctr = (np.float32(0.5) * (tb[:, 0:3] + tb[:, 4:7]) + (u(31, (n, 3)) - np.float32(0.5)) * np.float32(0.03)).astype(np.float32)
    hs = (u(32, (n, 3)) * np.float32(0.012) + np.float32(1e-4)).astype(np.float32)
    boxes = np.zeros((n, 8), dtype=np.float32)
    boxes[:, 0:3] = ctr - hs; boxes[:, 4:7] = ctr + hs
    ret = np.zeros(n, dtype=np.int32)
    for i in range(n):
        ret[i] = R.ref_intersect_prim_cell(p(tris[tid[i]:tid[i] + 1]), p(boxes[i:i + 1]))
    out["ipc_tid"] = tid.astype(np.int32); out["ipc_boxes"] = boxes; out["ipc_ret"] = ret

    # ---- compute_range ---------------------------------------------------------------------
    n = 4096
    dims = (u(40, (n, 3)) * 60 + 1).astype(np.int32)
    gbb = np.zeros((n, 8), dtype=np.float32)
    gbb[:, 0:3] = u(41, (n, 3)) - np.float32(0.5); gbb[:, 4:7] = gbb[:, 0:3] + u(42, (n, 3)) * np.float32(3) + np.float32(0.1)
    obb = np.zeros((n, 8), dtype=np.float32)
    a = gbb[:, 0:3] + (u(43, (n, 3)) * np.float32(1.4) - np.float32(0.2)) * (gbb[:, 4:7] - gbb[:, 0:3])
    obb[:, 0:3] = a; obb[:, 4:7] = a + u(44, (n, 3)) * np.float32(0.3)
    rng = np.zeros((n, 6), dtype=np.int32)
    for i in range(n):
        R.ref_compute_range(p(dims[i:i + 1]), p(gbb[i:i + 1]), p(obb[i:i + 1]), p(rng[i:i + 1]))
    out["range_dims"] = dims; out["range_grid_bb"] = gbb; out["range_obj_bb"] = obb; out["range_out"] = rng

    # ---- compute_grid_dims (libm cbrtf on the reference side) ----------------------------------
    n = 4096
    bbs = np.zeros((n, 8), dtype=np.float32)
    bbs[:, 4:7] = u(50, (n, 3)) * np.float32(2) + np.float32(0.01)
    nprims = (u(51, (n,)) ** 3 * 2000000).astype(np.int32)
    nprims[:16] = np.arange(16)
    dens = (u(52, (n,)) * np.float32(4) + np.float32(0.05)).astype(np.float32)
    gd = np.zeros((n, 3), dtype=np.int32)
    for i in range(n):
        R.ref_compute_grid_dims(p(bbs[i:i + 1]), int(nprims[i]), float(dens[i]), p(gd[i:i + 1]))
    out["gd_bb"] = bbs; out["gd_nprims"] = nprims; out["gd_density"] = dens; out["gd_out"] = gd

    # ---- lookup_entry on voxel maps (octree-shaped and flattened) of a small grid ------------------
    small = scene.make_soup(2000, seed=seed + 60)
    G = O.Grid.build(small, 0.12, 2.4)
    for tag in ("octree", "flat"):
        if tag == "flat":
            G.merge(0.995).flatten()
        ent = G.entries.copy()
        vd = np.int32(G.dims) << G.shift
        vox = (u(61 if tag == "octree" else 62, (8192, 3)) * vd[None, :]).astype(np.int32)
        res = np.zeros(8192, dtype=np.uint32)
        td = np.int32(G.dims)
        for i in range(8192):
            res[i] = R.ref_lookup_entry(p(ent), G.shift, p(td), p(vox[i:i + 1]))
        out[f"lk_{tag}_entries"] = ent; out[f"lk_{tag}_shift"] = np.int32(G.shift); out[f"lk_{tag}_dims"] = td
        out[f"lk_{tag}_voxels"] = vox; out[f"lk_{tag}_out"] = res

    # ---- foreach_ref -------------------------------------------------------------------------
    refs = np.int32([4, 7, 9, -1, 3, -1, 11, 12, 13, 14, -1])
    cells = np.zeros(4, dtype=O.CELL_DTYPE)
    cells["begin"] = [0, 4, 6, 2]; cells["end"] = [3, 5, 10, 2]
    scells = np.zeros(4, dtype=O.SMALL_CELL_DTYPE)
    scells["begin"] = [0, 4, 6, -1]
    fc = []; fs = []
    for i in range(4):
        vis = np.full(16, -7, dtype=np.int32)
        r = R.ref_foreach_ref_cell(p(cells[i:i + 1]), p(refs), p(vis)); fc.append(np.concatenate([[r], vis]))
        vis = np.full(16, -7, dtype=np.int32)
        r = R.ref_foreach_ref_small(p(scells[i:i + 1]), p(refs), p(vis)); fs.append(np.concatenate([[r], vis]))
    out["fe_refs"] = refs; out["fe_cell_begin"] = cells["begin"].copy(); out["fe_cell_end"] = cells["end"].copy()
    out["fe_small_begin"] = scells["begin"].copy()
    out["fe_cell_out"] = np.int32(fc); out["fe_small_out"] = np.int32(fs)

    np.savez_compressed(os.path.join(OUT, "l0_kat.npz"), **out)

    # ---- BASELINE config 1: brute-force hits with the reference arithmetic ----------------------
    tris = scene.make_soup(10000)
    lo, hi = scene.tris_bbox(tris)
    rays = scene.make_rays_incoherent(lo, hi, 65536, scene.RAY_SEED_BASE + 1)
    hits = O.brute_force(tris, rays, nthreads=8, use_ref=True)
    np.savez_compressed(os.path.join(OUT, "config1_hits.npz"), id=hits["id"].copy(), t=hits["t"].copy(),
                        tris_crc=np.uint32(np.bitwise_xor.reduce(tris.view(np.uint32).ravel())),
                        rays_crc=np.uint32(np.bitwise_xor.reduce(rays.view(np.uint32).ravel())))
    for f in ("l0_kat.npz", "config1_hits.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


def uvs():
    R = O.ref_lib_uvs()
    kat = np.load(os.path.join(OUT, "l0_kat.npz"))
    tris = np.ascontiguousarray(kat["tris"]); rays = np.ascontiguousarray(kat["ipr_rays"]); tid = kat["ipr_tid"]
    n = rays.shape[0]
    hit = np.zeros(n, dtype=O.HIT_DTYPE); ret = np.zeros(n, dtype=np.int32)
    for i in range(n):
        h = np.array([(-1, rays[i, 7], 0, 0)], dtype=O.HIT_DTYPE)
        ret[i] = R.ref_intersect_prim_ray(p(tris[tid[i]:tid[i] + 1]), p(rays[i:i + 1]), int(tid[i]), p(h))
        hit[i] = h[0]
    assert (ret == kat["ipr_ret"]).all() and (hit["t"].view(np.uint32) == kat["ipr_hit_t"].view(np.uint32)).all()
    np.savez_compressed(os.path.join(OUT, "l0_kat_uvs.npz"), ret=ret, id=hit["id"].copy(), t=hit["t"].copy(), u=hit["u"].copy(), v=hit["v"].copy())
    print("l0_kat_uvs.npz", os.path.getsize(os.path.join(OUT, "l0_kat_uvs.npz")), "bytes;", int(ret.sum()), "hits of", n)


if __name__ == "__main__":
    uvs() if sys.argv[1:] == ["uvs"] else main()
