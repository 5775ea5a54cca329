"""ANALYSIS (CPU, oracle traces): the walk of the general layout (trav_common.h GenWalk) simulated on the voxels a ray looks up -- per cell step: does the look-up stay
in the innermost block of the last one (one gather), or does it start at the top level again, and how many links does it then follow?  And what a second remembered
block (the parent of the innermost one) would save.   python tests/analysis/general_walk_model.py [clustered | soup:N:SD] [rays: primary | aimed]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hagrid_amd import scene
from oracle import oracle as O

what = sys.argv[1] if len(sys.argv) > 1 else "clustered"; kind = sys.argv[2] if len(sys.argv) > 2 else "primary"
if what == "clustered": tris = scene.make_clustered(); sd = 2.4
else: _, n, sd = what.split(":"); tris = scene.make_soup(int(n)); sd = float(sd)
G = O.Grid.full(tris, snd_density=sd)
print("grid", G.summary())
rays = scene.make_rays_primary(G.bbox_min, G.bbox_max, 256, 256) if kind == "primary" else scene.make_rays_aimed(G.bbox_min, G.bbox_max, 65536, 5)
n = rays.shape[0]; CAP = 512
L = O.lib(); L.orc_traverse_trace_voxels.argtypes = [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]; L.orc_traverse_trace_voxels.restype = None
lens = np.zeros((n, CAP), np.uint8); nc = np.zeros(n, np.int32); vox = np.zeros((n, CAP, 3), np.int16)
r = np.ascontiguousarray(rays, np.float32)
L.orc_traverse_trace_voxels(C.byref(G.g), tris.ctypes.data, r.ctypes.data, n, CAP, lens.ctypes.data, nc.ctypes.data, vox.ctypes.data)
entries = G.entries; shift = G.shift; dims = G.dims
steps = restarts = links = 0; parent_hits = 0; links_saved = 0
gathers = 0
hist = np.zeros(8, np.int64)
K0 = (0, 1, 2, 3); gathers_k0 = {k0: 0 for k0 in K0}        # a VIRTUAL top level k0 levels below the map's: the first gather of a restart resolves every link that ends at depth <= k0
for i in range(n):
    m = min(int(nc[i]), CAP)
    blk = None          # (first record, k, s) of the innermost block; parent likewise
    parent = None
    pv = None
    for sidx in range(m):
        v = vox[i, sidx].astype(np.int64)
        steps += 1
        g = 1
        if blk is not None and ((int(v[0]) ^ int(pv[0])) | (int(v[1]) ^ int(pv[1])) | (int(v[2]) ^ int(pv[2]))) >> (blk[2] + blk[1]) == 0:
            first, k, s = blk
            w = int(entries[first + ((int(v[0]) >> s) & ((1 << k) - 1)) + ((((int(v[1]) >> s) & ((1 << k) - 1)) + (((int(v[2]) >> s) & ((1 << k) - 1)) << k)) << k)])
            s_cur = s
        else:
            # would the parent of the innermost block still hold the voxel?
            if parent is not None and blk is not None and ((int(v[0]) ^ int(pv[0])) | (int(v[1]) ^ int(pv[1])) | (int(v[2]) ^ int(pv[2]))) >> (parent[2] + parent[1]) == 0:
                parent_hits += 1
            restarts += 1
            t = (int(v[0]) >> shift) + dims[0] * ((int(v[1]) >> shift) + dims[1] * (int(v[2]) >> shift))
            w = int(entries[t]); s_cur = shift; blk = None; parent = None
            ww, d, chain = w, 0, []
            while ww & 3:
                kk = ww & 3; d += kk; chain.append(d); ss = shift - d
                ww = int(entries[(ww >> 2) + ((int(v[0]) >> ss) & ((1 << kk) - 1)) + ((((int(v[1]) >> ss) & ((1 << kk) - 1)) + (((int(v[2]) >> ss) & ((1 << kk) - 1)) << kk)) << kk)])
            for k0 in K0: gathers_k0[k0] += 1 + sum(1 for dd in chain if dd > k0)
            restart_now = True
        while w & 3:
            k = w & 3; s_cur -= k; first = w >> 2
            parent = blk; blk = (first, k, s_cur)
            w = int(entries[first + ((int(v[0]) >> s_cur) & ((1 << k) - 1)) + ((((int(v[1]) >> s_cur) & ((1 << k) - 1)) + (((int(v[2]) >> s_cur) & ((1 << k) - 1)) << k)) << k)])
            links += 1; g += 1
        gathers += g; hist[min(g, 7)] += 1
        if not locals().get("restart_now", False):
            for k0 in K0: gathers_k0[k0] += g
        restart_now = False
        pv = v
print(f"{kind}: rays {n}, cell steps {steps} ({steps / n:.1f} per ray); look-ups that start at the top level {restarts / steps:.1%}; links followed {links / steps:.2f} per step; "
      f"gathers per step {gathers / steps:.2f}; restarts a remembered PARENT block would have served {parent_hits / max(restarts, 1):.1%}")
print("steps by number of dependent gathers (1 = inside the block):", {g: f"{hist[g] / steps:.1%}" for g in range(1, 8) if hist[g]})
top = int(dims[0]) * int(dims[1]) * int(dims[2])
print("gathers per step with a virtual top level k0 levels down (image + top * 8^k0 records of 16 bytes):",
      {k0: f"{gathers_k0[k0] / steps:.3f} (+{top * 8 ** k0 * 16 / 2 ** 20:.0f} MB)" for k0 in K0})
