"""ANALYSIS (CPU, oracle traces): the cost of an 8 x 8 pixel tile as the cells of its longest ray, per scene -- how skewed is the distribution the tile order sorts, and how many tiles
cost a multiple of the median tile (the share "traverse.quad_head" starts with four lanes per ray; profiles/NOTES.md "Round 5").   python tests/analysis/tile_cost_model.py"""
import ctypes as C, sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hagrid_amd import scene
from oracle import oracle as O
def costs(tris, W, H, **params):
    G = O.Grid.full(tris, **params)
    rays = scene.make_rays_primary(G.bbox_min, G.bbox_max, W, H)
    n = rays.shape[0]; CAP = 4
    L = O.lib(); L.orc_traverse_trace_voxels.argtypes = [C.c_void_p]*3 + [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]; L.orc_traverse_trace_voxels.restype = None
    lens = np.zeros((n, CAP), np.uint8); nc = np.zeros(n, np.int32); vox = np.zeros((n, CAP, 3), np.int16)
    r = np.ascontiguousarray(rays, np.float32)
    L.orc_traverse_trace_voxels(C.byref(G.g), tris.ctypes.data, r.ctypes.data, n, CAP, lens.ctypes.data, nc.ctypes.data, vox.ctypes.data)
    c = nc.reshape(H//8, 8, W//8, 8).max(axis=(1,3)).ravel()
    return c
for name, tris, params in (("soup", scene.make_soup(1_000_000), {}), ("clustered", scene.make_clustered(), {}), ("config3 grid", scene.make_soup(1_000_000), dict(top_density=0.15, snd_density=3.0))):
    for W,H in ((512,512),(1024,1024)):
        c = costs(tris, W, H, **params)
        m = np.median(c); mean = c.mean()
        print(name, W, "tiles", c.size, "median", m, "mean %.1f" % mean, "max", c.max(), "p90", np.percentile(c,90), "p99", np.percentile(c,99),
              "frac>=1.5med %.3f" % (c>=1.5*m).mean(), "frac>=2med %.3f" % (c>=2*m).mean(), "frac>=3med %.3f" % (c>=3*m).mean(), "frac>=max/2 %.3f" % (c>=c.max()/2).mean(),
              "frac>=2mean %.3f" % (c>=2*mean).mean(), "sum/max/8192 %.3f" % (c.sum()/c.max()/8192))
