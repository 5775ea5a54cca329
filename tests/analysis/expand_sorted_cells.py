"""DEV TOOL (GPU + oracle): what the divergence of the face walks costs hagrid_expand_grid.  The cells of the merged + flattened 1M-triangle grid are
renumbered so that cells with faces of similar area sit together (entries remapped; the grid is the same grid), and the expansion of the original and
of the renumbered arrays is timed.  The renumbered run is an upper bound on what sorting / binning the cells inside the expansion passes could give."""
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from hagrid_amd import api, scene

tris = scene.make_soup(int(os.environ.get("N", 1000000)))
G = O.Grid.build(tris); G.merge(); G.flatten()
cells = np.array(G.cells, copy=True); entries = np.array(G.entries, copy=True); refs = np.array(G.ref_ids, copy=True)
c = cells.view(np.int32).reshape(-1, 8)
ext = c[:, 4:7] - c[:, 0:3]
area = np.stack([ext[:, 1] * ext[:, 2], ext[:, 2] * ext[:, 0], ext[:, 0] * ext[:, 1]], axis=1).max(axis=1)
mem = api.MemManager(keep=True)
d_tris = mem.upload(tris)

def run(cells_a, entries_a, label):
    times = []
    for rep in range(4):
        grid = api.Grid.upload(mem, entries_a, refs, cells_a, None, G.bbox_min, G.bbox_max, G.dims, G.shift, G.offsets)
        ms = api.profile(lambda: api.expand_grid(mem, grid, d_tris, 3), mem)
        times.append(ms); out_cells = grid.num_cells
        grid.free()
    print(json.dumps({"cells": label, "expand ms (4 runs)": [round(t, 3) for t in times]}), flush=True)

run(cells, entries, "construction order")
def morton(v):
    v = v.astype(np.uint64); out = np.zeros(v.shape[0], np.uint64)
    for b in range(10):
        for a in range(3): out |= ((v[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return out
top = (c[:, 0:3] >> G.shift)
top_major = (top[:, 2].astype(np.int64) * G.dims[1] + top[:, 1]) * G.dims[0] + top[:, 0]
for label, order in (("Morton order of the cells' lower corners", np.argsort(morton(c[:, 0:3]), kind="stable")),
                     ("top-level cell major (x fastest), construction order inside", np.argsort(top_major, kind="stable")),
                     ("by largest face area, ascending", np.argsort(area, kind="stable")), ("by largest face area, descending", np.argsort(-area, kind="stable")),
                     ("random", np.random.default_rng(1).permutation(cells.shape[0]))):
    new_id = np.empty(cells.shape[0], np.int64); new_id[order] = np.arange(cells.shape[0])
    e = entries.astype(np.uint32).copy()
    leaf = (e & 3) == 0
    e[leaf] = (new_id[e[leaf] >> 2].astype(np.uint32) << 2)
    run(np.ascontiguousarray(cells[order]), e, label)
