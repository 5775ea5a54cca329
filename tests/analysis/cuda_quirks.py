"""DEV TOOL (CPU, oracle): what the two documented deviations from a literal CUDA run change -- cells, references,
cells visited and triangles tested per ray, for the 1M-triangle scene and the headline batch (N, W from the environment)."""
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from hagrid_amd import scene
N = int(os.environ.get("N", 1000000)); W = int(os.environ.get("W", 512))
tris = scene.make_soup(N)
ref = None
for mask, name in ((0, "documented intent (product, default oracle)"), (1, "D1: reversed rear partition (CUB)"), (2, "D2: stale expand buffer"), (3, "D1 + D2: as-CUDA")):
    t0 = time.time()
    with O.cuda_quirks(mask):
        G = O.Grid.full(tris)
    rays = scene.make_rays_primary(G.bbox_min, G.bbox_max, W, W)
    h, st = G.traverse(tris, rays, nthreads=8)
    if ref is None: ref = h
    same = bool((h["id"] == ref["id"]).all() and (h["t"].view(np.uint32) == ref["t"].view(np.uint32)).all())
    print(json.dumps({"mode": name, "cells": G.num_cells, "refs": G.num_refs, "entries": G.num_entries, "cells/ray": round(st["cells"] / st["rays"], 3),
                      "tests/ray": round(st["refs"] / st["rays"], 3), "hits identical to mode 0": same, "s": round(time.time() - t0, 1)}), flush=True)
