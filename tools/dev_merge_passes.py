"""DEV TOOL: how many merge passes the scenes of tests/test_build_gpu.py::test_merge_iterations_in_place... run (a pass index beyond 12 means the fifth iteration:
the mask of merge.cu:361 drops to 0 and every cell is dirty again), and the cells / references entering every pass."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene
mem = api.MemManager(keep=True)
for n, seed, td, sd, alpha in [(30000, 91, 0.12, 2.4, 0.995), (60000, 5, 0.12, 2.4, 0.9999), (20000, 6, 0.5, 8.0, 0.9999), (4000, 7, 0.3, 1.0, 0.99999), (200000, 8, 0.12, 2.4, 0.999), (1000000, None, 0.12, 2.4, 0.995)]:
    tris = scene.make_soup(n, seed=seed) if seed is not None else scene.make_soup(n)
    d = mem.upload(tris); g = api.Grid()
    api.build_grid(mem, d, n, g, td, sd); api.merge_grid(mem, g, alpha)
    bc = mem.build_counts()
    print(json.dumps({"tris": n, "alpha": alpha, "merge_passes": bc["merge_passes"], "cells": bc["merge_cells"][:bc["merge_passes"]], "merged_cells": bc["merged_cells"]}), flush=True)
    g.free(); mem.free(d)
