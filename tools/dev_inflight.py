"""DEV TOOL: throughput of the 1024^2 primary batch with 1, 2, 3 or 4 independent traverse_grid calls in flight -- one context (= one
HIP stream) per call in flight, all reading the same grid, each with its own hit buffer.  A single launch spends its second half
draining (profiles/dev_r2_wave_timeline_tail.txt); a caller with independent batches (tiles of a frame, frames, samples) can
put the next launch on another stream and fill that drain.  Hits are compared with the single-stream run.
SHARE=0: a traversal image per context instead of hagrid_share_traversal; WIDTH, STEPS."""
import ctypes as C
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hagrid_amd import api, scene

W = int(os.environ.get("WIDTH", 1024))
STEPS = int(os.environ.get("STEPS", 200))
SHARE = int(os.environ.get("SHARE", 1))          # 0: every context builds its own copy of the traversal image
mem0 = api.MemManager(keep=True)
tris = scene.make_soup(1_000_000); d_tris = mem0.upload(tris)
grid0 = api.build_all(mem0, d_tris, tris.shape[0])
rays = scene.make_rays_primary(grid0.bbox_min, grid0.bbox_max, W, W, eye_dist=0.8); n = rays.shape[0]

def context(i):
    """(mem, grid, d_rays, d_hits): context i with its own stream, ray copy, hit buffer and traversal image over the shared grid."""
    if i == 0:
        mem, grid = mem0, grid0
    else:
        mem = api.MemManager(keep=True)
        if SHARE:
            grid = api.share_traversal(mem, grid0)          # one traversal image for all contexts
        else:
            grid = api.Grid(); grid.mem = mem
            C.memmove(C.byref(grid.pod), C.byref(grid0.pod), C.sizeof(grid0.pod))
    st = torch.cuda.Stream()
    mem.use_stream(st.cuda_stream)
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    if i == 0 or not SHARE:
        api.setup_traversal(grid)
    return mem, grid, d_rays, d_hits, st

ctxs = [context(i) for i in range(4)]
want = None
for k in (1, 2, 3, 4, 1):
    use = ctxs[:k]
    for _ in range(40):
        for mem, grid, d_rays, d_hits, _s in use:
            api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(STEPS):
        mem, grid, d_rays, d_hits, _s = use[s % k]
        api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = True
    for mem, grid, d_rays, d_hits, _s in use:
        h = mem.download(d_hits, api.HIT_DTYPE, n)
        if want is None: want = h.copy()
        same = same and bool((h["id"] == want["id"]).all() and (h["t"].view(np.uint32) == want["t"].view(np.uint32)).all())
    print(json.dumps({"in flight": k, "steps": STEPS, "ms per batch": round(dt * 1e3 / STEPS, 4), "Grays/s": round(n * STEPS / dt / 1e9, 2),
                      "host issue ms per call": round(t_issue * 1e3 / STEPS, 4), "hits identical": same, "shared image": bool(SHARE)}), flush=True)
