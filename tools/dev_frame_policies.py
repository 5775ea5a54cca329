"""DEV TOOL: the frame loop of tools/dev_moving_camera.py (new rays into one buffer every frame, host synchronises per frame, traversal timed with HIP events) under a
list of option sets -- which policy serves rays that change from launch to launch?  Prints mean ms of frames 9.. per option set, the hits' checksum and what the
context remembers about the buffer afterwards (hagrid_kat_order_state).
usage: python tools/dev_frame_policies.py [--scene clustered] [--width 1024] [--frames 40] [--speed 1.0] [--sets "name:key=v,key=v;name2:..."]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene

arg = lambda name, default: (sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default)
W = int(arg("--width", "1024")); H = int(arg("--height", str(W))); frames = int(arg("--frames", "40")); speed = float(arg("--speed", "1.0")); SCENE = arg("--scene", "clustered")
DEFAULT_SETS = ("policy:;no_share_trial:traverse.share_trial=0;default_order:traverse.tile_order=0;"
                "default_order_no_trial:traverse.tile_order=0,traverse.share_trial=0;default_order_quad50:traverse.tile_order=0,traverse.quad_tail=50")
sets = []
for item in arg("--sets", DEFAULT_SETS).split(";"):
    name, _, kv = item.partition(":")
    sets.append((name, {k: int(v) for k, v in (p.split("=") for p in kv.split(",") if p)}))
RESET = {"traverse.tile_order": -1, "traverse.quad_head": 20, "traverse.quad_tail": -1, "traverse.share_trial": 1}
mem = api.MemManager(keep=True)
tris = {"soup": lambda: scene.make_soup(1_000_000), "clustered": scene.make_clustered, "gradient": scene.make_gradient, "shell": scene.make_shell,
        "stadium": getattr(scene, "make_stadium", None)}[SCENE]()
d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, tris.shape[0]); api.setup_traversal(grid)
n = W * H
d_rays = mem.alloc(32 * n); d_hits = mem.alloc(16 * n)
frame_rays = [scene.make_rays_primary(grid.bbox_min, grid.bbox_max, W, H, yaw=0.005 * speed * f, strafe=0.005 * speed * f) for f in range(frames)]
print(json.dumps({"scene": SCENE, "width": W, "height": H, "frames": frames, "speed": speed, "grid": grid.summary()}), flush=True)
for name, opts in sets:
    for k, v in {**RESET, **opts}.items(): mem.set_option(k, v)
    mem.copy_h2d(d_rays, frame_rays[0])
    for _ in range(200): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    mem.synchronize()
    ms = []
    for f in range(frames):
        mem.copy_h2d(d_rays, frame_rays[f])
        ms.append(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n), mem))
    h = mem.download(d_hits, api.HIT_DTYPE, n)
    cs = int(h["id"].astype(np.int64).sum()) ^ int(h["t"].view(np.uint32).astype(np.int64).sum())
    print(json.dumps({"set": name, "opts": opts, "mean_ms": round(float(np.mean(ms[8:])), 4), "first8": round(float(np.mean(ms[:8])), 4), "min": round(float(min(ms[8:])), 4),
                      "hits": cs, "state": mem.order_state(d_rays)}), flush=True)
