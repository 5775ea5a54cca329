"""DEV TOOL: a renderer's frame loop with a MOVING camera (the reference's viewer, main.cpp:597-603: new rays into the same buffer every
frame).  Per frame the camera turns by `speed` x 0.005 rad and moves sideways by `speed` x 0.005 scene diagonals (speed 1 = one mouse pixel
and one key event of the reference's viewer per frame); the frame's rays are written into ONE device buffer and traversed; only the
traversal is timed (HIP events), the host synchronises once per frame as a viewer does.  Reported per speed: mean ms per frame with the tile
order at its defaults and in the default order (traverse.tile_order = 0), frames 9 .. N and the first eight; with --long L also the mean over L frames at the
viewer's speed (the give-up period of orders that do not last doubles: what a camera that keeps moving pays in the long run); then a buffer REFILLED
with another image every 8th frame.

Round 6: `--scene soup|clustered|gradient|shell|stadium`; policies "order" (the defaults: rays that keep changing get no order for a while and run in the default
order with its measured share) and "default_order" (traverse.tile_order = 0).  Every loop also returns a checksum of the last frame's hits: the policies must agree.
(A third policy -- the order sorted again behind every launch from the previous frame's costs -- was measured here and removed: tools/proto/README.md.)

usage: python tools/dev_moving_camera.py [--scene soup] [--width 1024] [--frames 48] [--speeds 0,0.1,0.25,0.5,1,2] [--long 400]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene

arg = lambda name, default: (sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default)
W = int(arg("--width", "1024")); frames = int(arg("--frames", "48")); long_frames = int(arg("--long", "0"))
speeds = [float(v) for v in arg("--speeds", "0,0.1,0.25,0.5,1,2").split(",")]
mem = api.MemManager(keep=True)
SCENE = arg("--scene", "soup")
tris = {"soup": lambda: scene.make_soup(1_000_000), "clustered": scene.make_clustered, "gradient": scene.make_gradient, "shell": scene.make_shell,
        "stadium": getattr(scene, "make_stadium", None)}[SCENE]()
d_tris = mem.upload(tris)
grid = api.build_all(mem, d_tris, tris.shape[0]); api.setup_traversal(grid)
n = W * W
d_rays = mem.alloc(32 * n); d_hits = mem.alloc(16 * n)


def frame_rays(f, speed):
    return scene.make_rays_primary(grid.bbox_min, grid.bbox_max, W, W, yaw=0.005 * speed * f, strafe=0.005 * speed * f)


def loop(speed, refill_every=0):
    """mean / first-frames / last-frames ms of `frames` frames; a fresh context state: the buffer is dropped from the hints by traversing another one"""
    ms = []
    for f in range(frames):
        rays = frame_rays(f, speed)
        if refill_every and (f // refill_every) % 2:
            rays = np.ascontiguousarray(rays.reshape(W, W, 8)[::-1].reshape(n, 8))          # another image: flipped top to bottom
        mem.copy_h2d(d_rays, rays)
        ms.append(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n), mem))
    global last_sum
    h = mem.download(d_hits, api.HIT_DTYPE, n)
    last_sum = int(h["id"].astype(np.int64).sum()) ^ int(h["t"].view(np.uint32).astype(np.int64).sum())
    return ms


POLICIES = [("order", {"traverse.tile_order": -1}), ("default_order", {"traverse.tile_order": 0})]
last_sum = 0


def settle():
    rays = frame_rays(0, 0.0); mem.copy_h2d(d_rays, rays)
    for _ in range(200): api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    mem.synchronize()


for speed in speeds:
    row = {"speed": speed, "turn_rad_per_frame": 0.005 * speed, "strafe_diag_per_frame": 0.005 * speed}
    sums = set()
    for label, opts in POLICIES:
        for k, v in opts.items(): mem.set_option(k, v)
        settle()
        ms = loop(speed)
        row[label] = {"mean_ms": round(float(np.mean(ms[8:])), 4), "first8": round(float(np.mean(ms[:8])), 4)}
        sums.add(last_sum)
    row["hits_agree"] = len(sums) == 1
    print(json.dumps(row), flush=True)
if long_frames:
    row = {"viewer speed, frames": long_frames}
    for label, opts in POLICIES:
        for k, v in opts.items(): mem.set_option(k, v)
        settle()
        frames, keep = long_frames, frames
        ms = loop(1.0)
        frames = keep
        row[label] = {"mean_ms": round(float(np.mean(ms)), 4), "frames 1-16": round(float(np.mean(ms[:16])), 4), "frames 17-80": round(float(np.mean(ms[16:80])), 4),
                      "after 80": round(float(np.mean(ms[80:])), 4) if len(ms) > 80 else None}
    print(json.dumps(row), flush=True)
for label, opts in POLICIES:
    for k, v in opts.items(): mem.set_option(k, v)
    settle()
    ms = loop(0.0, refill_every=8)
    print(json.dumps({"refilled with another image every 8th frame": label, "mean_ms": round(float(np.mean(ms[8:])), 4),
                      "per frame": [round(x, 4) for x in ms[8:32]]}), flush=True)
