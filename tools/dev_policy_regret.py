"""DEV TOOL: the POLICY-REGRET table.  Scene families x image sizes x ray kinds: the default policy of hagrid_traverse_grid against every forced
setting of the options its rules choose between -- tile order, the share of tiles that start with four lanes per ray (tail / head), the share trial, mailbox,
padded triangles, two ids per round trip, band rows.  Per cell: steady-state ms (back-to-back launches, event-timed) of the default and of the best forced setting, and the
regret (default / best - 1).  A rule whose default loses more than 3 % somewhere is a candidate for a measured trial or for removal; an option that never wins by more than
3 % anywhere is a candidate for deletion.  Hits are checked (checksum) to be the same under every setting.

usage: python tools/dev_policy_regret.py [--scenes soup,clustered,gradient,shell,stadium] [--sizes 640x480,1280x720,1024x1024,1920x1080,4096x4096]
                                         [--kinds primary,bounce,incoherent,aimed] [--out profiles/...txt]"""
import json, os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import api, scene

arg = lambda name, default: (sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default)
scenes = arg("--scenes", "soup,clustered,gradient,shell,stadium").split(",")
sizes = [tuple(int(v) for v in s.split("x")) for s in arg("--sizes", "640x480,1280x720,1024x1024,1920x1080,4096x4096").split(",")]
kinds = arg("--kinds", "primary,bounce,incoherent,aimed").split(",")
FORCED = [("traverse.tile_order", (0, 1)), ("traverse.quad_tail", (0, 25, 50, 100)), ("traverse.quad_head", (0,)), ("traverse.share_trial", (0,)),
          ("traverse.mailbox", (0, 1)), ("traverse.tail_dual", (0, 1)), ("traverse.band_rows", (1, 4))]
DEFAULTS = {"traverse.tile_order": -1, "traverse.quad_tail": -1, "traverse.quad_head": 20, "traverse.share_trial": 1, "traverse.mailbox": -1,
            "traverse.tail_dual": -1, "traverse.band_rows": 0}
mem = api.MemManager(keep=True)
rows = []


def measure(go, n, d_hits):
    """steady state: settle (synchronised now and then: the trials of the policy take their samples over launches), then the median of 8 bursts.  The context starts
    from nothing (what its trials measured under the previous setting must not decide under this one)."""
    mem.forget_hints()
    t0 = time.time()
    while time.time() - t0 < 0.2:
        for _ in range(10): go()
        mem.synchronize()
    k = 20 if n <= (1 << 21) else 6
    ms = sorted(api.profile(lambda: [go() for _ in range(k)], mem) / k for _ in range(8))
    h = mem.download(d_hits, api.HIT_DTYPE, min(n, 1 << 20))
    return ms[3], zlib.crc32(h.tobytes())


for sc in scenes:
    tris = scene.make_soup(1_000_000) if sc == "soup" else getattr(scene, "make_" + sc)()
    d_tris = mem.upload(tris)
    grid = api.build_all(mem, d_tris, tris.shape[0]); api.setup_traversal(grid)
    lo, hi = grid.bbox_min, grid.bbox_max
    batches = []
    for kind in kinds:
        if kind == "primary":
            batches += [(f"primary {w}x{h}", scene.make_rays_primary(lo, hi, w, h), 0) for (w, h) in sizes]
        elif kind == "bounce":
            for (w, h) in ((1024, 1024), (2048, 2048)):
                prim = scene.make_rays_primary(lo, hi, w, h); n = prim.shape[0]
                d_r = mem.upload(prim); d_h = mem.alloc(16 * n); api.traverse_grid(grid, d_tris, d_r, d_h, n)
                ph = mem.download(d_h, api.HIT_DTYPE, n); mem.free(d_r); mem.free(d_h)
                batches.append((f"bounce {w}x{h}", scene.make_rays_bounce(tris, prim, ph, lo, hi, scene.RAY_SEED_BASE + 5), 0))
        elif kind == "incoherent":
            batches += [(f"incoherent {m}M binned", scene.make_rays_incoherent(lo, hi, m << 20, scene.RAY_SEED_BASE + 4), 1) for m in (1, 4)]
        elif kind == "aimed" and sc == "clustered":
            batches.append(("aimed 1M binned", scene.make_rays_aimed(lo, hi, 1 << 20, 5), 1))
    for name, rays, binning in batches:
        n = rays.shape[0]
        d_rays = mem.upload(np.ascontiguousarray(rays, np.float32)); d_hits = mem.alloc(16 * n)
        mem.set_ray_binning(binning)
        go = lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
        for k, v in DEFAULTS.items(): mem.set_option(k, v)
        base, crc = measure(go, n, d_hits)
        cell = {"scene": sc, "batch": name, "default_ms": round(base, 5), "forced": {}}
        for key, values in FORCED:
            for v in values:
                for k2, v2 in DEFAULTS.items(): mem.set_option(k2, v2)
                mem.set_option(key, v)
                ms, c2 = measure(go, n, d_hits)
                cell["forced"][f"{key}={v}"] = round(ms, 5)
                if c2 != crc: cell.setdefault("HITS_DIFFER", []).append(f"{key}={v}")
        for k, v in DEFAULTS.items(): mem.set_option(k, v)
        again, _ = measure(go, n, d_hits)                         # the default once more: the noise floor of the cell
        cell["default_again_ms"] = round(again, 5)
        best_key = min(cell["forced"], key=cell["forced"].get)
        cell["best"] = best_key; cell["best_ms"] = cell["forced"][best_key]
        cell["regret_pct"] = round(100.0 * (min(base, again) / cell["best_ms"] - 1.0), 1)
        rows.append(cell)
        print(json.dumps(cell), flush=True)
        mem.set_ray_binning(0)
        mem.free(d_rays); mem.free(d_hits)
    grid.free(); mem.free(d_tris)

print("\n| scene | batch | default ms | best forced setting | its ms | regret % |")
print("|---|---|---|---|---|---|")
for c in rows:
    print(f"| {c['scene']} | {c['batch']} | {min(c['default_ms'], c['default_again_ms']):.4f} | {c['best']} | {c['best_ms']:.4f} | {c['regret_pct']:+.1f} |")
wins = {}
for c in rows:
    for k, v in c["forced"].items():
        gain = 100.0 * (min(c["default_ms"], c["default_again_ms"]) / v - 1.0)
        wins[k] = max(wins.get(k, -1e9), gain)
print("\nlargest gain of every forced setting over the default, anywhere (per cent; <= 3: the setting never wins):")
for k in sorted(wins, key=wins.get, reverse=True): print(f"  {k:28s} {wins[k]:+6.1f}")
