#!/usr/bin/env python3
"""bench.py -- the headline benchmark: Mrays/s of primary-ray traversal (+ grid build ms) on the 1M-triangle
synthetic scene, BASELINE.json configs[1], on N MI355X of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one traverse_grid call over this rank's batch of rays (1024 x 1024 primary rays per GPU: weak
scaling -- rank r traverses sub-pixel sample r of N of the same camera, so every rank's batch is statistically
identical).  The grid is built on rank 0 with the gfx950 construction passes and broadcast once (RCCL); no
collective sits on the timed path.  Rank 0 prints ONE JSON line.

Everything measured runs through the C ABI (hagrid_amd/libhagrid_amd.so).  The CPU oracle is used only (a) as
the checker of the GPU hits and (b) as the timed `cpu_baseline` (rank 0, N = 1), never as the thing measured.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--rays", choices=["primary", "incoherent"], default="primary")
    ap.add_argument("--eye-dist", type=float, default=0.8, help="camera distance in scene diagonals (SURVEY proposed 1.5, where only ~17 %% of the pixels see the scene; DESIGN.md section 5)")
    ap.add_argument("--top-density", type=float, default=0.12)
    ap.add_argument("--snd-density", type=float, default=2.4)
    ap.add_argument("--alpha", type=float, default=0.995)
    ap.add_argument("--expansion", type=int, default=3)
    ap.add_argument("--compress", action="store_true")
    ap.add_argument("--build-iter", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    ap.add_argument("--image", type=int, default=2, choices=[0, 1, 2], help="traversal image built by setup_traversal: 0 off, 1 compact, 2 flat (default)")
    ap.add_argument("--bin-rays", type=int, default=None, help="ray binning before traversal (extension): default 1 for incoherent, 0 for primary rays")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo lets several ranks share one GPU in tests)")
    ap.add_argument("--device", type=int, default=None, help="GPU index (default: LOCAL_RANK)")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-rank code path (process group, broadcast, all-reduce) even with one rank")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from hagrid_amd import api, scene, dist as hdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback of the product path)")
    device = local_rank if args.device is None else args.device
    torch.cuda.set_device(device)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=args.backend)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    mem = api.MemManager(keep=True, device=device)
    info = mem.device_info()

    # ---- scene + grid: built on rank 0, broadcast once -------------------------------------------------------------
    n_tris = args.tris
    build_ms = None
    grid = None
    d_tris = 0
    t_bcast = 0.0
    if rank == 0:
        tris = scene.make_soup(n_tris)
        d_tris = mem.upload(tris)
        build = lambda g=None: api.build_all(mem, d_tris, n_tris, args.top_density, args.snd_density, args.alpha, args.expansion, args.compress, g)
        grid = build()                                  # warm-up build (also fills the buffer pool)
        times = []
        for _ in range(max(args.build_iter, 1)):        # main.cpp:494-508: free the grid, then time one full construction
            grid.free()
            times.append(api.profile(lambda: build(grid), mem))
        build_ms = float(np.mean(times))
        log(f"[bench] grid {grid.summary()} build_ms {build_ms:.2f} (min {min(times):.2f})")
    if multi:
        barrier(); t0 = time.perf_counter()
        grid, d_tris = hdist.broadcast_grid(mem, grid, d_tris, n_tris, src=0)
        barrier(); t_bcast = (time.perf_counter() - t0) * 1e3
    compressed = bool(grid.small_cells)

    # ---- this rank's ray batch ---------------------------------------------------------------------------------------
    n_rays = args.width * args.height
    if args.rays == "primary":
        rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, args.width, args.height, eye_dist=args.eye_dist, sample=rank, num_samples=world)
    else:
        rays = scene.make_rays_incoherent(grid.bbox_min, grid.bbox_max, n_rays, scene.RAY_SEED_BASE + 4, first=rank * n_rays)
    d_rays = mem.upload(rays)
    d_hits = mem.alloc(16 * n_rays)
    bin_rays = (1 if args.rays == "incoherent" else 0) if args.bin_rays is None else args.bin_rays
    mem.set_ray_binning(bin_rays)
    mem.set_option("traverse.image", args.image)
    api.setup_traversal(grid)                        # main.cpp:535; builds the traversal image (outside every timed region)
    setup_ms = api.profile(lambda: api.setup_traversal(grid), mem)

    # exact algorithmic byte counters of this batch (outside the timed region)
    stats = api.traverse_grid_stats(grid, d_tris, d_rays, d_hits, n_rays)
    ab = api.algorithmic_bytes(stats, compressed)

    # ---- timed region: W warm-up steps, then exactly K steps between barrier + synchronize ------------------------------
    for _ in range(args.warmup):
        api.traverse_grid(grid, d_tris, d_rays, d_hits, n_rays)
    barrier()
    t0 = time.perf_counter()
    api._check(mem, mem._L.hagrid_profile_begin(mem._ctx), "profile")         # HIP events on the launch stream
    for _ in range(args.steps):
        api.traverse_grid(grid, d_tris, d_rays, d_hits, n_rays)
    kernel_ms_total = mem._L.hagrid_profile_end(mem._ctx)                    # waits for the last launch
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0                                       # this rank's K steps; MAX over ranks below
    barrier()
    if multi:
        t = torch.tensor([elapsed, kernel_ms_total], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms_total = float(t[0]), float(t[1])
        s = torch.tensor([ab["B_ray"], ab["B_walk"], stats["hits"], stats["rays_hit_grid"]], dtype=torch.float64, device="cuda")
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        tot_bytes, tot_walk, tot_hits, tot_in = (float(v) for v in s)
    else:
        tot_bytes, tot_walk, tot_hits, tot_in = float(ab["B_ray"]), float(ab["B_walk"]), float(stats["hits"]), float(stats["rays_hit_grid"])

    hits = mem.download(d_hits, api.HIT_DTYPE, n_rays)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        total_rays = n_rays * world
        value = total_rays / (ms_per_step * 1e3)                   # Mrays/s, whole job
        kernel_ms = kernel_ms_total / args.steps                   # average launch duration (HIP events)
        achieved = ab["B_ray"] / (kernel_ms * 1e6)                 # GB/s of algorithmic bytes, rank-0 kernel
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath) and world == 1 and args.rays == "primary" and not args.compress and n_tris == 1_000_000:
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s (primary traversal) + grid build ms, 1M-tri scene @1/2/4/8 MI355X",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"soup-{n_tris} triangles, {args.rays} rays {args.width}x{args.height} per GPU"
                                   f" (BASELINE.json configs[1]), td {args.top_density} sd {args.snd_density} alpha {args.alpha} exp {args.expansion}"
                                   + (" compress" if args.compress else ""),
                       "rays_per_gpu": n_rays, "triangles": n_tris, "ray_binning": bin_rays, "traversal_image": {0: "off (construction format)", 1: "compact blocks", 2: "flat blocks: one 32-byte record per voxel, built by setup_traversal"}[args.image], "ray_packets": "8x8 pixel tiles, row length detected on the device at every call (buffer stays in image order)", "eye_dist_diagonals": args.eye_dist, "parallelism": f"ray-sharded x{world}, grid broadcast once",
                       "grid": grid.summary(), "device": info},
            "build_ms": None if build_ms is None else round(build_ms, 3),
            "grid_broadcast_ms": round(t_bcast, 3),
            "setup_traversal_ms": round(setup_ms, 3),
            "hit_fraction": round(tot_hits / total_rays, 4), "rays_entering_grid": round(tot_in / total_rays, 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "kernel": "traverse_kernel_img" if args.image else ("traverse_kernel_v3" if n_rays >= 24 * info["compute_units"] * 32 * 64 and not bin_rays else "traverse_kernel_v2"),
                         "kernel_ms": round(kernel_ms, 5),
                         "bytes_per_ray": round(ab["B_ray"] / n_rays, 1),
                         "walk_achieved": round(ab["B_walk"] / (kernel_ms * 1e6), 1),
                         "walk_frac": round(ab["B_walk"] / (kernel_ms * 1e6) / HBM_PEAK_GBPS, 4)},
        }
        # ---- CPU baseline + parity check: the oracle on the SAME grid, rank 0, N = 1 only ---------------------------------
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            d = grid.download()
            G = O.Grid.from_arrays(d["entries"], d["ref_ids"], d["cells"], d["small_cells"], d["bbox_min"], d["bbox_max"], d["dims"], d["shift"], d["offsets"])
            cores = os.cpu_count() or 1
            tris_h = scene.make_soup(n_tris)
            probe = min(65536, n_rays)
            t0 = time.perf_counter(); oh, _ = G.traverse(tris_h, rays[:probe], nthreads=cores); t_probe = time.perf_counter() - t0
            sample = int(min(n_rays, max(probe, probe * args.cpu_seconds / max(t_probe, 1e-6))))
            t0 = time.perf_counter(); oh, _ = G.traverse(tris_h, rays[:sample], nthreads=cores); t_cpu = time.perf_counter() - t0
            reps = 1
            while t_cpu < 0.5 * args.cpu_seconds and reps < 512:        # the whole batch is too small: repeat it
                t0 = time.perf_counter(); G.traverse(tris_h, rays[:sample], nthreads=cores); t_cpu += time.perf_counter() - t0; reps += 1
            same_id = bool((hits["id"][:sample] == oh["id"]).all())
            same_t = bool((hits["t"][:sample].view(np.uint32) == oh["t"].view(np.uint32)).all())
            out["cpu_baseline"] = {"value": round(sample * reps / t_cpu / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                                   "sample": f"first {sample} rays of the batch x{reps}, oracle traversal of the GPU-built grid, {cores} threads"}
            out["parity"] = {"rays_checked": sample, "ids_identical": same_id, "t_bit_identical": same_t}
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
