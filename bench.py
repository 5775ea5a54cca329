#!/usr/bin/env python3
"""bench.py -- Mrays/s of ray traversal (+ grid build ms) on the BASELINE scenes, on N MI355X of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3                       # the headline: BASELINE.json configs[1]
    python bench.py --config 3|4|5 [--scaling weak|strong]               # the other GPU configurations of BASELINE.md section 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--config C]

--config uses BASELINE.md's numbering (config C = BASELINE.json configs[C - 1]):
  2  soup-1M, defaults, 1024 x 1024 primary rays PER GPU (weak scaling: rank r traces sub-pixel sample r of N)
  3  soup-1M, --top-density 0.15 --snd-density 3.0 --expansion 3, 4096 x 4096 primary rays
  4  soup-1M, defaults, 128M incoherent rays (random origin + direction), ray binning on
  5  soup-8M, defaults + --compress, 64M diffuse-bounce rays leaving the hit points of an 8192 x 8192 primary image
Configs 3-5 shard ONE batch over the ranks (strong scaling, contiguous ranges: hagrid_amd.scene.shard_range) unless
--scaling weak is given; config 2 is weak unless --scaling strong is given.

One "step" = one traverse_grid call over this rank's rays, resident in HBM.  The grid is built on rank 0 with the gfx950
construction passes and broadcast once (RCCL); no collective sits on the timed path.  Rank 0 prints ONE JSON line.

Everything measured runs through the C ABI (hagrid_amd/libhagrid_amd.so).  The CPU oracle is used only (a) as the checker
of the GPU hits and (b) as the timed `cpu_baseline` (rank 0, N = 1), never as the thing measured.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
# What else can bind a gather kernel, measured on this part (profiles/micro_r2l_vector_memory.txt: tools/micro/l1_gather.hip, every lane its own random
# 16-byte record; tools/micro/valu_lanes.hip): requests per second the memory side serves by where the working set lives, and the SIMD cycles a
# wavefront VALU instruction costs whatever its live lanes.
L2_GATHER_GPS = 256.0       # G requests/s, working set 1 MB (L2-resident)
MALL_GATHER_GPS = 80.0      # G requests/s, working set 16 MB ... 256 MB (Infinity Cache)
HBM_GATHER_GPS = 55.0       # G requests/s, working set 1 GB (52 - 57 measured): the chip's rate of random 64-byte fetches from HBM
VALU_CYCLES_PER_INST = 2.2
TA_CYCLES_PER_LANE_ACCESS = 1.25    # CU-cycles per live lane of a 16-byte gather that hits the vector L1 (80 per 64-lane instruction)
TA_EXTRA_CYCLES_PER_L1_MISS = 1.17  # ... and on top when it goes to L2 (155 per instruction)
NUM_CUS = 256
SHADER_CLOCK_HZ = 2.4e9
NUM_SIMDS = 1024


def binding_resources(counters: dict, kernel_ms: float, working_set_bytes: int, traffic_bytes):
    """Fractions of the resources a traversal launch can be bound by, from the counters of tools/gpu_traffic_config.sh (per launch) and the
    kernel time of THIS run.  Every fraction is achieved / what the part delivers for that access pattern: <= 1 up to measurement noise."""
    t = kernel_ms * 1e-3
    out = {}
    if traffic_bytes is not None:
        out["hbm_bytes"] = {"achieved": round(traffic_bytes / t / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(traffic_bytes / t / 1e9 / HBM_PEAK_GBPS, 4),
                            "what": "fabric-side bytes of the L2s (FETCH_SIZE raw + WRITE_SIZE; Infinity-Cache hits included) against the HBM peak"}
    if "TCC_MISS_sum" in counters:
        beyond = working_set_bytes > (256 << 20)
        peak = HBM_GATHER_GPS if beyond else MALL_GATHER_GPS
        out["fabric_fetch_rate"] = {"achieved": round(counters["TCC_MISS_sum"] / t / 1e9, 2), "peak": peak, "unit": "G requests/s", "frac": round(counters["TCC_MISS_sum"] / t / 1e9 / peak, 4),
                                    "what": "L2 misses per second against the measured rate of random fetches " + ("from HBM (working set beyond the 256 MB Infinity Cache)" if beyond else "from the Infinity Cache (working set within its 256 MB)")}
    if "TCP_TCC_READ_REQ_sum" in counters:
        out["l2_request_rate"] = {"achieved": round(counters["TCP_TCC_READ_REQ_sum"] / t / 1e9, 2), "peak": L2_GATHER_GPS, "unit": "G requests/s",
                                  "frac": round(counters["TCP_TCC_READ_REQ_sum"] / t / 1e9 / L2_GATHER_GPS, 4), "what": "vector-L1 misses per second against the measured rate of L2-resident random gathers"}
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in counters and "TCP_TCC_READ_REQ_sum" in counters:
        cyc = counters["TCP_TOTAL_CACHE_ACCESSES_sum"] * TA_CYCLES_PER_LANE_ACCESS + counters["TCP_TCC_READ_REQ_sum"] * TA_EXTRA_CYCLES_PER_L1_MISS
        out["vector_memory_path"] = {"achieved": round(counters["TCP_TOTAL_CACHE_ACCESSES_sum"] / t / 1e9, 2), "unit": "G lane accesses/s",
                                     "peak": round(NUM_CUS * SHADER_CLOCK_HZ / TA_CYCLES_PER_LANE_ACCESS / 1e9, 1), "frac": round(cyc / (NUM_CUS * SHADER_CLOCK_HZ * t), 4),
                                     "what": f"lane accesses of the vector L1s x {TA_CYCLES_PER_LANE_ACCESS} CU-cycles + their misses x {TA_EXTRA_CYCLES_PER_L1_MISS} more (16-byte gathers, every lane its own line: "
                                             "tools/micro/ta_lanes.hip, 80 / 155 cycles per 64-lane instruction from L1 / L2) against 256 CUs x 2.4 GHz: the address / tag path of the CUs"}
    if "SQ_INSTS_VALU" in counters:
        cyc = counters["SQ_INSTS_VALU"] * VALU_CYCLES_PER_INST / NUM_SIMDS
        out["valu_issue"] = {"achieved": round(counters["SQ_INSTS_VALU"] / t / 1e9, 2), "peak": round(NUM_SIMDS * SHADER_CLOCK_HZ / VALU_CYCLES_PER_INST / 1e9, 1), "unit": "G wavefront-instructions/s",
                             "frac": round(cyc / (SHADER_CLOCK_HZ * t), 4), "what": f"VALU wavefront-instructions x {VALU_CYCLES_PER_INST} SIMD-cycles (whatever the live lanes) against 1024 SIMDs x 2.4 GHz",
                             "lanes_enabled": None if not counters.get("SQ_THREAD_CYCLES_VALU") else round(counters["SQ_THREAD_CYCLES_VALU"] / counters["SQ_INSTS_VALU"] / 64, 3)}
    if counters.get("SQ_WAVE_CYCLES"):
        out["wave_time"] = {"waiting_for_memory": round(counters.get("SQ_WAIT_ANY", 0) / counters["SQ_WAVE_CYCLES"], 3), "waiting_to_issue": round(counters.get("SQ_WAIT_INST_ANY", 0) / counters["SQ_WAVE_CYCLES"], 3),
                            "issuing": round(counters.get("SQ_ACTIVE_INST_ANY", 0) / counters["SQ_WAVE_CYCLES"], 3), "what": "where the resident wavefronts spend their time (not a throughput fraction)"}
    ranked = sorted(((v["frac"], k) for k, v in out.items() if "frac" in v), reverse=True)
    return out, (ranked[0][1] if ranked else None), (ranked[0][0] if ranked else None)


def binding_limiter(resources: dict, top, top_frac):
    """`top` names the MOST USED throughput resource.  It is the limiter only when it is close to 1: a launch whose wavefronts spend most of their
    time waiting for dependent gathers while no throughput resource is saturated is bound by the latency of its chains at the occupancy it has
    (Little's law).  Checked by intervention on the per-GPU share of configuration 5 (profiles/NOTES.md "Round 4", gpurun_out/r4w2, r4w3): triangles
    padded to 64 bytes took 15 % of the fabric fetches away (85.9M -> 72.5M per launch) and 1.2 % of the kernel time.  Returns (limiter, waiting)."""
    waiting = (resources.get("wave_time") or {}).get("waiting_for_memory", 0.0)
    return ("memory_latency" if (waiting >= 0.6 and top_frac is not None and top_frac < 0.95) else top), waiting

def roofline_headline(algorithmic_gbps: float, traffic_bytes, kernel_ms: float):
    """`achieved` / `frac` of the roofline block say ONE thing: the HBM rate -- bytes that crossed the fabric side of the L2s per launch (counter passes on these very
    kernel sources) over the kernel time, against the HBM peak.  Without counters for these sources the algorithmic bytes stand in, and `frac_kind` says so: that figure
    is not bounded by 1 for a cache-resident working set (configuration 3: 1.3)."""
    if traffic_bytes is not None:
        a = traffic_bytes / (kernel_ms * 1e6)
        return round(a, 1), round(a / HBM_PEAK_GBPS, 4), "hbm_measured: fabric-side bytes of the L2s per launch (rocprofv3 --pmc passes, traffic_source) / kernel time / HBM peak"
    return round(algorithmic_gbps, 1), round(algorithmic_gbps / HBM_PEAK_GBPS, 4), "algorithmic: bytes the kernel gathers (frac_image) -- no counter file for these kernel sources and this batch, see traffic_source"


CONFIGS = {
    2: dict(baseline="1M-triangle synthetic scene, default densities, 1M primary rays on 1xMI355X", tris=1_000_000, rays="primary",
            width=1024, height=1024, scaling="weak", params={}),
    3: dict(baseline="1M-triangle scene, --top-density 0.15 --snd-density 3.0 --expansion 3, 16M rays, 1xMI355X (build-time + traversal bench)",
            tris=1_000_000, rays="primary", width=4096, height=4096, scaling="strong", params=dict(top_density=0.15, snd_density=3.0, expansion=3)),
    4: dict(baseline="1M-triangle scene, 128M incoherent (random origin+dir) rays sharded across 8xMI355X via RCCL grid broadcast",
            tris=1_000_000, rays="incoherent", total=1 << 27, scaling="strong", params={}, bin_rays=1),
    5: dict(baseline="8M-triangle scene with --compress voxel map, 64M diffuse-bounce rays, 8xMI355X",
            tris=8_000_000, rays="bounce", width=8192, height=8192, scaling="strong", params=dict(compress=True)),
    # not a BASELINE.json configuration: the non-uniform scene irregular grids exist for (main.cpp:246-275 loads such scenes) -- six dense blobs in a
    # sparse soup, grid shift 6 -- `--config clustered`; `--rays aimed`: 1M incoherent rays aimed at the blobs
    6: dict(baseline="(extension, not in BASELINE.json) clustered 1M-triangle scene (scene.make_clustered: six dense blobs in a sparse soup, six-level voxel map), 1M primary rays",
            tris=1_000_000, scene="clustered", rays="primary", width=1024, height=1024, scaling="weak", params={}),
    # a MESH-shaped scene through the front-end's own triangle packing: scene.make_stadium -- tori and spheres with shared vertices and a grain of
    # dust inside a hall of ten huge triangles, edges over four orders of magnitude ("teapot in a stadium", the reference README's motivation) -- `--config stadium`
    7: dict(baseline="(extension, not in BASELINE.json) stadium: a 0.95M-triangle indexed mesh (scene.make_stadium: finely tessellated tori / spheres / a grain of dust inside a hall "
                     "of ten huge triangles, edges 1.4 ... 1.3e-4; six-level voxel map, lists of up to hundreds of references), 1M primary rays",
            tris=948_786, scene="stadium", rays="primary", width=1024, height=1024, scaling="weak", params={}),
}
CONFIG_NAMES = {"clustered": 6, "stadium": 7}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def upload_generated(mem, d_rays, gen, first, count, chunk=1 << 20, threads=None, keep_host=0):
    """Generates rays [first, first + count) in chunks on a thread pool (the generators are counter-based: any slice is
    reproducible) and copies each chunk to its place in the device buffer.  Returns the first `keep_host` rays."""
    threads = threads or min(32, os.cpu_count() or 1)
    starts = list(range(0, count, chunk))
    kept = []
    with ThreadPoolExecutor(max_workers=threads) as ex:
        for off, arr in zip(starts, ex.map(lambda o: gen(first + o, min(chunk, count - o)), starts)):
            mem.copy_h2d(d_rays + 32 * off, arr)
            if off < keep_host:
                kept.append(arr[:keep_host - off])
    return np.concatenate(kept) if kept else np.zeros((0, 8), np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=lambda v: CONFIG_NAMES.get(v) or int(v), default=2, choices=sorted(CONFIGS),
                    help="BASELINE.md configuration (config C = BASELINE.json configs[C - 1]); extensions: `clustered` (= 6), the non-uniform soup; `stadium` (= 7), a mesh-shaped scene")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None, help="default: weak for config 2, strong (one batch sharded over the ranks) for 3-5")
    ap.add_argument("--tris", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--rays", choices=["primary", "incoherent", "bounce", "aimed"], default=None, help="aimed: incoherent origins, directions towards the blobs of the clustered scene")
    ap.add_argument("--total-rays", type=int, default=None, help="size of the incoherent batch (config 4: 2^27)")
    ap.add_argument("--eye-dist", type=float, default=0.8, help="camera distance in scene diagonals (SURVEY proposed 1.5, where no ray reaches the grid within tmax = clip; DESIGN.md section 5)")
    ap.add_argument("--top-density", type=float, default=None)
    ap.add_argument("--snd-density", type=float, default=None)
    ap.add_argument("--alpha", type=float, default=0.995)
    ap.add_argument("--expansion", type=int, default=None)
    ap.add_argument("--compress", action="store_true")
    ap.add_argument("--build-iter", type=int, default=10)
    ap.add_argument("--settle-ms", type=float, default=100.0, help="untimed burst of the same step before the warm-up steps, so that the timed region runs at settled clocks (0: none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-order-compare", action="store_true", help="skip the extra K steps in the default tile order after the timed region (`tile_order` in the line); "
                    "the counter passes of tools/gpu_round.sh use it so that their per-launch averages are the steady state's")
    ap.add_argument("--inflight", type=int, default=2, help="after the timed region (never part of `value`): the same K steps with this many independent calls in flight, "
                    "one context = one stream each over ONE traversal image (hagrid_share_traversal); reported as `pipelined`; 0 or 1: skip")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target CPU time of the all-cores baseline sample")
    ap.add_argument("--image", type=int, default=2, choices=[0, 1, 2], help="traversal image built by setup_traversal: 0 off (construction format), 1 / 2 on (default)")
    ap.add_argument("--hits-hash", action="store_true", help="experiments: `hits_sha256` of this rank's whole hit buffer after the timed steps in the line")
    ap.add_argument("--opts", default="", help="experiments: comma-separated key=value pairs for hagrid_set_option, e.g. traverse.tile_order=0")
    ap.add_argument("--bin-rays", type=int, default=None, help="ray binning before traversal (extension): default 1 for incoherent, 0 otherwise")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo lets several ranks share one GPU in tests)")
    ap.add_argument("--device", type=int, default=None, help="GPU index (default: LOCAL_RANK)")
    ap.add_argument("--shard", default=None, metavar="R/W", help="one GPU, strong scaling only: trace the contiguous range rank R of W ranks would get (the per-GPU share of an 8-GPU run, "
                    "e.g. --config 4 --shard 3/8 = 16M of the 128M rays); `value` is then this GPU's rate on that share, the line says so in config.shard")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-rank code path (process group, broadcast, all-reduce) even with one rank")
    args = ap.parse_args()

    cfg = CONFIGS[args.config]
    n_tris = args.tris or cfg["tris"]
    ray_kind = args.rays or cfg["rays"]
    width = args.width or cfg.get("width", 1024)
    height = args.height or cfg.get("height", 1024)
    scaling = args.scaling or cfg["scaling"]
    top_density = args.top_density if args.top_density is not None else cfg["params"].get("top_density", 0.12)
    snd_density = args.snd_density if args.snd_density is not None else cfg["params"].get("snd_density", 2.4)
    expansion = args.expansion if args.expansion is not None else cfg["params"].get("expansion", 3)
    compress = args.compress or cfg["params"].get("compress", False)

    import torch
    import torch.distributed as dist
    from hagrid_amd import api, scene, dist as hdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and "RANK" not in os.environ and args.gpus > 1:
            # started bare (`python bench.py --gpus N`, the shape of the driver's N = 1 command): become the launch line DESIGN.md section 7
            # names -- one rank per GPU under torch.distributed.run -- instead of giving up; rank 0's JSON line is this process's stdout.
            port = os.environ.get("MASTER_PORT") or str(29600 + os.getpid() % 2000)
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
                   "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
            log(f"[bench] --gpus {args.gpus} without a launcher: re-executing under torch.distributed.run (port {port})")
            sys.stdout.flush(); sys.stderr.flush()
            os.execv(sys.executable, cmd)
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
        if int(os.environ.get("LOCAL_WORLD_SIZE", world)) == world:
            # one node: RCCL's bootstrap sockets need no interface but the loopback (whatever else the box has need not route to itself);
            # the data path between the GPUs is xGMI / shared memory either way.  A caller's own NCCL_SOCKET_IFNAME stands.
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
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
    build_ms = build_ms_min = None
    build_block = None
    grid = None
    d_tris = 0
    t_bcast = 0.0
    tris_host = None
    clustered = cfg.get("scene") == "clustered"; stadium = cfg.get("scene") == "stadium"
    if rank == 0 or ray_kind == "bounce":
        tris_host = scene.make_clustered() if clustered else (scene.make_stadium() if stadium else scene.make_soup(n_tris))    # bounce rays need the hit triangles' normals on every rank
        n_tris = tris_host.shape[0]
    if rank == 0:
        d_tris = mem.upload(tris_host)
        build = lambda g=None: api.build_all(mem, d_tris, n_tris, top_density, snd_density, args.alpha, expansion, compress, g)
        grid = build()                                  # warm-up builds (the first also fills the buffer pool; the clocks settle
        grid.free(); build(grid)                        # over the first few launches of a process)
        times = []
        for _ in range(max(args.build_iter, 1)):        # main.cpp:494-508: free the grid, then time one full construction
            grid.free()
            times.append(api.profile(lambda: build(grid), mem))
        build_ms = float(np.mean(times)); build_ms_min = float(min(times))
        bc = mem.build_counts()
        bb = api.build_algorithmic_bytes(bc)
        build_block = {"bytes": bb, "ms": round(build_ms, 3), "achieved": round(bb["total"] / (build_ms * 1e6), 1), "unit": "GB/s",
                       "frac": round(bb["total"] / (build_ms * 1e6) / HBM_PEAK_GBPS, 4),
                       "formula": "SURVEY.md 8(d) 'algorithmic bytes -- build' on the sizes the passes recorded (hagrid_get_build_counts)",
                       "counts": {k: bc[k] for k in ("top_refs", "level_refs", "level_cells", "build_cells", "build_refs", "build_entries", "merge_passes", "merged_cells", "merged_refs", "flatten_entries_out")}}
        log(f"[bench] grid {grid.summary()} build_ms {build_ms:.2f} (min {min(times):.2f})")
    if multi:
        barrier(); t0 = time.perf_counter()
        grid, d_tris = hdist.broadcast_grid(mem, grid, d_tris, n_tris, src=0)
        barrier(); t_bcast = (time.perf_counter() - t0) * 1e3
    compressed = bool(grid.small_cells)

    # ---- this rank's rays ------------------------------------------------------------------------------------------------
    # weak: every rank traces a batch of the configuration's full size (primary: sub-pixel sample `rank` of `world` of the same
    # camera; incoherent / bounce: its own stretch of the sequence).  strong: the ONE batch is cut into contiguous ranges.
    if ray_kind in ("incoherent", "aimed"):
        total = args.total_rays or cfg.get("total", width * height)
    else:
        total = width * height
    shard = None
    if args.shard:
        if world != 1 or scaling != "strong":
            raise SystemExit("--shard R/W is the one-GPU view of a strong-scaling run: needs --gpus 1 and strong scaling")
        shard = tuple(int(v) for v in args.shard.split("/"))
        if not (len(shard) == 2 and 0 <= shard[0] < shard[1]):
            raise SystemExit("--shard R/W: 0 <= R < W")
    if scaling == "strong":
        first, end = scene.shard_range(total, *(shard or (rank, world)))
        n_rays = end - first
        sample, nsamples = 0, 1
    else:
        first, n_rays = (0, total) if ray_kind not in ("incoherent", "aimed") else (rank * total, total)
        sample, nsamples = rank, world
    d_rays = mem.alloc(32 * n_rays)
    d_hits = mem.alloc(16 * n_rays)
    keep = min(n_rays, 1 << 20)                         # host copy of the first rays: CPU baseline + parity sample
    t_gen = time.perf_counter()
    if ray_kind in ("incoherent", "aimed"):
        make = scene.make_rays_aimed if ray_kind == "aimed" else scene.make_rays_incoherent
        gen = lambda f, c: make(grid.bbox_min, grid.bbox_max, c, scene.RAY_SEED_BASE + 4, first=f)
        rays_head = upload_generated(mem, d_rays, gen, first, n_rays, keep_host=keep)
    else:
        gen = lambda f, c: scene.make_rays_primary(grid.bbox_min, grid.bbox_max, width, height, first=f, count=c, eye_dist=args.eye_dist, sample=sample, num_samples=nsamples)
        rays_head = upload_generated(mem, d_rays, gen, first, n_rays, chunk=1 << 22, keep_host=keep)
    bin_rays = (cfg.get("bin_rays", 1 if ray_kind in ("incoherent", "aimed") else 0)) if args.bin_rays is None else args.bin_rays
    mem.set_option("traverse.image", args.image)
    for kv in filter(None, args.opts.split(",")):                 # experiments: any hagrid_set_option key
        k, v = kv.split("="); mem.set_option(k, int(v))
    api.setup_traversal(grid)                        # main.cpp:535; builds the traversal image (outside every timed region)
    setup_ms = api.profile(lambda: api.setup_traversal(grid), mem)
    if ray_kind == "bounce":
        # BASELINE config 5: diffuse-bounce rays leaving the primary hit points, in the image order of the primary rays
        api.traverse_grid(grid, d_tris, d_rays, d_hits, n_rays)
        chunk = 1 << 22
        kept = []
        for off in range(0, n_rays, chunk):
            c = min(chunk, n_rays - off)
            prim = mem.download(d_rays + 32 * off, np.float32, 8 * c).reshape(c, 8)
            ph = mem.download(d_hits + 16 * off, api.HIT_DTYPE, c)
            b = scene.make_rays_bounce(tris_host, prim, ph, grid.bbox_min, grid.bbox_max, scene.RAY_SEED_BASE + 5, first=first + off)
            mem.copy_h2d(d_rays + 32 * off, b)
            if off < keep:
                kept.append(b[:keep - off])
        rays_head = np.concatenate(kept)
    log(f"[bench] rank {rank}: {n_rays} {ray_kind} rays [{first}, {first + n_rays}) generated in {time.perf_counter() - t_gen:.1f} s")
    mem.set_ray_binning(bin_rays)

    # exact algorithmic byte counters of this batch (outside the timed region)
    stats = api.traverse_grid_stats(grid, d_tris, d_rays, d_hits, n_rays)
    record_bytes = mem.image_record_bytes(grid) if args.image else 32
    ab = api.algorithmic_bytes(stats, compressed, record_bytes)

    # ---- timed region: W warm-up steps, then exactly K steps between barrier + synchronize ------------------------------
    # the statistics pass and the host work above leave the GPU idle for a while; a short burst of the same step lets the clocks
    # settle before the W warm-up steps (steady state is what a renderer sees: 200 timed steps run 3 % faster than 20 without it)
    t_settle = time.perf_counter()
    while args.settle_ms > 0 and (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        for _ in range(8):
            api.traverse_grid(grid, d_tris, d_rays, d_hits, n_rays)
        torch.cuda.synchronize()
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
    sums = [ab["B_ray"], ab["B_walk"], stats["hits"], stats["rays_hit_grid"], n_rays, ab["B_image"]]
    if multi:
        t = torch.tensor([elapsed, kernel_ms_total], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms_total = float(t[0]), float(t[1])
        s = torch.tensor(sums, dtype=torch.float64, device="cuda")
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        sums = [float(v) for v in s]
    tot_bytes, tot_walk, tot_hits, tot_in, total_rays, tot_image = sums

    n_head = rays_head.shape[0]
    hits = mem.download(d_hits, api.HIT_DTYPE, n_head)
    hits_sha = None
    if args.hits_hash:      # (experiments: two libraries or two code paths over the same rays must leave the same bytes)
        import hashlib
        hits_sha = hashlib.sha256(mem.download(d_hits, api.HIT_DTYPE, n_rays).tobytes()).hexdigest()[:16]

    # ---- what the learned tile order is worth (outside the timed region) --------------------------------------------------------
    # Launches over a ray buffer the context has seen before dispatch their 8x8 tiles longest first, by the costs the previous launches
    # left (traverse.hip "tile order"); the W warm-up steps are where that is learned, the K timed steps are the steady state -- what the
    # reference's own benchmark loop measures (main.cpp:398-447: the same rays, iteration after iteration).  The same K steps in the default
    # order (= what the FIRST launch over a new buffer costs) are reported next to it.
    tile_order = None
    if not bin_rays and not args.no_order_compare:
        try:
            mem.set_option("traverse.tile_order", 0)
            for _ in range(max(args.warmup, 1)):
                api.traverse_grid(grid, d_tris, d_rays, d_hits, n_rays)
            # (settled like the timed region: launches in this order for --settle-ms first, synchronised now and then as a renderer's frames are -- the share of tiles that
            # start with four lanes per ray measures itself over the first launches in this order, and the host only learns a sample's time once it looks again)
            t_settle = time.perf_counter()
            while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
                for _ in range(10): api.traverse_grid(grid, d_tris, d_rays, d_hits, n_rays)
                mem.synchronize()
            ms0 = api.profile(lambda: [api.traverse_grid(grid, d_tris, d_rays, d_hits, n_rays) for _ in range(args.steps)], mem) / args.steps
            hits0 = mem.download(d_hits, api.HIT_DTYPE, n_head)
            tile_order = {"ms_per_step_default_order": round(ms0, 5),
                          # the fractions of the roofline block on the default-order time (what a first launch over a new buffer reaches)
                          "walk_frac_default_order": round(ab["B_walk"] / (ms0 * 1e6) / HBM_PEAK_GBPS, 4),
                          "frac_image_default_order": round(ab["B_image"] / (ms0 * 1e6) / HBM_PEAK_GBPS, 4), "hits_identical": bool((hits0["id"] == hits["id"]).all() and
                          (hits0["t"].view(np.uint32) == hits["t"].view(np.uint32)).all()),
                          "how": "`value` is the steady state of a renderer's loop: tiles dispatched longest first, by the costs the previous launches over the same ray "
                                 "buffer left (learned in the warm-up steps, refreshed every 32nd launch); ms_per_step_default_order = the "
                                 "same K steps with traverse.tile_order = 0: launches in the default order, as the first launches over a new buffer and every launch of a camera that moves run "
                                 "(in that order the share of tiles that start with four lanes per ray measures itself over the first launches: the rule's share against a half; "
                                 "the very first launch runs with the rule's).  Hits do not depend on the order"}
            # A frame loop with a MOVING camera (the reference's viewer, main.cpp:597-603: new rays into the same buffer every frame): per frame the
            # view turns by 0.005 rad and the eye moves sideways by 0.005 scene diagonals -- one mouse pixel and one key event of that viewer --
            # and a quarter of that; the host synchronises per frame.  The order is followed only while the buffer's rays are near the ones it was
            # learned on (checked on the device); a buffer refilled with another image (flipped) every 8th frame shows what a stale order costs now.
            if ray_kind == "primary" and n_rays == width * height and n_rays <= (1 << 22) and world == 1:
                def frames_ms(speed, order, refill=0, frames=32):
                    mem.set_option("traverse.tile_order", order)
                    # (a loop starts from a settled context: the frozen frame traversed often enough for an order to be learned and for the pause the
                    # context takes from learning -- 64 launches and twice as many the next time, after orders that did not last in the loop before -- to be over:
                    # 64 + a first order + the 32 launches after which an order that lasted resets that period fit into 200)
                    mem.copy_h2d(d_rays, scene.make_rays_primary(grid.bbox_min, grid.bbox_max, width, height, eye_dist=args.eye_dist))
                    for _ in range(200):
                        api.traverse_grid(grid, d_tris, d_rays, d_hits, n_rays)
                    mem.synchronize()
                    ms = []
                    for f in range(frames):
                        r = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, width, height, eye_dist=args.eye_dist, yaw=0.005 * speed * f, strafe=0.005 * speed * f)
                        if refill and (f // refill) % 2:
                            r = np.ascontiguousarray(r.reshape(height, width, 8)[::-1].reshape(n_rays, 8))
                        mem.copy_h2d(d_rays, r)
                        ms.append(api.profile(lambda: api.traverse_grid(grid, d_tris, d_rays, d_hits, n_rays), mem))
                    return round(float(np.mean(ms[8:])), 5)
                def pair(speed, refill=0):
                    # the two policies in turn, twice (default order first: whichever loop runs first after a pause runs a few per cent slower)
                    a0 = frames_ms(speed, 0, refill); a1 = frames_ms(speed, -1, refill); b0 = frames_ms(speed, 0, refill); b1 = frames_ms(speed, -1, refill)
                    return {"ms_per_frame": round((a1 + b1) / 2, 5), "ms_per_frame_default_order": round((a0 + b0) / 2, 5), "runs": {"policy": [a1, b1], "default_order": [a0, b0]}}
                tile_order["moving_camera"] = {
                    "viewer_speed": pair(1.0), "quarter_speed": pair(0.25), "frozen": pair(0.0), "refilled_every_8th_frame": pair(0.0, refill=8),
                    "how": "mean traversal ms (HIP events) of frames 9-32 of a loop that writes each frame's rays into the one ray buffer and synchronises per frame, two loops per "
                           "policy in turn; viewer speed = 0.005 rad turn + 0.005 scene diagonals sideways per frame (main.cpp:579-586: one mouse pixel, one key event).  An order that is "
                           "stale again within four launches is given up at once (64 launches without, doubling): a moving camera runs in the default order, with the share of "
                           "tiles that start with four lanes per ray chosen by measurement (traverse.hip, share trial); the default_order loops run the same trial"}
                mem.copy_h2d(d_rays, scene.make_rays_primary(grid.bbox_min, grid.bbox_max, width, height, eye_dist=args.eye_dist))      # the batch of the line again
        except Exception as e:                                       # (an option the library does not know: older build)
            log(f"[bench] tile order block skipped: {e}")
        finally:
            try: mem.set_option("traverse.tile_order", -1)
            except Exception: pass

    # ---- independent batches in flight (extension; outside the timed region, never `value`) -------------------------------
    # One launch over 1M rays keeps the machine full for half of its time, the rest is the drain of its last wavefronts.  A caller
    # with independent batches puts each on a stream of its own: contexts 1.. traverse with context 0's traversal image.
    pipelined = None
    if args.inflight > 1 and not multi and not bin_rays and args.image:
        try:
            k = args.inflight
            streams = [torch.cuda.Stream() for _ in range(k)]
            lanes = [(mem, grid, d_hits)]
            for _ in range(1, k):
                m = api.MemManager(keep=True, device=device)
                lanes.append((m, api.share_traversal(m, grid), m.alloc(16 * n_rays)))
            for (m, _g, _h), st in zip(lanes, streams):
                m.use_stream(st.cuda_stream)
            for i in range(8 * k):
                m, g, h = lanes[i % k]; api.traverse_grid(g, d_tris, d_rays, h, n_rays)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            launches = max(args.steps * k, 200) if n_rays <= (1 << 22) else args.steps * k      # long enough for a steady state
            for i in range(launches):
                m, g, h = lanes[i % k]; api.traverse_grid(g, d_tris, d_rays, h, n_rays)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            same = True
            for m, _g, h in lanes[1:]:
                other = m.download(h, api.HIT_DTYPE, n_head)
                same = same and bool((other["id"] == hits["id"]).all() and (other["t"].view(np.uint32) == hits["t"].view(np.uint32)).all())
            pipelined = {"in_flight": k, "steps": launches, "ms_per_step": round(dt * 1e3 / launches, 5),
                         "value": round(n_rays * launches / dt / 1e6, 2), "unit": "Mrays/s", "hits_identical_to_single_stream": same,
                         # the same algorithmic bytes per batch over the wall time per batch (not a kernel duration: launches overlap)
                         "frac": round(ab["B_image"] / (dt / launches * 1e9) / HBM_PEAK_GBPS, 4), "frac_contract": round(ab["B_ray"] / (dt / launches * 1e9) / HBM_PEAK_GBPS, 4),
                         "walk_frac": round(ab["B_walk"] / (dt / launches * 1e9) / HBM_PEAK_GBPS, 4),
                         "how": "one context (stream, hit buffer) per call in flight, all over the traversal image of context 0 (hagrid_share_traversal); "
                                "the next launch fills the drain of the previous one.  NOT the headline: `value` is one call at a time"}
            for m, _g, h in lanes[1:]:
                m.use_stream(None); m.close()
            mem.use_stream(None)
            api._current = mem
        except Exception as e:                                   # an extra: never takes the bench line with it
            log(f"[bench] pipelined block skipped: {e}")
            mem.use_stream(None)

    # ---- the launch's critical path, measured (outside the timed region; never `value`) --------------------------------------------------------
    # A launch of a few rounds of resident wavefronts cannot end before its longest dependent chain does -- the 8 x 8 tile whose longest ray crosses the most cells,
    # gather after gather.  That chain is timed ALONE: the 64 rays of each of the eight tiles with the longest ray (step counts of the statistics pass) traversed as
    # a launch of their own on an otherwise empty GPU, as one wavefront (one ray per lane, four once 16 are left: how the tile runs inside the full launch) and as
    # four wavefronts with four lanes per ray from the start; an empty launch (64 rays that miss the grid) gives the launch overhead to subtract.
    critical_path = None
    if ray_kind == "primary" and n_rays == width * height and n_rays <= (1 << 22) and world == 1 and not bin_rays and args.image and not args.no_order_compare:
        try:
            d_steps = mem.alloc(4 * n_rays)
            api.traverse_grid_stats(grid, d_tris, d_rays, d_hits, n_rays, d_steps)
            steps = mem.download(d_steps, np.int32, n_rays).reshape(height // 8, 8, width // 8, 8)
            mem.free(d_steps)
            tile_max = steps.max(axis=(1, 3))
            worst = np.argsort(-tile_max.ravel())[:8]
            rays_all = mem.download(d_rays, np.float32, 8 * n_rays).reshape(height, width, 8)
            d_r64 = mem.alloc(32 * 64); d_h64 = mem.alloc(16 * 64)
            def alone(rays64, quad):
                mem.set_option("traverse.quad_tail", quad); mem.set_option("traverse.tile_order", 0)
                mem.copy_h2d(d_r64, np.ascontiguousarray(rays64.reshape(64, 8)))
                for _ in range(3): api.traverse_grid(grid, d_tris, d_r64, d_h64, 64)
                return min(api.profile(lambda: api.traverse_grid(grid, d_tris, d_r64, d_h64, 64), mem) for _ in range(7))
            miss = np.zeros((64, 8), np.float32); miss[:, 0:3] = grid.bbox_max + 10.0; miss[:, 4:7] = 1.0; miss[:, 7] = 1.0
            overhead = alone(miss, 0)
            one, four, nsteps = [], [], []
            for t in worst:
                ty, tx = divmod(int(t), width // 8)
                r64 = rays_all[8 * ty:8 * ty + 8, 8 * tx:8 * tx + 8]
                one.append(alone(r64, 0)); four.append(alone(r64, 100)); nsteps.append(int(tile_max.ravel()[t]))
            mem.free(d_r64); mem.free(d_h64)
            cp1 = max(one) - overhead; cp4 = max(four) - overhead
            critical_path = {"critical_path_ms": round(cp1, 5), "critical_path_ms_four_lanes_per_ray": round(cp4, 5), "launch_overhead_ms": round(overhead, 5),
                             "steps_of_the_longest_rays": nsteps, "tile_alone_ms": [round(x, 5) for x in one], "tile_alone_ms_four_lanes_per_ray": [round(x, 5) for x in four],
                             "how": "the 64 rays of each of the eight 8x8 tiles with the longest ray (cells + triangle tests of the statistics pass) traversed as a launch of their own "
                                    "(HIP events, best of 7, minus an empty 64-ray launch): the time of the launch's longest dependent chain with the machine to itself.  The "
                                    "full launch cannot be shorter; ms_per_step / critical_path_ms says how much of it is that chain"}
        except Exception as e:
            log(f"[bench] critical path block skipped: {e}")
        finally:
            for k in ("traverse.quad_tail", "traverse.tile_order"):
                try: mem.set_option(k, -1)
                except Exception: pass

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = total_rays / (ms_per_step * 1e3)                   # Mrays/s, whole job
        kernel_ms = kernel_ms_total / args.steps                   # average duration of a step on the launch stream (HIP events)
        achieved = ab["B_ray"] / (kernel_ms * 1e6)                 # GB/s of algorithmic bytes (SURVEY formula), rank-0 kernel
        achieved_img = ab["B_image"] / (kernel_ms * 1e6)           # GB/s of the bytes the image kernel gathers for the same walk
        peak = mem.bandwidth_probe(1 << 30, 5)                     # measured in this process: float4 copy / triad over 1 GiB arrays
        traffic = None; traffic_source = None; l2_hit = None
        # HBM bytes per launch and the other counters of `binding`: collected in separate rocprofv3 --pmc passes of this very command
        # (tools/gpu_traffic_config.sh) and committed as profiles/traffic_config<C>.json together with the hash of the kernel sources and the
        # ray count of the launches it measured.  A file measured on other sources or another batch is refused: traffic = null, traffic_source says why.
        from hagrid_amd import build as _build
        src_hash = _build.source_hash()
        # (a batch of another ray kind than the configuration's own -- `--config clustered --rays aimed` -- has a counter file of its own)
        tpath = os.path.join(ROOT, "profiles", f"traffic_config{args.config}{'' if ray_kind == cfg['rays'] else '_' + ray_kind}.json")
        counters = None
        std_shape = (args.image == 2 and n_tris == cfg["tris"] and (ray_kind in ("incoherent", "aimed") or (width, height) == (cfg.get("width"), cfg.get("height")))
                     and (top_density, snd_density, expansion, bool(compress)) == (cfg["params"].get("top_density", 0.12), cfg["params"].get("snd_density", 2.4), cfg["params"].get("expansion", 3), bool(cfg["params"].get("compress", False)))
                     and not args.opts)
        if os.path.exists(tpath) and world == 1 and std_shape:
            try:
                tj = json.load(open(tpath))
                if tj.get("source_hash") != src_hash:
                    traffic_source = f"stale: {os.path.relpath(tpath, ROOT)} measured kernel sources {tj.get('source_hash', 'unknown')}, this run has {src_hash}"
                elif tj.get("rays", n_rays) != n_rays:
                    traffic_source = f"other batch: {os.path.relpath(tpath, ROOT)} measured launches of {tj.get('rays')} rays, this run has {n_rays}"
                else:
                    # fabric-side bytes: by request size where that pass exists; else FETCH_SIZE raw (+ WRITE_SIZE) for the gather kernels -- their L2 misses
                    # fetch 64-byte sectors (52.8 / 62.8 bytes per miss in profiles/r4a) -- and the x2-corrected figure only in round 3's file
                    traffic = tj.get("hbm_bytes_per_launch_by_request_size") or (tj.get("hbm_bytes_per_launch_raw") if "counters" in tj else tj.get("hbm_bytes_per_launch"))
                    l2_hit = tj.get("l2_hit_rate"); counters = tj.get("counters")
                    traffic_source = f"{os.path.relpath(tpath, ROOT)} (kernel sources {src_hash}, commit {tj.get('commit', '?')}): " + tj.get("source", "")
            except Exception as e:
                traffic = None; traffic_source = f"unreadable: {e}"
        fmt = mem.image_format(grid) if args.image else {}
        tail = fmt.get("slim_id_bits") and "traverse.tail=0" not in args.opts and "traverse.variant" not in args.opts
        kernel_name = ("traverse_kernel_tail" if tail else "traverse_kernel_img") if args.image else "traverse_kernel_v2"
        # what the roofline is quoted on: with a traversal image the bytes the image kernel gathers for the walk (B_image: 48 per ray +
        # record bytes per cell + 48 per test + 4 per id of a by-index list) -- `frac` then cannot exceed what a copy reaches; the
        # SURVEY 8(d) formula on the CONSTRUCTION format (entries + cells + ids + triangles, which this kernel never reads) is
        # carried next to it as `*_contract`.  Without an image the two are the same thing.
        ach = achieved_img if args.image else achieved
        head_achieved, head_frac, head_kind = roofline_headline(ach, traffic, kernel_ms)
        cells_b = grid.num_cells * (16 if compressed else 32)
        image_b = mem.image_bytes(grid)
        out = {
            "metric": "Mrays/s (primary traversal) + grid build ms, 1M-tri scene @1/2/4/8 MI355X",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"BASELINE.json configs[{args.config - 1}]: " if args.config <= 5 else "") + f"{cfg['baseline']} -- {'clustered' if clustered else ('stadium' if stadium else 'soup')}-{n_tris} triangles, "
                                   + (f"{ray_kind} rays, {total} in the batch" + (f" ({width}x{height})" if ray_kind not in ("incoherent", "aimed") else ""))
                                   + (f", rays [{first}, {first + n_rays}) = the share of rank {shard[0]} of {shard[1]}, on ONE GPU" if shard else f", sharded over {world} GPU(s)" if scaling == "strong" else f" per GPU x {world} GPU(s)")
                                   + f"; td {top_density} sd {snd_density} alpha {args.alpha} exp {expansion}" + (" compress" if compress else ""),
                       "baseline_config": args.config, "shard": args.shard, "rays_total": int(total_rays), "rays_rank0": n_rays, "triangles": n_tris, "ray_binning": bin_rays,
                       "traversal_image": "off (construction format)" if not args.image else (lambda f: f"{record_bytes}-byte slim records, " + ("uniform layout (a record per voxel, table-free)" + (" with the table layout next to it for binned batches" if f.get("two_layouts") else "") if f.get("uniform") else "general layout (a record per voxel-map entry: links, wide records)" if f.get("general") else "table layout (a block of records per top-level cell)") + ", built by setup_traversal")(mem.image_format(grid)),
                       "ray_packets": "8x8 pixel tiles, row length detected on the device (kept per ray buffer, looked for again every 16th call; buffer stays in image order); from the second launch over a buffer on the tiles are dispatched longest first, by the costs the previous launches left (`tile_order`)",
                       "eye_dist_diagonals": args.eye_dist, "parallelism": f"ray-sharded x{world} ({scaling}), grid broadcast once",
                       "grid": grid.summary(), "device": info},
            "build_ms": None if build_ms is None else round(build_ms, 3),                   # mean of --build-iter full constructions after two warm-up builds
            "build_ms_min": None if build_ms is None else round(build_ms_min, 3),
            "grid_broadcast_ms": round(t_bcast, 3),
            "setup_traversal_ms": round(setup_ms, 3),
            "hits_sha256": hits_sha,
            "hit_fraction": round(tot_hits / total_rays, 4), "rays_entering_grid": round(tot_in / total_rays, 4),
            # `achieved` / `frac`: measured HBM bytes per launch over the kernel time against the HBM peak (roofline_headline above); `binding` names what
            # does bind, from the counters of this configuration measured on these kernel sources (tools/gpu_traffic_config.sh): every fraction there is
            # against what the part delivers for that access pattern.
            "roofline": {
                "bound": "hbm", "achieved": head_achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": head_frac, "frac_kind": head_kind,
                # the algorithmic figures next to it: what the kernel gathers (image records, triangles, ids of long lists) and the SURVEY.md 8(d) formula on the
                # construction format (entries + cells + ids + triangles, which this kernel never reads); neither is a bound for a cache-resident working set
                "achieved_image": round(ach, 1), "frac_image": round(ach / HBM_PEAK_GBPS, 4),
                "bytes": "B_image: what the traversal-image kernel gathers (DESIGN.md 4.2)" if args.image else "B_ray: SURVEY.md 8(d) on the construction format",
                "achieved_contract": round(achieved, 1), "frac_contract": round(achieved / HBM_PEAK_GBPS, 4),     # SURVEY.md 8(d) formula, construction format
                "traffic": traffic, "traffic_source": traffic_source,
                "binding": None,
                "hbm_measured": None if traffic is None else round(traffic / (kernel_ms * 1e6), 1),
                "hbm_measured_frac": None if traffic is None else round(traffic / (kernel_ms * 1e6) / HBM_PEAK_GBPS, 4),
                "l2_hit_rate": l2_hit,
                "peak_measured": {"copy": round(peak["copy_GBps"], 1), "triad": round(peak["triad_GBps"], 1), "unit": "GB/s",
                                  "frac_of_copy": round(head_achieved / max(peak["copy_GBps"], 1e-9), 4)},
                "kernel": kernel_name, "kernel_ms": round(kernel_ms, 5), "kernel_sources": src_hash,
                "bytes_per_ray": round((ab["B_image"] if args.image else ab["B_ray"]) / n_rays, 1),
                "bytes_per_ray_contract": round(ab["B_ray"] / n_rays, 1),
                "walk_achieved": round(ab["B_walk"] / (kernel_ms * 1e6), 1),
                "walk_frac": round(ab["B_walk"] / (kernel_ms * 1e6) / HBM_PEAK_GBPS, 4),
                "walk_target": 0.40,
                "walk_achieved_image": round(ab["B_image_walk"] / (kernel_ms * 1e6), 1)},
            "pipelined": pipelined,
            "tile_order": tile_order,
            "roofline_build": build_block,
            "memory": {"cells": cells_b, "entries": 4 * grid.num_entries, "refs": 4 * grid.num_refs, "tris": 48 * n_tris,
                       "traversal_image": image_b, "releasable_after_setup_traversal": cells_b + 4 * grid.num_entries, "rays": 32 * n_rays, "hits": 16 * n_rays, "pool_now": mem.usage(), "pool_peak": mem.max_usage(),
                       "unit": "bytes", "reference": "main.cpp:523-533"},
        }
        working_set = image_b + 48 * n_tris + 4 * grid.num_refs if args.image else cells_b + 4 * grid.num_entries + 4 * grid.num_refs + 48 * n_tris
        if counters:
            res, top, top_frac = binding_resources(counters, kernel_ms, working_set, traffic)
            limiter, waiting = binding_limiter(res, top, top_frac)
            out["roofline"]["binding"] = {"resource": top, "frac": top_frac, "limiter": limiter, "resources": res, "working_set_bytes": working_set,
                                          "how": "counters per launch from separate rocprofv3 --pmc passes on these kernel sources (traffic_source), divided by this run's kernel time; "
                                                 "`resource` = the most used throughput resource; `limiter` = memory_latency when the wavefronts wait for memory >= 60 % of their time and no "
                                                 "throughput fraction reaches 0.95 (dependent gathers at the occupancy the kernel has: fewer fetches do not make such a launch faster)"}
            if counters.get("TCP_TCC_READ_REQ_LATENCY_sum") and counters.get("TCP_TCC_READ_REQ_sum"):
                # Little's law on the vector L1s' requests to L2: requests in flight = sum of their latencies / kernel time; against the rays that can have one in flight
                lat_cycles = counters["TCP_TCC_READ_REQ_LATENCY_sum"] / counters["TCP_TCC_READ_REQ_sum"]
                in_flight = counters["TCP_TCC_READ_REQ_LATENCY_sum"] / (kernel_ms * 1e-3 * SHADER_CLOCK_HZ)
                out["roofline"]["binding"]["littles_law"] = {
                    "l1_miss_latency_cycles": round(lat_cycles, 1), "l1_miss_latency_ns": round(lat_cycles / SHADER_CLOCK_HZ * 1e9, 1), "requests_in_flight": round(in_flight, 1),
                    "requests_in_flight_per_cu": round(in_flight / NUM_CUS, 2),
                    "request_rate_from_latency": round(in_flight / (lat_cycles / SHADER_CLOCK_HZ) / 1e9, 2), "request_rate_measured": round(counters["TCP_TCC_READ_REQ_sum"] / (kernel_ms * 1e-3) / 1e9, 2),
                    "unit": "G requests/s", "what": "vector-L1 misses: average latency (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ) and the number in flight (sum of latencies / kernel time); "
                            "rate = in flight / latency by construction -- a launch whose rays each wait for ONE dependent gather at a time raises it only with more rays in flight "
                            "(resident wavefronts x live lanes) or a shorter latency, not with fewer bytes"}
            if critical_path:
                critical_path["ms_per_step_over_critical_path"] = round(ms_per_step / critical_path["critical_path_ms"], 3) if critical_path["critical_path_ms"] > 0 else None
                out["roofline"]["binding"]["critical_path"] = critical_path
            out["roofline"]["bound"] = top if top in ("hbm_bytes",) else (f"{top} (not hbm bytes: see binding)" if limiter == top else
                                                                         f"memory latency of dependent gathers (wavefronts wait {waiting:.2f} of their time; most used resource: {top} {top_frac:.2f}; see binding)")
        else:
            out["roofline"]["binding"] = {"resource": None, "why": traffic_source or "no counter file for this configuration and batch (tools/gpu_traffic_config.sh)"}
            if critical_path:
                critical_path["ms_per_step_over_critical_path"] = round(ms_per_step / critical_path["critical_path_ms"], 3) if critical_path["critical_path_ms"] > 0 else None
                out["roofline"]["binding"]["critical_path"] = critical_path
        # ---- CPU baseline + parity check: the oracle on the SAME grid, rank 0, N = 1 only ---------------------------------
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            d = grid.download()
            G = O.Grid.from_arrays(d["entries"], d["ref_ids"], d["cells"], d["small_cells"], d["bbox_min"], d["bbox_max"], d["dims"], d["shift"], d["offsets"])
            # all host cores: worker threads take 4096-ray chunks from a shared counter and are pinned (oracle/hagrid_oracle.c run_jobs_on) -- once to
            # one hardware thread per physical core, once to every hardware thread; the better of the two is `value`
            phys = O.physical_cpus(); allhw = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
            # the container may hold less CPU time than the cores it sees (cgroup quota): more threads than that only take turns
            quota = None
            try:
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                quota = None if q == "max" else float(q) / float(per)
            except Exception:
                try:
                    q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    quota = q / per if q > 0 else None
                except Exception:
                    pass
            if quota and quota < len(phys):
                phys = phys[:max(1, int(quota + 0.999))]; allhw = allhw[:max(1, int(2 * quota + 0.999))]
            cores = len(phys)
            probe = min(65536, n_head)
            t0 = time.perf_counter(); oh, _ = G.traverse(tris_host, rays_head[:probe], nthreads=cores, cpus=phys); t_probe = time.perf_counter() - t0
            sample_n = int(min(n_head, max(probe, probe * 0.5 * args.cpu_seconds / max(t_probe, 1e-6))))
            def timed(cpus):
                t0 = time.perf_counter(); h, _ = G.traverse(tris_host, rays_head[:sample_n], nthreads=len(cpus), cpus=cpus); t = time.perf_counter() - t0
                reps = 1
                while t < 0.25 * args.cpu_seconds and reps < 512:        # the whole sample is too small: repeat it
                    t0 = time.perf_counter(); G.traverse(tris_host, rays_head[:sample_n], nthreads=len(cpus), cpus=cpus); t += time.perf_counter() - t0; reps += 1
                return h, sample_n * reps / t / 1e6, reps
            oh, rate_phys, reps = timed(phys)
            rate_all = None
            if len(allhw) > len(phys):
                _, rate_all, _ = timed(allhw)
            same_id = bool((hits["id"][:sample_n] == oh["id"]).all())
            same_t = bool((hits["t"][:sample_n].view(np.uint32) == oh["t"].view(np.uint32)).all())
            # one thread: a bounded slice of the same rays (about a third of the all-cores budget)
            one_n = int(min(sample_n, 16384))
            t0 = time.perf_counter(); G.traverse(tris_host, rays_head[:one_n], nthreads=1); t1 = time.perf_counter() - t0
            one_n = int(min(sample_n, max(one_n, one_n * 0.3 * args.cpu_seconds / max(t1, 1e-6))))
            t0 = time.perf_counter(); G.traverse(tris_host, rays_head[:one_n], nthreads=1); t1 = time.perf_counter() - t0
            rate_one = one_n / t1 / 1e6
            best, used = (rate_all, len(allhw)) if rate_all and rate_all > rate_phys else (rate_phys, len(phys))
            out["cpu_baseline"] = {"value": round(best, 4), "unit": "Mrays/s", "cores": used, "kind": "port",
                                   "sample": f"first {sample_n} rays of the batch x{reps}, oracle traversal of the GPU-built grid, {used} pinned threads taking 4096-ray chunks from a shared counter",
                                   "physical_cores": {"value": round(rate_phys, 4), "threads": len(phys), "scaling_vs_one_thread": round(rate_phys / rate_one, 1),
                                                      "efficiency": round(rate_phys / rate_one / len(phys), 3)},
                                   "all_hardware_threads": None if rate_all is None else {"value": round(rate_all, 4), "threads": len(allhw), "scaling_vs_one_thread": round(rate_all / rate_one, 1)},
                                   "single_thread": {"value": round(rate_one, 4), "unit": "Mrays/s", "cores": 1, "sample": f"first {one_n} rays of the batch"},
                                   "host": {"hardware_threads": os.cpu_count(), "physical_cores_visible": len(O.physical_cpus()), "cgroup_cpu_quota_cores": quota}}
            if n_tris <= 1_000_000:      # CPU construction of the same grid, one core (the oracle's passes are scalar)
                t0 = time.perf_counter()
                Gc = O.Grid.full(tris_host, top_density, snd_density, args.alpha, expansion, compress)
                out["cpu_baseline"]["cpu_build_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
                out["cpu_baseline"]["cpu_build_matches_gpu"] = bool(Gc.summary() == grid.summary())
            out["parity"] = {"rays_checked": sample_n, "ids_identical": same_id, "t_bit_identical": same_t}
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
