"""Condenses the rocprofv3 passes of tools/gpu_traffic_config.sh into traffic_config<C>.json: per launch of the timed traversal kernel
(the statistics launch `traverse_kernel<...>` and -- config 5 -- the primary launch that produces the bounce rays are left out: only the
launches of the most frequent grid size of `traverse_kernel_tail / _img / _v2` count) the HBM bytes (FETCH_SIZE / WRITE_SIZE in KiB, FETCH
doubled per the gfx950 note of MI355X_MICROARCH.md), the L2 hit rate, and the SQ / TCP / TA counters the bench line's roofline block is
computed from.  usage: python tools/summarize_counters.py gpurun_out/TAG/configC C"""
import collections, csv, glob, json, os, sys

out, config = sys.argv[1], int(sys.argv[2])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def timed(name):
    return any(k in name for k in ("traverse_kernel_tail", "traverse_kernel_img", "traverse_kernel_v2"))

counters = {}
launch_ns = []
kernel_name = None
for f in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if timed(r["Kernel_Name"])]
    if not rows:
        continue
    # (config 5: the primary launch that produces the bounce rays runs the same kernel on the same grid: it is the FIRST dispatch of the kernel)
    first_dispatch = min(int(r["Dispatch_Id"]) for r in rows)
    bench = json.load(open(os.path.join(out, "bench.json"))) if os.path.exists(os.path.join(out, "bench.json")) else {}
    bounce = "bounce" in bench.get("config", {}).get("workload", "")
    sizes = collections.Counter(r["Grid_Size"] for r in rows)
    size = sizes.most_common(1)[0][0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in rows:
        if r["Grid_Size"] != size or (bounce and int(r["Dispatch_Id"]) == first_dispatch):
            continue
        kernel_name = r["Kernel_Name"].split("(")[0]
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for c, (v, n) in acc.items():
        counters[c] = {"avg_per_launch": v / n, "launches": n}

stats = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
kernel_avg_ns = None
table = []
if stats:
    for r in csv.DictReader(open(stats[0])):
        table.append((r["Name"][:90], int(r["Calls"]), float(r["AverageNs"]), float(r["TotalDurationNs"])))
        if timed(r["Name"]) and kernel_avg_ns is None:
            kernel_avg_ns = float(r["AverageNs"])            # (the table is sorted by total time: the timed kernel is the first match)

res = {"config": config, "kernel": kernel_name, "counters": {k: round(v["avg_per_launch"], 1) for k, v in sorted(counters.items())},
       "launches": {k: v["launches"] for k, v in sorted(counters.items())}}
bj = os.path.join(out, "bench.json")
if os.path.exists(bj):
    try:
        b = json.load(open(bj))
        res["rays"] = b["config"]["rays_rank0"]; res["bench_kernel_ms"] = b["roofline"]["kernel_ms"]; res["bench_value"] = b["value"]
        res["shard"] = b["config"].get("shard")
    except Exception as e:
        res["bench_error"] = str(e)
res["rocprof_kernel_avg_ms"] = None if kernel_avg_ns is None else round(kernel_avg_ns / 1e6, 5)
c = {k: v["avg_per_launch"] for k, v in counters.items()}
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    f = c["FETCH_SIZE"] * 1024; w = c["WRITE_SIZE"] * 1024
    res["hbm_bytes_per_launch_raw"] = f + w
    res["hbm_bytes_per_launch"] = 2 * f + w                 # gfx950 FETCH_SIZE correction (MI355X_MICROARCH.md, HBM section)
if "TCC_EA0_RDREQ_sum" in c and "WRITE_SIZE" in c:
    # the read requests of the L2s to the fabric by size: what FETCH_SIZE approximates (it tallies a 128-byte request as 64 bytes on gfx950)
    n32 = c.get("TCC_EA0_RDREQ_32B_sum", 0.0); n128 = c.get("TCC_EA0_RDREQ_128B_sum", 0.0); n64 = c["TCC_EA0_RDREQ_sum"] - n32 - n128
    res["fabric_read_requests"] = {"32B": n32, "64B": n64, "128B": n128}
    res["hbm_bytes_per_launch_by_request_size"] = 32 * n32 + 64 * n64 + 128 * n128 + c["WRITE_SIZE"] * 1024
if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
    res["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
from hagrid_amd import build as _build
res["source_hash"] = _build.source_hash()
res["commit"] = os.environ.get("HAGRID_COMMIT", "unknown")
res["source"] = ("separate rocprofv3 --pmc passes (one counter set each, with --kernel-trace only) of `" + open(os.path.join(out, "command.txt")).read().strip().replace(os.getcwd() + "/", "")
                 + "` (tools/gpu_traffic_config.sh); FETCH x2 per the gfx950 note of MI355X_MICROARCH.md; per launch of the timed traversal kernel; NOT measured in the bench run itself")
json.dump(res, open(os.path.join(out, f"traffic_config{config}{os.environ.get('TRAFFIC_SUFFIX', '')}.json"), "w"), indent=1)      # (TRAFFIC_SUFFIX=_aimed: a batch of another ray kind)
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for name, calls, avg, tot in table[:12]:
    print(f"{name:90s} calls {calls:5d} avg_us {avg / 1e3:10.2f} total_ms {tot / 1e6:9.3f}")
print(json.dumps(res, indent=1))
