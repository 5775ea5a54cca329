"""DEV TOOL: the timeline of ONE construction from a rocprofv3 kernel trace (tools/gpu_build_timeline.sh): every dispatch of the last build_all of tools/dev_build_time.py with
its duration and the GAP since the previous dispatch ended -- kernel time against host round trips and launch latencies.
usage: python tools/dev_build_timeline.py TRACE.csv [--all]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "").replace("hagrid_impl::", ""))[:44]
# a construction starts with bbox_partials
starts = [i for i, r in enumerate(rows) if "bbox_partials" in r["Kernel_Name"]]
if len(starts) < 2: raise SystemExit("no construction found in the trace")
a, b = starts[-2], starts[-1]                     # the last COMPLETE construction
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"]); prev_end = t0
tot_k = tot_gap = 0; big = []
stage = {"build": [0, 0, 0], "merge": [0, 0, 0], "flatten": [0, 0, 0], "expand": [0, 0, 0]}
cur = "build"
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); n = name(r)
    if "merge_counts" in n or n.startswith("ip_") and cur == "build": cur = "merge"
    if "collapse" in n or "flatten" in n: cur = "flatten"
    if "overlap_step" in n or "fill_voxel_cells" in n or "expand_" in n: cur = "expand"
    gap = max(0, s - prev_end); dur = e - s
    tot_k += dur; tot_gap += gap; stage[cur][0] += dur; stage[cur][1] += gap; stage[cur][2] += 1
    if "--all" in sys.argv or gap > 8000 or dur > 40000: big.append((s - t0, dur, gap, n))
    prev_end = max(prev_end, e)
print(f"one construction: {len(seg)} launches, span {(prev_end - t0) / 1e3:.1f} us, kernels {tot_k / 1e3:.1f} us, gaps {tot_gap / 1e3:.1f} us")
for k, v in stage.items(): print(f"  {k:8s} kernels {v[0] / 1e3:8.1f} us  gaps {v[1] / 1e3:7.1f} us  launches {v[2]}")
print("  dispatches with a gap > 8 us in front of them or longer than 40 us:")
for off, dur, gap, n in big: print(f"    +{off / 1e3:8.1f} us  {n:44s} {dur / 1e3:7.1f} us   gap {gap / 1e3:6.1f} us")
