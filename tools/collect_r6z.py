"""DEV TOOL: copies the evidence of round 6 from gpurun_out/TAG (tools/gpu_round6a.sh, gpu_round6b.sh) into profiles/r6z and profiles/ and prints the figures
the documents quote.  usage: python tools/collect_r6z.py [TAG = r6z]"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r6z"
S = os.path.join(ROOT, "gpurun_out", TAG); D = os.path.join(ROOT, "profiles", "r6z"); os.makedirs(D, exist_ok=True)
def cp(src, dst):
    if os.path.exists(src): shutil.copyfile(src, dst); return True
    print("missing", os.path.relpath(src, ROOT)); return False
for f in ("bench.json", "bench_clustered.json", "bench_stadium.json", "bench_clustered_aimed.json", "bench_clustered_aimed_unbinned.json", "bench_clustered_aimed_4M.json", "bench_config3.json",
          "bench_config4_shard.json", "bench_soup_sd5_4096.json", "bench_config5_shard.json", "bench_config4.json", "bench_config5.json", "build_time_8M.txt", "construction_traffic.txt",
          "scale_preflight.txt", "source_hash.txt", "smoke.log", "pytest_gpu_tail.txt", "policy_regret.txt", "timeline_soup.txt", "timeline_clustered.txt", "timeline_stadium.txt",
          "timeline_all_soup.txt", "build_time_soup.txt", "build_time_clustered.txt", "build_time_stadium.txt", "kernel_stats_soup.csv", "kernel_stats_clustered.csv", "kernel_stats_stadium.csv"):
    cp(os.path.join(S, f), os.path.join(D, f))
if os.path.exists(os.path.join(S, "pytest_gpu.log")):
    open(os.path.join(D, "pytest_gpu_tail.txt"), "w").write("".join(open(os.path.join(S, "pytest_gpu.log")).readlines()[-16:]))
for c in (2, 3, 4, 5, 6, 7):
    d = os.path.join(S, f"config{c}")
    if cp(os.path.join(d, f"traffic_config{c}.json"), os.path.join(D, f"traffic_config{c}.json")):
        shutil.copyfile(os.path.join(d, f"traffic_config{c}.json"), os.path.join(ROOT, "profiles", f"traffic_config{c}.json"))
    cp(os.path.join(d, "summary.txt"), os.path.join(D, f"summary_config{c}.txt"))
    cp(os.path.join(d, "stats", "trace_kernel_stats.csv"), os.path.join(D, f"kernel_stats_config{c}.csv"))
cp(os.path.join(S, "construction_traffic.txt"), os.path.join(ROOT, "profiles", "pmc_r6z_construction_traffic.txt"))
if cp(os.path.join(S, "config6_aimed", "traffic_config6_aimed.json"), os.path.join(D, "traffic_config6_aimed.json")):
    shutil.copyfile(os.path.join(S, "config6_aimed", "traffic_config6_aimed.json"), os.path.join(ROOT, "profiles", "traffic_config6_aimed.json"))
    cp(os.path.join(S, "config6_aimed", "summary.txt"), os.path.join(D, "summary_config6_aimed.txt")); cp(os.path.join(S, "config6_aimed", "stats", "trace_kernel_stats.csv"), os.path.join(D, "kernel_stats_config6_aimed.csv"))
def line(name):
    p = os.path.join(D, name)
    if not os.path.exists(p): return None
    try: return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e: print(name, "unreadable", e); return None
for name in sorted(os.listdir(D)):
    if not name.startswith("bench") or not name.endswith(".json"): continue
    j = line(name)
    if not j: continue
    r = j.get("roofline", {}); b = r.get("binding", {}) or {}
    print(f"{name:30s} {j['value']:9.1f} Mrays/s  ms {j['ms_per_step']:.4f}  frac {r.get('frac')} ({r.get('frac_kind')})  walk {r.get('walk_frac')}  image {r.get('frac_image')}  limiter {b.get('limiter')}  build_ms {j.get('build_ms')}  "
          f"build_frac {(j.get('roofline_build') or {}).get('frac')}  default_order {(j.get('tile_order') or {}).get('ms_per_step_default_order')}")
    mc = (j.get("tile_order") or {}).get("moving_camera")
    if mc: print("   moving camera:", {k: (v.get('ms_per_frame'), v.get('ms_per_frame_default_order')) for k, v in mc.items() if isinstance(v, dict)})
    cp_ = (b.get('critical_path') or {})
    if cp_: print('   critical path:', cp_.get('critical_path_ms'), 'four lanes', cp_.get('critical_path_ms_four_lanes_per_ray'), 'ratio', cp_.get('ms_per_step_over_critical_path'))
    if j.get("pipelined"): print("   pipelined:", json.dumps(j["pipelined"])[:300])
for f in ("build_time_soup.txt", "build_time_clustered.txt", "build_time_stadium.txt", "build_time_8M.txt"):
    p = os.path.join(D, f)
    if os.path.exists(p): print(f, open(p).read()[:200])
p = os.path.join(D, "construction_traffic.txt")
if os.path.exists(p): print(open(p).read().strip().splitlines()[-1])
