#!/bin/bash
# HBM traffic of the construction kernels against the algorithmic bytes: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in
# separate passes over tools/dev_build_time.py (ITERS=1: three constructions of the 1M-triangle grid).  usage: tools/gpu_build_traffic.sh TAG
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-traffic}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
python -c "import __graft_entry__ as g; g.build()" || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && ITERS=1 PYTHONPATH=$ROOT timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $ROOT/$OUT/$C -o pmc -- python $ROOT/tools/dev_build_time.py > $ROOT/$OUT/$C.out 2> $ROOT/$OUT/$C.err)
done
python - <<PY > $OUT/construction_traffic.txt
import csv, glob, collections, json, re
acc = collections.defaultdict(lambda: {"n": 0, "ns": 0.0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % C, recursive=True)
    if not f: print("no output for", C); raise SystemExit
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != C: continue
        name = re.sub(r"\(anonymous namespace\)::|hagrid_impl::|void ", "", r["Kernel_Name"]).split("(")[0][:60]
        a = acc[name]; a[C] += float(r["Counter_Value"])
        if C == "FETCH_SIZE":
            a["n"] += 1; a["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
builds = 3                      # dev_build_time.py with ITERS=1: one warm-up, one staged, one timed construction
tot_f = tot_w = 0.0
rows = []
for name, a in acc.items():
    if name.startswith("__amd") and "fill" not in name and "copy" not in name: continue
    f_mb = a["FETCH_SIZE"] * 2 * 1024 / 1e6 / builds; w_mb = a["WRITE_SIZE"] * 1024 / 1e6 / builds
    tot_f += f_mb; tot_w += w_mb
    rows.append((a["ns"] / builds / 1e3, name, a["n"] / builds, f_mb, w_mb))
print("kernel                                                         calls/build  us/build  fetch(x2) MB  write MB   TB/s")
for us, name, n, f_mb, w_mb in sorted(rows, reverse=True):
    print(f"{name:62s} {n:8.1f} {us:10.1f} {f_mb:12.1f} {w_mb:10.1f} {((f_mb + w_mb) / us if us else 0):6.2f}")
print(f"TOTAL per construction: fetch(x2) {tot_f:.0f} MB + write {tot_w:.0f} MB = {(tot_f + tot_w) / 1e3:.2f} GB")
PY
cat $OUT/construction_traffic.txt
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete
