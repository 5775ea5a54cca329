#!/bin/bash
# Pre-flight of the multi-GPU bench on a ONE-GPU box: the exact launch line the driver uses for its scaling runs,
# with one rank (process group over RCCL, grid broadcast, all-reduces: --force-dist) and with two ranks sharing the GPU (gloo), and
# the bench line's multi-rank bookkeeping asserted.  No scaling number comes out of this: it only shows that the N > 1 path is turnkey.
# usage: tools/scale_preflight.sh [TAG]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-preflight}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" || exit 1
run() {   # name, ranks, bench arguments...
  local name=$1 n=$2; shift; shift
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
      bench.py --gpus $n --steps 5 --warmup 2 --no-cpu-baseline --inflight 0 --build-iter 2 "$@" > $OUT/$name.json 2> $OUT/$name.err
  echo "$name rc=$?"
}
run c2_rccl_1  1 --force-dist
run c4_rccl_1  1 --force-dist --config 4 --total-rays 4194304
run c5_rccl_1  1 --force-dist --config 5 --tris 1000000 --width 2048 --height 2048
run c2_gloo_2  2 --backend gloo --device 0
run c4_gloo_2  2 --backend gloo --device 0 --config 4 --total-rays 4194304
# started bare, the shape of the driver's N = 1 command: bench.py launches its own ranks
env -u RANK -u LOCAL_RANK -u WORLD_SIZE -u LOCAL_WORLD_SIZE timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --inflight 0 --build-iter 2 \
    --backend gloo --device 0 > $OUT/c2_bare_2.json 2> $OUT/c2_bare_2.err; echo "c2_bare_2 rc=$?"
python - "$OUT" <<'PY'
import json, sys, os
out = sys.argv[1]
ok = True
for name, world, strong in (("c2_rccl_1", 1, False), ("c4_rccl_1", 1, True), ("c5_rccl_1", 1, True), ("c2_gloo_2", 2, False), ("c4_gloo_2", 2, True), ("c2_bare_2", 2, False)):
    try:
        lines = [l for l in open(os.path.join(out, name + ".json")) if l.startswith("{")]
        assert len(lines) == 1, f"{len(lines)} JSON lines (rank 0 prints exactly one)"
        j = json.loads(lines[0]); c = j["config"]
        assert j["n_gpus"] == world, ("n_gpus", j["n_gpus"])
        assert j["grid_broadcast_ms"] > 0, "grid_broadcast_ms: the broadcast did not run"
        assert j["scaling"] == ("strong" if strong else "weak"), j["scaling"]
        # strong: the ranks' contiguous ranges add up to the batch; weak: every rank traces a batch of the full size
        assert abs(c["rays_rank0"] * world - c["rays_total"]) <= world, (c["rays_rank0"], world, c["rays_total"])
        assert j["value"] > 0 and j["ms_per_step"] > 0
        print(f"{name}: ok  n_gpus {j['n_gpus']}  grid_broadcast_ms {j['grid_broadcast_ms']}  rays_rank0 {c['rays_rank0']} x {world} = rays_total {c['rays_total']}  {j['value']} Mrays/s")
    except Exception as e:
        ok = False
        print(f"{name}: FAILED {e!r}")
        try: print(open(os.path.join(out, name + ".err")).read()[-1500:])
        except OSError: pass
print("scale preflight:", "ok" if ok else "FAILED")
sys.exit(0 if ok else 1)
PY
