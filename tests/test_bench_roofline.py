"""The roofline block of the bench line, recomputed on the CPU from one committed set of evidence (profiles/r4z/: the counter files of
tools/gpu_traffic_config.sh as `bench.py` read them, and the bench lines measured with them): every fraction of
`roofline.binding` is a fraction (<= 1 up to measurement noise), the block in the committed line is what the counters give, and the rule that
names the limiter (bench.binding_limiter) says memory latency where the wavefronts wait and nothing is saturated."""
import json, os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LINES = {2: "bench.json", 3: "bench_config3.json", 4: "bench_config4_shard.json", 5: "bench_config5_shard.json"}


def _load(config):
    t = os.path.join(ROOT, "profiles", "r4z", f"traffic_config{config}.json"); l = os.path.join(ROOT, "profiles", "r4z", LINES[config])
    if not (os.path.exists(t) and os.path.exists(l)):
        pytest.skip("no committed evidence for this configuration")
    return json.load(open(t)), json.load(open(l))


@pytest.mark.parametrize("config", [2, 3, 4, 5])
def test_binding_block_follows_from_the_committed_counters(config):
    import bench
    tj, line = _load(config)
    roof = line["roofline"]; b = roof["binding"]
    assert roof["kernel_sources"] == tj["source_hash"], "the line and its counters were measured on the same kernel sources"
    assert line["config"]["rays_rank0"] == tj["rays"]
    # the kernel time of the bench run and of the profiled run agree (rocprof's average is what the contract asks to be compared)
    assert abs(roof["kernel_ms"] - tj["rocprof_kernel_avg_ms"]) / roof["kernel_ms"] < 0.08
    traffic = tj.get("hbm_bytes_per_launch_by_request_size") or tj["hbm_bytes_per_launch_raw"]
    assert roof["traffic"] == pytest.approx(traffic)
    res, top, top_frac = bench.binding_resources(tj["counters"], roof["kernel_ms"], b["working_set_bytes"], traffic)
    for name, r in res.items():
        if "frac" in r:
            assert 0.0 < r["frac"] <= 1.02, (name, r["frac"])
            assert r["frac"] == pytest.approx(b["resources"][name]["frac"], abs=2e-3), name
    assert top == b["resource"] and top_frac == pytest.approx(b["frac"], abs=2e-3)
    limiter, waiting = bench.binding_limiter(res, top, top_frac)
    assert limiter == b["limiter"]
    wt = res["wave_time"]
    assert 0.95 < wt["waiting_for_memory"] + wt["waiting_to_issue"] + wt["issuing"] < 1.05
    # measured HBM bytes stay below the algorithmic bytes of the walk: nothing is re-read from HBM
    assert traffic <= 1.05 * roof["bytes_per_ray"] * tj["rays"] or config == 5
    assert roof["hbm_measured_frac"] == pytest.approx(traffic / (roof["kernel_ms"] * 1e-3) / 8e12, abs=2e-3)


def test_limiter_rule():
    import bench
    busy = {"wave_time": {"waiting_for_memory": 0.48}}
    assert bench.binding_limiter(busy, "valu_issue", 0.66) == ("valu_issue", 0.48)
    waiting = {"wave_time": {"waiting_for_memory": 0.68}}
    assert bench.binding_limiter(waiting, "fabric_fetch_rate", 0.89)[0] == "memory_latency"
    assert bench.binding_limiter(waiting, "hbm_bytes", 0.97)[0] == "hbm_bytes"          # saturated: the resource itself
    assert bench.binding_limiter({}, "valu_issue", 0.7)[0] == "valu_issue"               # no wave-time counters: the most used resource


def test_roofline_headline_is_the_measured_hbm_rate():
    """`roofline.achieved` / `frac` = measured HBM bytes per launch / kernel time / 8 TB/s when counters of these kernel sources exist; the algorithmic figure
    stands in -- and says so -- when they do not."""
    import bench
    a, f, kind = bench.roofline_headline(6840.0, 306.9e6, 0.13468)
    assert a == pytest.approx(2278.7, abs=0.5) and f == pytest.approx(0.2848, abs=1e-3) and kind.startswith("hbm_measured")
    a, f, kind = bench.roofline_headline(6840.0, None, 0.13468)
    assert a == 6840.0 and f == pytest.approx(0.855) and kind.startswith("algorithmic")


def test_clustered_configuration_is_named():
    import bench
    assert bench.CONFIG_NAMES["clustered"] in bench.CONFIGS and bench.CONFIGS[bench.CONFIG_NAMES["clustered"]]["scene"] == "clustered"
    from hagrid_amd import scene
    r = scene.make_rays_aimed([0, 0, 0], [1, 1, 1], 12, 5)
    assert r.shape == (12, 8) and (scene.make_rays_aimed([0, 0, 0], [1, 1, 1], 6, 5, first=6) == r[6:]).all()


def test_bench_started_bare_with_several_gpus_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it re-executes itself under torch.distributed.run.  Here there is no
    GPU, so the two ranks it starts stop at "needs a GPU" -- what is checked is that it is THEY who stop, not the bare process at the WORLD_SIZE test."""
    import subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=env)
    assert "re-executing under torch.distributed.run" in r.stderr
    assert "launch with torch.distributed.run" not in r.stderr
    assert r.stderr.count("bench.py needs a GPU") >= 2 or "local_rank: 1" in r.stderr
