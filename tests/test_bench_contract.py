"""The bench lines committed under profiles/ (outputs of bench.py on an MI355X) carry every field the
driver's contract names, with consistent values; and bench.py's command line keeps the flags the driver
passes.  No GPU needed."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[123]_bench_*.json")))
LINES = [p for p in LINES if not p.endswith("_summary.json")]
HEADLINE = os.path.join(ROOT, "profiles", "r03_bench_config4.json")


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_committed_bench_line_has_the_contract_fields(path):
    with open(path) as f:
        text = f.read().strip()
    assert text.count("\n") == 0, "one JSON line"
    j = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["vs_baseline"] is None and j["higher_is_better"] is True and j["scaling"] in ("weak", "strong") and j["data"] == "synthetic"
    assert isinstance(j["config"].get("workload"), str) and "model" not in j["config"]
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    issue_bound = r["bound"] == "valu"
    if issue_bound:
        # a pass bound by vector-instruction issue: counted instructions x 4 cycles over the 1 024 SIMDs at the measured
        # clock; the byte figure stays beside it
        assert r["unit"] == "G wave-instr/s" and 1800.0 < r["clock_mhz"] < 2600.0 and abs(r["peak"] - 1024 * r["clock_mhz"] / 4e3) < 0.1
        assert abs(r["achieved"] - r["valu_insts_per_step"] / (r["kernel_ms"] * 1e6)) < 0.5
        assert os.path.exists(os.path.join(ROOT, r["counters_source"].split(" ")[0]))
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.0 < r["frac"] < 1.0
        r = r["hbm"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.0 < r["frac"] < 1.0
    if r["traffic"] is not None:
        # measured bytes within a factor of the algorithmic ones; the issue-bound line's count leaves out what its step
        # moves beside the lists (the ballot words zeroed for every group, the work lists) and says so
        assert 0.5 < r["traffic"] / r["algorithmic_bytes"] < (1.5 if not issue_bound else 2.0)
        assert not issue_bound or "traffic_note" in r
    if "all_bounds" in r:  # lines of a cull with the instance pre-cull: the figure with every bound counted as read
        a = r["all_bounds"]
        assert a["algorithmic_bytes"] - a["bounds_bytes_not_touched"] <= r["algorithmic_bytes"] + 4 * j["config"].get("pairs_reaching_hiz", 0)
        assert abs(a["frac"] - a["achieved"] / r["peak"]) < 1e-3
    c = j["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["unit"] == j["unit"] and c["cores"] >= 1
    assert j["value"] > c["value"] > 0


def test_headline_line_is_the_metric_baseline_names():
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    with open(HEADLINE) as f:
        j = json.load(f)
    assert j["n_gpus"] == 1 and j["dtype"] == "f32"
    assert "meshlet-instances" in j["metric"] and "meshlet-instances" in base["metric"]
    assert j["config"]["meshlet_instances_per_gpu"] == 100_000_000
    assert j["roofline"]["frac"] >= 0.6  # the target the measurement plan sets for the combined cull + compact step


def test_headline_traffic_agrees_with_the_algorithmic_bytes():
    """The counter passes of the headline command (profiles/r03_bench_config4_summary.json: FETCH_SIZE doubled,
    WRITE_SIZE as is, per kernel and launch) against the bytes the line prices the step at: the counters are
    exact on this chip (a calibration kernel reads back its own request count), so the ratio has to be tight."""
    with open(HEADLINE) as f:
        r = json.load(f)["roofline"]
    with open(os.path.join(ROOT, "profiles", "r03_bench_config4_summary.json")) as f:
        summ = json.load(f)
    step = ("instance_precull_kernel", "cull_mask_kernel", "scan_tiles_kernel", "compact_kernel", "emit_commands_kernel")
    traffic = sum(row.get("fetch_bytes_corrected", 0) + row.get("write_bytes_corrected", 0)
                  for name, row in summ.items() if any(k in name for k in step))
    assert 0.95 <= traffic / r["algorithmic_bytes"] <= 1.25, traffic / r["algorithmic_bytes"]
    if r["traffic"] is not None:  # the line quotes the summary that was committed when it ran: counter passes repeat within a per cent
        assert abs(r["traffic"] - traffic) / traffic < 0.01
    trace = {}
    with open(os.path.join(ROOT, "profiles", "r03_bench_config4_kernel_stats.csv")) as f:
        for line in f.read().splitlines()[1:]:
            cells = line.rsplit(",", 5)
            trace[cells[0]] = float(cells[2])
    per_step = sum(v for k, v in trace.items() if any(s in k for s in step))
    assert abs(per_step / 1e3 - r["kernel_ms"]) / r["kernel_ms"] < 0.05, "the trace's kernel durations add up to the step the line reports"


def test_bench_command_line_keeps_the_driver_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload"):
        assert flag in out.stdout


def _bench_module():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _run_bench(argv, env=None):
    full = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        full.pop(k, None)
    full.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=300, env=full)


@pytest.mark.parametrize("world", [2, 4])
def test_gpus_n_without_a_launcher_starts_its_own_ranks(world):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: N child ranks rendezvous over gloo
    (the CPU-only workload), stdout carries exactly rank 0's JSON line, exit status 0."""
    out = _run_bench(["--gpus", str(world), "--workload", "rendezvous"])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [x for x in out.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == world and j["ranks_seen"] == (1 << world) - 1
    assert j["instances"] == 781250, "the contiguous shards of all ranks cover the scene exactly once"


def test_a_failing_rank_fails_the_launcher():
    out = _run_bench(["--gpus", "2", "--workload", "rendezvous"], env={"MCULL_BENCH_FAIL_RANK": "1"})
    assert out.returncode == 3, (out.returncode, out.stderr[-2000:])


def test_under_a_launcher_no_ranks_are_spawned():
    """With WORLD_SIZE set (torch.distributed.run, the driver's way) the process is a rank, not a launcher."""
    port = _bench_module().free_port()
    out = _run_bench(["--gpus", "1", "--workload", "rendezvous"],
                     env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    assert out.returncode == 0 and json.loads(out.stdout.strip())["n_gpus"] == 1
    bad = _run_bench(["--gpus", "2", "--workload", "rendezvous"],
                     env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    assert bad.returncode != 0 and "does not match WORLD_SIZE" in bad.stderr


def test_strong_scaling_shards_concatenate_to_the_single_rank_scene():
    """The counter-based scene generator: the shards ranks generate locally equal slices of the scene one
    rank generates (checked on the CPU generator; ragged shard boundaries inside generator chunks)."""
    import torch

    b = _bench_module()
    n, per = 3 * b.SCENE_CHUNK + 217, 8
    whole = b.make_scene_range("cpu", 0, n, per, seed=42)
    for world in (1, 2, 3, 8):
        cuts = [b.shard_range(n, world, r) for r in range(world)]
        assert cuts[0][0] == 0 and sum(c[1] for c in cuts) == n
        assert all(cuts[r][0] + cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
        parts = [b.make_scene_range("cpu", f, k, per, seed=42) for f, k in cuts]
        for i in range(3):
            assert torch.equal(torch.cat([p[i] for p in parts]), whole[i])
    other = b.make_scene_range("cpu", 0, n, per, seed=43)
    assert not torch.equal(other[0], whole[0])
    # unit axes, coneW in [0, 1] with a share pinned to 1, radii in [0.01, 0.05]
    axis_len = whole[0][:, :3].norm(dim=1)
    assert torch.allclose(axis_len, torch.ones_like(axis_len), atol=1e-5)
    assert 0.05 < float((whole[0][:, 3] == 1.0).float().mean()) < 0.15
    assert float(whole[1][:, 3].min()) >= 0.01 and float(whole[1][:, 3].max()) <= 0.05
