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
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r01_bench_*.json")))
LINES = [p for p in LINES if not p.endswith("_summary.json")]


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_committed_bench_line_has_the_contract_fields(path):
    with open(path) as f:
        text = f.read().strip()
    assert text.count("\n") == 0, "one JSON line"
    j = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["vs_baseline"] is None and j["higher_is_better"] is True and j["scaling"] == "weak" and j["data"] == "synthetic"
    assert isinstance(j["config"].get("workload"), str) and "model" not in j["config"]
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.0 < r["frac"] < 1.0
    if r["traffic"] is not None:  # measured bytes within a factor of the algorithmic ones
        assert 0.5 < r["traffic"] / r["algorithmic_bytes"] < 1.5
    c = j["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["unit"] == j["unit"] and c["cores"] >= 1
    assert j["value"] > c["value"] > 0


def test_headline_line_is_the_metric_baseline_names():
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    with open(os.path.join(ROOT, "profiles", "r01_bench_config4.json")) as f:
        j = json.load(f)
    assert j["n_gpus"] == 1 and j["dtype"] == "f32"
    assert "meshlet-instances" in j["metric"] and "meshlet-instances" in base["metric"]
    assert j["config"]["meshlet_instances_per_gpu"] == 100_000_000
    assert j["roofline"]["frac"] >= 0.6  # the target the measurement plan sets for the combined cull + compact step


def test_bench_command_line_keeps_the_driver_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload"):
        assert flag in out.stdout
