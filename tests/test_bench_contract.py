"""bench.py contract: CLI flags exist (CPU), and a reduced-size run prints ONE JSON line with every field the
driver and the judge read (GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


def test_cli_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


@pytest.mark.gpu
def test_reduced_run_prints_one_json_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--tuples", "4000000", "--cpu-sample", "200000", "--no-extras"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["verified"] is True and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "tuples/s" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "u64"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # the two scatter variants of a two-pass join are priced separately; `roofline` itself is the dominant one (pass 1)
    v = r["variants"]
    assert set(v) == {"pass 1", "pass 2"} and abs(v["pass 1"]["achieved"] - r["achieved"]) < 1e-6
    assert r["avg_launch_ms"] == v["pass 1"]["avg_launch_ms"]
    # no traffic figure unless profiles/traffic.json was measured on this size AND on these kernel sources
    assert r["traffic"] is None
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c
    assert c["kind"] in ("reference", "port") and c["cpu_model"] and c["cores"] in (1, 8)
    if c["kind"] == "reference":                      # SURVEY §8d protocol: 8 threads and 1 thread, median of >= 3 runs each
        assert set(c["tuples_per_s"]) == {"8_threads", "1_threads"} and all(len(v) >= 3 for v in c["runs_s"].values())
