"""The bench line's contract (GPU box): `python bench.py` prints ONE JSON line with the driver's keys, BASELINE.json's metric and unit,
the two roofline objects, the CPU baseline and the side measurements - run here on a reduced number of steps, as a child process like
the driver runs it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_default_line():
    d = _run(["--steps", "4", "--warmup", "1"])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == base.get("unit", d["unit"]) and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["config"]["streams_per_gpu"] == 4096 and d["config"]["frames_per_step"] == 64
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms_avg"] * 1e-3) / 1e9) < 0.05 * r["achieved"]
    assert abs(d["value"] - 4096 * 64 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["value"] and c["value"] < d["value"]
    assert d["value"] > 30.0                                    # round 1 measured 40; anything below is a regression of the path, not noise
    for k in ("host_io", "t1", "t4", "t8", "serial_calls", "parity_sample", "per_rank_ms", "other_workloads", "host_enqueue_ms_per_step"):
        assert k in d, k
    # counters come from profiles/ and are quoted only when they were collected on the sources that are running (bench.source_hash)
    # (a kernel edit makes the committed counters stale until they are collected again: the line then says so - `traffic_note`, traffic null - and
    # that is a valid line; parity does not depend on a profile being fresh)
    if r["traffic"]:
        assert "traffic_note" not in r and 0 < d["roofline_valu"]["frac"] < 1 and 0 < d["roofline_valu"]["issue"]["insts_per_simd_cycle"] < 1
        assert abs(r["traffic_over_algorithmic"] - r["traffic"] / r["algorithmic_bytes_per_launch"]) < 0.01
    else:
        assert r["traffic"] is None and "roofline_valu" not in d
    assert d["host_io"]["value"] < d["value"] and d["t1"]["value"] < d["value"] and d["t1"]["host_enqueue_ms_per_step"] > 0
    assert d["config"]["input_ready"] is True and 0 < d["host_enqueue_ms_per_step"] < d["ms_per_step"] * 1.5
    # the timed launches' own bytes against the CPU oracle: silent, full-scale, periodic and plain streams, all calls of the run
    assert d["parity_sample"]["frames"] >= 5 * 64 and d["parity_sample"]["differ"] == 0, d["parity_sample"]
    assert len(d["per_rank_ms"]) == 1 and abs(d["per_rank_ms"][0] - d["ms_per_step"]) < 0.02 * d["ms_per_step"]
    for w in ("c3", "c4", "c5", "c96"):
        assert d["other_workloads"][w]["value"] and d["other_workloads"][w]["nonempty"], (w, d["other_workloads"][w])


def test_other_workloads_and_flags():
    d = _run(["--workload", "c3", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    assert d["config"]["channels"] == 2 and d["config"]["stereo_frames_per_step_all_gpus"] == 2048 * 16 and "cpu_baseline" not in d
    d = _run(["--workload", "d1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert "decoded" in d["metric"] and (("traffic_note" in d["roofline"]) or d["roofline"]["traffic"] is None or "traffic_gbps" in d["roofline"])
    assert d["config"]["input_ready"] is True and d["parity_sample"]["frames"] >= 5 * 64 and d["parity_sample"]["differ"] == 0, d.get("parity_sample")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert p.returncode == 3 and not p.stdout.strip()           # more ranks than devices: refused, no line
