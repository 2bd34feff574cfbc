"""CPU: the contract of bench.py's LAST stdout line (the one the driver parses): built from a stored full record, it must be one JSON
object under 4 KB carrying the contract keys, the stage / whole-step / LK rooflines and the CPU baseline as plain numbers."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_compact_line_from_a_stored_record():
    sys.path.insert(0, ROOT)
    from benchlib.report import compact_line
    detail = json.load(open(os.path.join(ROOT, "profiles", "r04e_bench_detail.json")))
    line = compact_line(detail)
    assert "\n" not in line and len(line) < 4096, len(line)
    c = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in c, k
    assert c["config"]["workload"] and c["config"]["streams_per_gpu"] == 128 and "model" not in c["config"]
    r = c["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > r["algorithmic_bytes_per_launch"]
    assert 0 < r["frame"]["frac"] < 1 and 0 < r["lk"]["frac"] < 1
    b = c["cpu_baseline"]
    assert b["kind"] == "port" and b["cores"] >= 1 and b["value"] > 0 and isinstance(b["sample"], str)
    assert c["ba"]["ms_per_iter"] > 0 and set(c["ba"]["windows_ms_per_iter"]) >= {"P5_free_20_const", "P20", "P50", "P100"}
    assert c["single_stream"]["live"] > 0 and c["tolerance_mode"]["value"] > c["value"] * 0.5
    assert c["parity_vs_oracle"]["ok"] is True
    # a record of any size still yields a parseable line: optional objects are dropped before the hard limit
    fat = dict(detail); fat["configs"] = {f"cfg{i}": {"value": float(i)} for i in range(2000)}
    assert len(compact_line(fat)) <= 8000
