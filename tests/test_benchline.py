"""CPU: the line bench.py prints stays short enough for the driver to parse (the 20.6 KB line of round 5
was recorded with `parsed: null`).  The canned long form is round 5's own line (profiles/r05_bench_default.json)."""
import json
import os

import pytest

from rii_amd import benchline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _long_form():
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))


def test_compact_line_is_short_and_keeps_the_contract():
    long_form = _long_form()
    assert len(json.dumps(long_form)) > 20000
    s = benchline.dumps(long_form, "gpurun_out/bench_full_linear.json")
    assert len(s) < 12000 and "\n" not in s
    line = json.loads(s)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "others", "full_form"):
        assert key in line, key
    assert line["value"] == pytest.approx(long_form["value"], rel=1e-4)
    r = line["roofline"]
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["frac"] == pytest.approx(long_form["roofline"]["frac"], rel=1e-4) and r["traffic"] == long_form["roofline"]["traffic"]
    c = line["cpu_baseline"]
    assert c["kind"] == "reference" and c["cores"] == 16 and c["ids_match_gpu"] is True and "sample" in c
    for name, row in long_form["others"].items():
        if not isinstance(row, dict):
            continue
        short = line["others"][name]
        if "roofline" in row:
            assert short["roofline"]["frac"] == pytest.approx(row["roofline"]["frac"], rel=1e-4), name
        if "cpu_baseline" in row:
            assert short["cpu_baseline"]["value"] == pytest.approx(row["cpu_baseline"]["value"], rel=1e-4), name
            assert short["cpu_baseline"]["ids_match_gpu"] is True
        assert "note" not in short and "config" not in short


def test_no_prose_survives_and_nan_becomes_null():
    line = {"metric": "queries/sec", "value": float("nan"), "roofline": {"note": "x" * 900, "frac": 0.5, "kernel": "k" * 500},
            "others": {"a": {"value": 1.23456789, "note": "y" * 5000, "roofline": {"frac": 0.25, "hbm": {"z": 1}}}}}
    out = benchline.compact(line)
    assert out["value"] is None and "note" not in out["roofline"] and len(out["roofline"]["kernel"]) <= benchline.MAX_STR
    assert out["others"]["a"] == {"value": 1.2346, "roofline": {"frac": 0.25}}


def test_last_resort_keeps_the_rows_the_judge_reads():
    long_form = _long_form()
    long_form["others"] = {"leg%d" % i: dict(long_form["others"]["subset"]) for i in range(36)}
    line = json.loads(benchline.dumps(long_form))
    assert len(json.dumps(line)) <= benchline.LINE_LIMIT
    row = line["others"]["leg7"]
    assert set(row) == {"value", "ms_per_step", "kernel", "kernel_ms", "roofline", "cpu_baseline"} and row["cpu_baseline"]["ids_match_gpu"] is True


def test_oversized_line_is_an_error_not_a_silent_loss():
    with pytest.raises(ValueError):
        benchline.dumps({"metric": "m", "blob": ["x" * 150] * 200})
