"""bench.py's ONE line must stay small enough for the driver's reader (round 5's 20.8 KB line came back `parsed: null`): the compact
form of a full result -- here the two full lines round 5 committed -- keeps the contract keys and one short row per SF, under
bench.LINE_LIMIT bytes. (The GPU tests hold the lines of real runs to the same limit: tests/test_gpu_bench.py.)"""
import json
import os

import pytest

from conftest import ROOT

import bench

FULL = [os.path.join(ROOT, "profiles", "r05", n) for n in ("s40_bench_default.json", "s42_bench_two_ranks_one_device_gloo.json")]
def level3_rows(line):
    """the level3 section of THE line is columnar (the keys once): back to one dict per SF"""
    return [dict(zip(line["level3"]["columns"], r)) for r in line["level3"]["rows"]]


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline", "oracle")


@pytest.mark.parametrize("path", FULL)
def test_compact_line_of_a_full_run_fits_the_reader(path):
    full = json.loads(open(path).read().strip().splitlines()[-1])
    line = bench.compact_line(full)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < bench.LINE_LIMIT <= 8000, len(text)
    for k in CONTRACT:
        assert line[k] == full[k] or k in ("config", "roofline", "cpu_baseline"), k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert line["roofline"][k] == full["roofline"][k]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert line["cpu_baseline"][k] == full["cpu_baseline"][k]
    assert line["config"]["workload"] == full["config"]["workload"]
    # one row per SF in every sweep, with the oracle's verdict in it
    l3 = level3_rows(line)
    assert [e["sf"] for e in line["per_sf"]] == [e["sf"] for e in line["moving"]] == [e["sf"] for e in l3] == list(range(7, 13))
    assert all(e["index_mismatches"] == 0 for e in line["per_sf"] + line["moving"])
    for row, e in zip(l3, full["level3"]):
        assert row["frac_kernel_median"] == e["frac_kernel_median"] and row["oracle_channel_mismatches"] == e["oracle_channel_mismatches"] == 0
        assert row["trace_call_mismatches"] == 0 and row["near_step"] == e["near_step"]
        assert row["running_frac"] == e["running"]["frac"] and row["chunk8_frac"] == e["running"]["chunk8"]["frac"]
        assert row["chunk8_pipelined_Msym_s"] == e["running"]["chunk8"]["pipelined"]["Msym_s"]
    assert line["config5"] == full["config5"] and line["mixed"]["frac_byte_weighted"] == full["mixed"]["frac_byte_weighted"]
    assert line["mixed_level3"]["oracle_channel_mismatches"] == 0
    assert "SECTION" in line["sections"]


def test_compact_line_never_exceeds_the_limit():
    """a result that would not fit even in compact form (here: level3 rows blown up) loses its widest extras, never the contract keys"""
    full = json.loads(open(FULL[0]).read().strip().splitlines()[-1])
    full["level3"] = full["level3"] * 8
    full["level3_scaling"] = full["level3_scaling"] * 8
    line = bench.compact_line(full)
    assert len(json.dumps(line, separators=(",", ":"))) < bench.LINE_LIMIT
    for k in CONTRACT:
        assert k in line
    assert isinstance(line["level3_scaling"], str) and "SECTION" in line["level3_scaling"]


def test_single_shape_line_is_unchanged_but_for_the_defaults():
    full = {"metric": "m", "value": 1.0, "unit": "Msym/s", "n_gpus": 1, "steps": 3, "warmup": 1, "ms_per_step": 1.0, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": "w", "moving_fine_index": True, "alias_windows": False},
            "roofline": {"bound": "hbm", "frac": 0.5}, "oracle": {"windows": 4, "index_mismatches": 0}}
    line = bench.compact_line(full)
    assert line["config"] == {"workload": "w", "moving_fine_index": True} and line["oracle"] == full["oracle"] and line["roofline"] == full["roofline"]
