"""GPU: bench.py's contract line and its BASELINE configs[3] mode, at toy sizes (the driver runs the real thing)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(*args):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return the_line(r.stdout)


def the_line(stdout):
    """THE line is the last line of stdout and the only one that starts with `{`; it must fit the driver's reader (bench.LINE_LIMIT)"""
    import bench
    lines = stdout.strip().splitlines()
    assert lines[-1].startswith("{") and sum(ln.startswith("{") for ln in lines) == 1
    assert len(lines[-1]) < bench.LINE_LIMIT, len(lines[-1])
    return json.loads(lines[-1])


def sections(stdout):
    return {ln.split(" ", 2)[1]: json.loads(ln.split(" ", 2)[2]) for ln in stdout.strip().splitlines() if ln.startswith("SECTION ")}


def test_mixed_sf_config_one_rank_over_rccl(gpu):
    """BASELINE configs[3] on one GPU: 16384 channels bucketed by SF, one launch per bucket on its own stream, symbols
    gathered through torch.distributed's nccl backend (= RCCL) and checked against the sent ones"""
    d = run_bench("--config", "mixed", "--steps", "3", "--warmup", "1")
    m = d["mixed"]
    assert m["channels"] == 16384 and m["symbols_checked"] == 16384 * 16
    assert m["symbol_errors_vs_sent"] == 0
    assert "nccl" in m["gather_backend"]
    assert 0 < m["frac_byte_weighted"] < 1 and d["value"] == m["Msym_s"]


def test_single_shape_line_keeps_the_contract(gpu):
    d = run_bench("--sf", "8", "--channels", "256", "--symbols", "16", "--steps", "3", "--warmup", "1", "--ramp-seconds", "0", "--cpu-seconds", "0.3")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "oracle"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["symbol_error_rate_vs_sent"] == 0.0
    assert d["oracle"]["windows"] == 256 * 16 and d["oracle"]["index_mismatches"] == 0
    assert d["roofline"]["bound"] == "hbm" and d["cpu_baseline"]["kind"] in ("reference", "port")
    m = run_bench("--sf", "8", "--channels", "256", "--symbols", "16", "--steps", "3", "--warmup", "1", "--ramp-seconds", "0", "--no-cpu-baseline", "--moving")
    assert m["oracle"]["index_mismatches"] == 0 and m["config"]["moving_fine_index"] is True


def test_gpus_n_without_a_launcher_starts_n_ranks(gpu):
    """`python bench.py --gpus 2` with WORLD_SIZE unset re-launches itself under torch.distributed.run (it used to run one rank and
    report n_gpus: 1). The box has one GPU: both ranks share it over gloo (the test hooks of tools/gpu_session.sh "multi")."""
    env = dict(os.environ, LORA_BENCH_BACKEND="gloo", LORA_BENCH_ONE_DEVICE="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--sf", "7", "--channels", "256", "--symbols", "16", "--steps", "3",
                        "--warmup", "1", "--ramp-seconds", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = the_line(r.stdout)
    assert d["n_gpus"] == 2 and "2 rank(s)" in d["config"]["parallelism"]
    assert "re-launching" in r.stderr


def test_two_rank_line_carries_cpu_baseline_and_oracle(gpu):
    """A --gpus N line is graded like the N = 1 line: rank 0 times the CPU reference and checks its batch against the oracle after the
    timed region (the other rank waits on the host), every rank checks a sample of its own shard. One GPU here: both ranks share it."""
    env = dict(os.environ, LORA_BENCH_BACKEND="gloo", LORA_BENCH_ONE_DEVICE="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--sf", "7", "--channels", "256", "--symbols", "16", "--steps", "3",
                        "--warmup", "1", "--ramp-seconds", "0", "--cpu-seconds", "0.3"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = the_line(r.stdout)
    assert d["n_gpus"] == 2
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["value"] > 0
    assert d["oracle"]["windows"] == 256 * 16 and d["oracle"]["index_mismatches"] == 0
    assert d["oracle"]["every_rank_sample"] == dict(d["oracle"]["every_rank_sample"], ranks=2, windows=2 * 256 * 16, index_mismatches=0)
    assert d["symbol_error_rate_vs_sent"] == 0.0


def test_gpus_n_that_contradicts_the_launcher_is_refused(gpu):
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--sf", "7", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr)


def test_default_line_fits_the_reader_one_rank_and_two(gpu):
    """The DEFAULT run (headline + every sweep section), at few steps: THE line stays under bench.LINE_LIMIT (round 5's 20.8 KB line was
    lost by the driver's reader), carries the contract keys with roofline / cpu_baseline / oracle and a row per SF for every sweep; the
    full sections are the earlier SECTION lines and gpurun_out/bench_sections.json. Then the same over two ranks sharing the GPU (gloo)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    for n, extra in ((1, {}), (2, {"LORA_BENCH_BACKEND": "gloo", "LORA_BENCH_ONE_DEVICE": "1"})):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "2", "--cpu-seconds", "1"],
                           capture_output=True, text=True, timeout=900, env=dict(env, **extra), cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        d = the_line(r.stdout)
        assert d["n_gpus"] == n and d["roofline"]["frac"] > 0.3 and d["cpu_baseline"]["value"] > 0 and d["oracle"]["index_mismatches"] == 0
        l3 = [dict(zip(d["level3"]["columns"], r_)) for r_ in d["level3"]["rows"]]        # (columnar: the keys once)
        for k, rows_ in (("per_sf", d["per_sf"]), ("moving", d["moving"]), ("level3", l3)):
            assert [e["sf"] for e in rows_] == list(range(7, 13)), k
        assert all(e["index_mismatches"] == 0 for e in d["per_sf"] + d["moving"])
        assert all(e["oracle_channel_mismatches"] == 0 and e["trace_call_mismatches"] == 0 for e in l3)
        assert "errors" not in d["level3"], d["level3"]["errors"]
        assert d["config5"]["gpu_vs_cpu_index_mismatches"] == 0 and d["mixed"]["oracle"]["index_mismatches"] == 0
        assert d["mixed_level3"]["oracle_channel_mismatches"] == 0
        full = sections(r.stdout)
        assert set(full) >= {"config", "roofline", "cpu_baseline", "per_sf", "moving", "level3", "config5", "mixed", "mixed_level3"}
        assert full["level3"][0]["running"]["chunk8"]["frac"] == l3[0]["chunk8_frac"]
        saved = json.load(open(os.path.join(ROOT, "gpurun_out", "bench_sections.json")))
        assert saved["level3"] == full["level3"] and saved["value"] == d["value"]
        if n == 2:
            assert d["oracle"]["every_rank_sample"]["ranks"] == 2 and d["cpu_baseline"]["kind"] in ("reference", "port")
