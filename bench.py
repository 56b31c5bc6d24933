#!/usr/bin/env python
"""Benchmark of the MI355X LoRa demod hot path: Msymbols/s demodulated (dechirp + FFT + argmax), per SF.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (one lorahip_detect_batch launch) over one batch of synthetic IQ already resident
in HBM: `channels` channels x `symbols` symbol windows of 2^SF cf32 samples. The headline workload (value / config /
roofline) is BASELINE.json configs[1]: 4096 channels SF7 (N=128 FFT) x 256 windows per channel = 1 GiB of IQ per step.
With N > 1 every rank demodulates its own channels (independent units, no data-path collective): weak scaling; every
value is the whole-job aggregate. The CPU legs (cpu_baseline, the oracle checks) run on rank 0, on its own shard, after the timed
regions, while the other ranks wait at a host-side (gloo) barrier; every rank also holds a sample of its shard to the oracle.

Rank 0 prints ONE JSON line -- the COMPACT form of the result (compact_line: under LINE_LIMIT = 8000 bytes; round 5's 20.8 KB line was lost by the
driver's reader): the contract keys, roofline, cpu_baseline and oracle as they are, one short row per SF of every sweep (level3 by columns). The FULL
sections go out first, one `SECTION <name> {...}` line each, and into gpurun_out/bench_sections.json. Besides the contract fields the result carries
  roofline      HBM roofline of the detect kernel: algorithmic bytes/launch (8*2^SF+14 per window) /
                average launch duration measured with HIP events on the launch stream inside the C ABI
  cpu_baseline  the reference CPU path (oracle/_ref: the real LoRaDemod.cpp + kissfft compiled in place) timed on this
                box's host cores on a bounded sample of the same IQ, at the three flag sets BASELINE.md names
  per_sf        the metric is "per SF": the same measurement at SF7..12 (SF12 = BASELINE configs[2], 1024 channels), each
                with its roofline fraction, a CPU baseline, and ALL windows of the batch compared with the CPU oracle
  moving        the locked-receiver batch shape (every window its own fine-tune error and start index, LoRaDemod.cpp:160-162)
  level3        whole LoRaDemod blocks (frame sync, frequency estimate, packets) through the streaming kernel, end to end; `running`: the same
                capture arriving in chunks of 128 / 8 windows through lorahip_demod_receive -- sequential, pipelined (async = 2), with the block's
                signals delivered, and RESIDENT (async = 3: one kernel launch across the steps)
  config5       BASELINE configs[4]: SF10, 8192 channels x 64 windows at -10 dB SNR, symbol error rate GPU and CPU
  mixed         BASELINE configs[3]: 16384 channels, SF = 7 + c mod 6, byte-weighted shards, symbols gathered over RCCL
Single-shape runs for profiling: --sf S [--moving] [--alias-windows] [--fine-gather]; --config mixed runs configs[3] alone.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
METRIC = "Msymbols/sec demodulated (dechirp+FFT+argmax)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ramp-seconds", type=float, default=0.3,
                    help="untimed launches before the warm-up steps until the GPU has left its idle clocks "
                         "(a cold MI355X needs ~40 ms of load to ramp: tools/ramp.py, profiles/r01/s4_clock_ramp.txt)")
    ap.add_argument("--sf", type=int, default=None, help="single-shape run at this SF (no sweep); default: SF7 headline + SF7..12 sweep")
    ap.add_argument("--channels", type=int, default=None, help="channels per GPU (default: 1 GiB of IQ per step)")
    ap.add_argument("--symbols", type=int, default=None, help="symbol windows per channel per step")
    ap.add_argument("--noise-sigma", type=float, default=0.5, help="AWGN per I/Q component (signal amplitude 1)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="headline line only (no per_sf / moving / level3 / config5 / mixed)")
    ap.add_argument("--cpu-seconds", type=float, default=6.0)
    ap.add_argument("--config", choices=["default", "mixed"], default="default",
                    help="mixed: BASELINE configs[3] alone (16384 channels, SF = 7 + c mod 6, sharded over the ranks)")
    ap.add_argument("--moving", action="store_true",
                    help="single-shape diagnostic: per-window fine-tune error and start index (the DATASYMBOLS shape of a locked "
                         "receiver) instead of the launch-uniform steady state")
    ap.add_argument("--alias-windows", action="store_true",
                    help="single-shape diagnostic: every window reads window 0 (no HBM traffic): the compute-only time")
    ap.add_argument("--fine-gather", action="store_true",
                    help="A/B: read the fine-tune table in HBM (round-1 path) instead of the split tables (lorahip_set_fine_gather)")
    ap.add_argument("--traffic", type=float, default=None,
                    help="HBM bytes per launch from a separate rocprofv3 --pmc pass (profiles/), if known")
    return ap.parse_args()


LINE_LIMIT = 8000        # bytes of THE line. The driver keeps an ~8 KB tail of stdout and its reader lost round 5's 20.8 KB line
SECTIONS_FILE = os.path.join("gpurun_out", "bench_sections.json")


def _get(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(full):
    """THE line from the full result: the contract keys, roofline, cpu_baseline and oracle as they are; of every sweep section one short
    row per SF (rates, roofline fractions, oracle mismatch counts, the near-boundary counters). Everything measured stays available:
    the full sections are printed as earlier `SECTION <name> {...}` lines and written to gpurun_out/bench_sections.json."""
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                "dtype", "data") if k in full}
    cfg = dict(full.get("config", {}))
    for k, dflt in (("kernel_variant", 0), ("alias_windows", False), ("moving_fine_index", False), ("fine_gather", False)):
        if cfg.get(k) == dflt:
            cfg.pop(k)                                   # (diagnostic switches: named only when they are on)
    out["config"] = cfg
    for k in ("symbol_error_rate_vs_sent", "bin_offset", "rccl_ranks"):
        if k in full:
            out[k] = full[k]
    if "roofline" in full:
        rf = dict(full["roofline"])
        src = rf.pop("traffic_source", None)
        if src is not None:
            rf["traffic_source"] = _pick(src, "file", "session", "matches_this_tree", "error")
        if rf.get("counters_source") is None:
            rf.pop("counters_source", None)
        out["roofline"] = rf
    if "cpu_baseline" in full:
        cb = dict(full["cpu_baseline"])
        cb.pop("affinity", None)
        if "other_flags" in cb:
            cb["other_flags"] = [_pick(x, "flags", "value") for x in cb["other_flags"]]
        out["cpu_baseline"] = cb
    if "oracle" in full:
        out["oracle"] = full["oracle"]
    pc = full.get("pcie_inclusive")
    if pc:
        out["pcie_inclusive"] = {"Msym_s": pc.get("Msym_s"), "GB_s": pc.get("GB_s"), "pinned_GB_s": _get(pc, "pinned", "GB_s"),
                                 "same_symbols": bool(pc.get("same_symbols")) and bool(_get(pc, "pinned", "same_symbols", default=True)),
                                 "level1_shim_us_per_detect": _get(pc, "level1_shim", "us_per_detect")}
    if "per_sf" in full:
        out["per_sf"] = [dict(_pick(e, "sf", "channels", "Msym_s", "frac", "valu_busy", "error"), index_mismatches=_get(e, "oracle", "index_mismatches"),
                              cpu_Msym_s=_get(e, "cpu_baseline", "value")) for e in full["per_sf"]]
    if "moving" in full:
        out["moving"] = [dict(_pick(e, "sf", "Msym_s", "frac", "valu_busy", "error"), index_mismatches=_get(e, "oracle", "index_mismatches")) for e in full["moving"]]
    if "level3" in full:
        # one row per SF, by columns (the keys once, not six times): rates, roofline fractions, the oracle's verdict, the near-boundary
        # counters; the receiver steps at 128- and 8-window chunks -- sequential, pipelined, with the block's signals kept, resident
        cols = (("sf", ("sf",)), ("channels", ("channels",)), ("lanes_log2", ("lanes_log2",)), ("frac_kernel", ("frac_kernel",)),
                ("frac_kernel_median", ("frac_kernel_median",)), ("with_signals_frac_kernel_median", ("with_signals", "frac_kernel_median")),
                ("frac_e2e", ("frac_e2e",)), ("near_squelch", ("near_squelch",)), ("near_step", ("near_step",)),
                ("oracle_channel_mismatches", ("oracle_channel_mismatches",)), ("trace_call_mismatches", ("trace_call_mismatches",)),
                ("running_Msym_s", ("running", "Msym_s")), ("running_frac", ("running", "frac")), ("running_pipelined_frac", ("running", "pipelined", "frac")),
                ("running_with_signals_frac", ("running", "with_signals", "frac")), ("running_resident_frac", ("running", "resident", "frac")),
                ("same_packets_as_one_shot", ("running", "same_packets_as_one_shot")),
                ("chunk8_Msym_s", ("running", "chunk8", "Msym_s")), ("chunk8_frac", ("running", "chunk8", "frac")),
                ("chunk8_pipelined_Msym_s", ("running", "chunk8", "pipelined", "Msym_s")), ("chunk8_pipelined_frac", ("running", "chunk8", "pipelined", "frac")),
                ("chunk8_with_signals_frac", ("running", "chunk8", "with_signals", "frac")),
                ("chunk8_with_signals_pipelined_frac", ("running", "chunk8", "with_signals", "pipelined", "frac")),
                ("chunk8_resident_Msym_s", ("running", "chunk8", "resident", "Msym_s")), ("chunk8_resident_frac", ("running", "chunk8", "resident", "frac")),
                ("chunk8_resident_depth3_Msym_s", ("running", "chunk8", "resident_depth3", "Msym_s")), ("chunk8_resident_depth3_frac", ("running", "chunk8", "resident_depth3", "frac")),
                ("chunk8_resident_kernel", ("running", "chunk8", "resident", "kernel_resident")),
                ("chunk8_resident_same_packets", ("running", "chunk8", "resident", "same_packets_as_one_shot")),
                ("pothos_ports_off", ("pothos_block", "ports_off", "Msym_s")), ("pothos_pinned_input_slabs", ("pothos_block", "ports_off_pinned_input_slabs", "Msym_s")),
                ("pothos_ports_on", ("pothos_block", "ports_on", "Msym_s")), ("pothos_vs_cpu_same_threads", ("pothos_block", "vs_cpu_same_threads")))
        l3 = {"columns": [c for c, _ in cols], "rows": [[_get(e, *path) for _, path in cols] for e in full["level3"]]}
        errs = {}
        for e in full["level3"]:
            for where, v in (("", e.get("error")), ("parity", e.get("parity_error")), ("running", _get(e, "running", "error")),
                             ("resident128", _get(e, "running", "resident", "error")), ("resident", _get(e, "running", "chunk8", "resident", "error")),
                             ("resident_depth3", _get(e, "running", "chunk8", "resident_depth3", "error")),
                             ("pothos_block", _get(e, "pothos_block", "error"))):
                if v:
                    errs["sf%s %s" % (e.get("sf"), where)] = str(v)[:100]
            if e.get("oracle_kind", "reference") != "reference":
                errs["sf%s oracle_kind" % e.get("sf")] = e["oracle_kind"]     # (named only when the pinned restatement stood in for oracle/_ref)
        if errs:
            l3["errors"] = errs
        out["level3"] = l3
    if "config5" in full:
        out["config5"] = full["config5"]
    if "mixed" in full:
        m = dict(full["mixed"])
        for k in ("scheduler", "iq_bytes_per_step", "my_channels_rank0", "sf_rule"):
            m.pop(k, None)
        out["mixed"] = m
    if "mixed_level3" in full:
        m = dict(full["mixed_level3"])
        for k in ("object", "sf_rule", "oracle_kind"):
            m.pop(k, None)
        out["mixed_level3"] = m
    sc = full.get("level3_scaling")
    if isinstance(sc, list):
        out["level3_scaling"] = {"columns": ["sf", "channels", "lanes_log2", "frac", "frac_median"],
                                 "rows": [[e.get("sf"), e.get("channels"), e.get("lanes_log2"), e.get("default_frac"), e.get("default_frac_median")] for e in sc]}
    elif sc is not None:
        out["level3_scaling"] = sc
    if any(k in full for k in SECTION_KEYS):
        out["sections"] = "full sections: the earlier stdout lines `SECTION <name> {...}` and %s" % SECTIONS_FILE
    n = len(json.dumps(out, separators=(",", ":")))
    if n > LINE_LIMIT:                                  # never print a line the reader may lose: drop the widest extras first
        for k in ("level3_scaling", "pcie_inclusive", "moving", "per_sf", "level3"):
            if k in out and n > LINE_LIMIT:
                out[k] = "dropped from THE line (%d bytes over): see the SECTION lines" % (n - LINE_LIMIT)
                n = len(json.dumps(out, separators=(",", ":")))
    return out


SECTION_KEYS = ("config", "roofline", "cpu_baseline", "pcie_inclusive", "per_sf", "moving", "level3", "config5", "mixed", "mixed_level3", "level3_scaling")


def emit(env, line):
    """Rank 0 prints THE line -- after the process group is gone and every C-level stdout buffer is flushed (RCCL prints a version
    banner through C stdio, which would otherwise land after the line when the process exits). The line is the compact form
    (compact_line, under LINE_LIMIT bytes); the full sections go out first, one `SECTION <name> {...}` line each, and into
    gpurun_out/bench_sections.json."""
    import ctypes
    env.close()
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if line is not None:
        for k in SECTION_KEYS:
            if k in line:
                print("SECTION %s %s" % (k, json.dumps(line[k], separators=(",", ":"))), flush=True)
        if "per_sf" in line or "mixed" in line:           # (the default run and --config mixed: single-shape runs have one section)
            try:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(ROOT, SECTIONS_FILE), "w") as f:
                    json.dump(line, f, indent=1)
            except OSError:                               # scratch only: a read-only tree must not cost the line
                pass
        print(json.dumps(compact_line(line), separators=(",", ":")), flush=True)


def r4(x):
    """4 significant digits: keeps the one JSON line short"""
    return float("%.4g" % x)


class Env:
    """torch / distributed plumbing shared by the sections"""

    def __init__(self, a):
        import torch
        self.torch = torch
        self.a = a
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != a.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: the line would not describe the run" % (a.gpus, self.world))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
        # test hooks (tools/gpu_r02.sh "multi"): several ranks on ONE GPU over gloo exercise the N > 1 code path where only a
        # single device exists; a real multi-GPU run uses neither
        self.backend = os.environ.get("LORA_BENCH_BACKEND", "nccl")
        if os.environ.get("LORA_BENCH_ONE_DEVICE"):
            local = 0
        self.local = local
        torch.cuda.set_device(local)
        self.dev = torch.device("cuda", local)
        self.dist = None
        self.rccl_ranks = None
        self.host_group = None
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(self.backend)
            # prove the collective backend is up on device tensors before anything is timed
            t = torch.ones(1, device=self.dev if self.backend == "nccl" else "cpu")
            dist.all_reduce(t)
            self.rccl_ranks = int(t.item()) if self.backend == "nccl" else None
            # a HOST-side group for the waits around rank 0's CPU legs (cpu_baseline, the oracle checks): a gloo barrier blocks on a
            # socket, an RCCL barrier would keep seven host threads and seven GPUs spinning beside the CPU code being timed
            self.host_group = dist.new_group(backend="gloo") if self.backend == "nccl" else None

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def host_barrier(self):
        """the ranks meet on the host (no device work in flight is waited for, nothing spins): where rank 0 runs its CPU legs"""
        if self.dist is not None:
            self.dist.barrier(group=self.host_group)

    def max_over_ranks(self, *vals):
        if self.dist is None:
            return vals
        t = self.torch.tensor(list(vals), dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return tuple(float(v) for v in t)

    def sum_over_ranks(self, *vals):
        if self.dist is None:
            return vals
        t = self.torch.tensor(list(vals), dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return tuple(float(v) for v in t)

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


class Shape:
    """one batch of B channels x S windows at one SF, resident in HBM, with its outputs"""

    def __init__(self, env, L, sf, B, S, noise_sigma, variant=0):
        torch = env.torch
        self.env, self.L, self.sf, self.N, self.B, self.S, self.W = env, L, sf, 1 << sf, B, S, B * S
        self.ctx = L.Context(sf, device=env.local)
        self.ctx.set_variant(variant)
        self.ctx.use_torch_stream()
        g = self.g = torch.Generator(device=env.dev)
        g.manual_seed(0x10AA + env.rank + 977 * sf)
        self.sym = torch.randint(0, self.N, (self.W,), generator=g, device=env.dev, dtype=torch.int32).to(torch.int16)
        self.iq = self.ctx.synth_symbols(self.sym, ampl=1.0, noise_sigma=noise_sigma, seed=0x5EED0000 + env.rank + 131 * sf)
        self.out = self.new_out()
        self.fine_err = self.fine_idx0 = None

    def new_out(self):
        torch, W, dev = self.env.torch, self.W, self.env.dev
        return dict(sym=torch.empty(W, dtype=torch.int16, device=dev), power=torch.empty(W, dtype=torch.float32, device=dev),
                    powerAvg=torch.empty(W, dtype=torch.float32, device=dev), fIndex=torch.empty(W, dtype=torch.float32, device=dev))

    def moving_inputs(self):
        """per-window _finefreqError in [-2, 2) bins and a random start index: the DATASYMBOLS shape"""
        torch = self.env.torch
        if self.fine_err is None:
            self.fine_err = (torch.rand(self.W, generator=self.g, device=self.env.dev) * 4.0 - 2.0).to(torch.float32)
            self.fine_idx0 = torch.randint(0, 128 * self.N, (self.W,), generator=self.g, device=self.env.dev, dtype=torch.int32)
        return self.fine_err, self.fine_idx0

    def measure(self, steps, warmup, ramp_seconds, moving=False, alias=False):
        """ramp, W warm-up launches, then exactly K timed launches between barriers; returns (elapsed_s, kernel_ms) = max over ranks"""
        env, torch, ctx = self.env, self.env.torch, self.ctx
        out = self.new_out() if moving else self.out
        offsets = torch.zeros(self.W, dtype=torch.int64, device=env.dev) if alias else None
        fe = fi = None
        if moving:
            fe, fi = self.moving_inputs()
            self.out_moving = out
        batch = ctx.make_batch(self.iq, self.W, out["sym"], out["power"], out["powerAvg"], out["fIndex"], chirp_sel_all=self.L.CHIRP_UP,
                               offsets=offsets, fine_err=fe, fine_idx0=fi)
        # clock ramp: the first ~40 ms of load after idle run at reduced clocks (profiles/r01/s4_clock_ramp.txt)
        t_ramp = time.perf_counter()
        while time.perf_counter() - t_ramp < ramp_seconds:
            for _ in range(10):
                ctx.detect_batch_raw(batch)
            torch.cuda.synchronize()
        for _ in range(warmup):
            ctx.detect_batch_raw(batch)
        env.barrier()
        t0 = time.perf_counter()
        ctx.timer_start()
        for _ in range(steps):
            ctx.detect_batch_raw(batch)
        kernel_ms = ctx.timer_stop()          # HIP events on the launch stream, around exactly the K launches
        env.barrier()
        elapsed = time.perf_counter() - t0
        return env.max_over_ranks(elapsed, kernel_ms)

    def ser_vs_sent(self, out=None):
        """genChirp's phase ramp is one sample ahead of the demod's table: a window-aligned symbol s lands in
        bin s+1; the frame sync of the real receiver removes that constant. Returns (symbol error rate, the constant)."""
        torch = self.env.torch
        got = (out or self.out)["sym"].to(torch.int32) & 0xffff
        sent = self.sym.to(torch.int32) & 0xffff
        diff = (got - sent) % self.N
        off = int(torch.mode(diff).values)
        return float((diff != off).float().mean()), off

    def host_iq(self):
        if not hasattr(self, "_host"):
            self._host = self.iq.cpu().numpy()
        return self._host

    def oracle_check(self, threads, moving=False, limit=None):
        """ALL windows of the batch (or the first `limit`) through the CPU oracle (oracle/lora_oracle.c, pinned to the reference):
        indices must be identical; power / fIndex differences are reported"""
        import numpy as np
        from oracle.oracle import Oracle
        out = self.out_moving if moving else self.out
        k = self.W if limit is None else min(self.W, int(limit))
        kw = {}
        if moving:
            kw = dict(fine_err=self.fine_err[:k].cpu().numpy(), fine_idx0=self.fine_idx0[:k].cpu().numpy())
        iq = self.host_iq() if k == self.W else self.iq.reshape(-1)[:k * self.N].cpu().numpy()
        o = Oracle().detect_batch(self.sf, iq, nthreads=threads, **kw)
        gs = out["sym"][:k].cpu().numpy().view(np.uint16)
        fin = np.isfinite(o["power"])
        return {"windows": int(k), "index_mismatches": int((o["sym"] != gs).sum()),
                "max_dB": r4(float(np.abs(out["power"][:k].cpu().numpy() - o["power"])[fin].max())),
                "max_fIndex": r4(float(np.abs(out["fIndex"][:k].cpu().numpy() - o["fIndex"]).max()))}

    def oracle_check_every_rank(self, limit=65536, moving=False):
        """several ranks: EVERY rank holds a bounded sample of its own shard to the CPU oracle (its share of the host cores), the
        counts are summed over the ranks -- beside rank 0's check of its whole batch"""
        env = self.env
        thr = max(1, min(32, (os.cpu_count() or 1) // env.world))
        r = self.oracle_check(thr, moving=moving, limit=limit)
        win, bad = env.sum_over_ranks(r["windows"], r["index_mismatches"])
        (db,) = env.max_over_ranks(r["max_dB"])
        return {"ranks": env.world, "windows": int(win), "index_mismatches": int(bad), "max_dB": r4(db), "threads_per_rank": thr}

    def close(self):
        self.ctx.close()


def host_cpu_info():
    cores = os.cpu_count() or 1
    info = {"host_threads": cores}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        info["cgroup_cpu_max"] = "unlimited" if q[0] == "max" else r4(float(q[0]) / float(q[1]))
    except Exception:
        pass
    return info


_best_threads = {}


def cpu_baseline(sf, iq_host, samples_per_stream, n_streams, seconds, flags="-O2", probe=True):
    """Time the reference CPU path (the verbatim LoRaDemod.cpp work() loop: dechirp + kissfft + detect + frame machine) on a
    bounded sample of the same IQ. Returns the JSON object."""
    from oracle.oracle import Oracle, Ref
    cores = os.cpu_count() or 1
    if Ref.available(flags):
        impl, kind = Ref(flags), "reference"
    elif flags == "-O2":
        impl, kind = Oracle(), "port"
    else:
        return None
    cands = sorted({min(cores, n_streams), max(1, cores // 2), max(1, cores // 4), max(1, cores // 8)}, reverse=True)
    if not probe and sf in _best_threads:
        cands = [_best_threads[sf]]
    elif not probe and _best_threads:
        cands = [list(_best_threads.values())[0]]
    best = None
    for threads in cands:
        # the thread count that is fastest on this box (SMT / cgroup quota can make "all hardware threads" slower): ~0.5 s probes
        threads = min(threads, n_streams)
        t0 = time.perf_counter()
        calls = impl.demod_bench(sf, iq_host, samples_per_stream, n_streams, threads, 1)
        dt = time.perf_counter() - t0
        if len(cands) > 1:
            rep = max(1, int(0.5 / max(dt, 1e-4)))
            t0 = time.perf_counter()
            calls = impl.demod_bench(sf, iq_host, samples_per_stream, n_streams, threads, rep)
            dt = (time.perf_counter() - t0) / rep
            calls //= rep
        if best is None or calls / dt > best[0]:
            best = (calls / dt, threads, dt)
    threads = best[1]
    _best_threads.setdefault(sf, threads)
    repeat = max(1, int(seconds / max(best[2], 1e-4)))
    t0 = time.perf_counter()
    calls = impl.demod_bench(sf, iq_host, samples_per_stream, n_streams, threads, repeat)
    dt = time.perf_counter() - t0
    v = calls / dt / 1e6
    return {"value": r4(v), "unit": "Msym/s", "cores": threads, "per_core": r4(v / threads), "kind": kind, "flags": "g++ %s, no FMA" % flags,
            "sample": "%d ch x %d samples of the same SF%d IQ, %d work() calls in %.1f s" % (n_streams, samples_per_stream, sf, calls, dt)}


def roofline_obj(sf, W, launch_s, traffic, L, traffic_src=True, shape="steady", default_geometry=True):
    alg = W * L.bytes_per_symbol(sf)
    ach = alg / launch_s / 1e9
    r = {"bound": "hbm", "achieved": r4(ach), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r4(ach / HBM_PEAK_GBS), "traffic": traffic,
         "traffic_source": traffic_table()[1] if traffic_src else None,
         "kernel": "lorahip detect (dechirp+FFT+detect fused)", "launch_us": r4(launch_s * 1e6), "algorithmic_bytes_per_launch": alg,
         "bytes_per_symbol": L.bytes_per_symbol(sf)}
    # north_star: "LDS/VALU utilisation ... against gfx950 peak" -- SQ counters of this kernel at this geometry from separate rocprofv3 --pmc
    # passes (profiles/counters.json, tools/pmc_counters.py), replayed like `traffic` and only while they were measured on these kernels
    c = counters_for(sf, shape) if default_geometry else None
    r["valu_busy"] = c.get("valu_busy") if c else None
    r["lds_busy"] = c.get("lds_busy") if c else None
    r["instr_per_sample"] = c.get("instr_per_sample") if c else None
    if c:
        r["wait_share"], r["lds_conflict"] = c.get("wait_share"), c.get("lds_conflict")
    r["counters_source"] = _counters_cache.get("src")
    return r


_counters_cache = {}


def counters_for(sf, shape="steady"):
    """profiles/counters.json entry of (shape, SF), or None when the file is missing or was measured on other kernels"""
    if "t" not in _counters_cache:
        t, src = None, {"file": "profiles/counters.json"}
        try:
            from lora_sdr_amd.build import kernel_digest
            t = json.load(open(os.path.join(ROOT, "profiles", "counters.json")))
            src["session"] = t.get("session")
            src["matches_this_tree"] = t.get("sources_sha16") == kernel_digest()
            if not src["matches_this_tree"]:
                t = None
        except Exception as e:
            src["error"] = str(e)[:80]
            t = None
        _counters_cache["t"], _counters_cache["src"] = t, src
    t = _counters_cache["t"]
    try:
        return t[shape][str(sf)] if t else None
    except Exception:
        return None


_traffic_cache = {}


def traffic_table():
    """profiles/traffic.json: HBM bytes per launch from separate rocprofv3 --pmc passes of `bench.py --sf S` (committed). The counters are
    NOT collected in this run: the table is replayed, and only while the kernels it was measured on are the ones being timed --
    its `sources_sha16` (lora_sdr_amd.build.kernel_digest() of the build that was profiled) must equal this tree's digest."""
    if "t" not in _traffic_cache:
        t, src = None, {"file": "profiles/traffic.json", "collected": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, replayed here"}
        try:
            from lora_sdr_amd.build import kernel_digest
            t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            src["sources_sha16"], src["session"] = t.get("sources_sha16"), t.get("session")
            src["this_tree_sha16"] = kernel_digest()          # over the files the detect kernels are compiled from
            src["matches_this_tree"] = src["sources_sha16"] == src["this_tree_sha16"]
            if not src["matches_this_tree"]:
                t = None                                    # measured on other kernels: say so, report null
        except Exception as e:
            src["error"] = str(e)[:80]
            t = None
        _traffic_cache["t"], _traffic_cache["src"] = t, src
    return _traffic_cache["t"], _traffic_cache["src"]


def traffic_for(sf, a):
    """HBM bytes per launch: --traffic if given (a PMC pass of this very command), else the committed table while it is current"""
    if a.traffic is not None:
        return a.traffic
    if a.channels is not None or a.symbols is not None:
        return None
    t, _ = traffic_table()
    try:
        return t["per_sf"][str(sf)].get("total_bytes") if t else None
    except Exception:
        return None


# ------------------------------------------------------------------------------------------------------------------
def packets_vs_reference(gpu_packets, r, B):
    """the device's packets (channel, round, length, symbols back to back) against a run_many() result of the CPU reference over the
    same B streams: per channel the number of packets, their lengths and every symbol. -> (bool array of differing channels, packets
    that did not even fit the reference's arrays)"""
    ch, _rd, ln, sy = gpu_packets
    # the device's packets per channel in time order, laid out like the reference's arrays
    order = np.argsort(ch, kind="stable")
    start = np.concatenate([[0], np.cumsum(ln)])[:-1]
    n_gpu = np.bincount(ch, minlength=B).astype(np.int32)
    first = np.concatenate([[0], np.cumsum(n_gpu)])[:-1]
    g_syms = np.zeros_like(r["pkt_syms"])
    g_lens = np.zeros_like(r["pkt_lens"])
    overflow = 0
    fill = np.zeros(B, np.int64)
    for k, p in enumerate(order.tolist()):
        c = int(ch[p]); j = k - int(first[c]); n = int(ln[p])
        if j >= g_lens.shape[1] or fill[c] + n > g_syms.shape[1]:
            overflow += 1
            continue
        g_lens[c, j] = n
        g_syms[c, fill[c]:fill[c] + n] = sy[start[p]:start[p] + n]
        fill[c] += n
    bad_ch = (n_gpu != r["n_packets"]) | (g_lens != r["pkt_lens"]).any(axis=1) | (g_syms != r["pkt_syms"]).any(axis=1)
    return bad_ch, overflow


def level3_parity(L, sf, iq, host, nsyms, gpu_packets, data, gpu_not_ok, threads):
    """EVERY channel of the level-3 workload through the CPU reference block (oracle/_ref: the verbatim LoRaDemod.cpp, one fresh
    block per channel; the pinned restatement where _ref did not travel) on the same IQ: packets compared symbol for symbol, then --
    from one extra, untimed, traced pass of the streaming kernel -- every work() call's consumption and label kind
    (LoRaDemod.cpp:213-232,245,282,302-320), i.e. the frame machine's path call by call. Also answers why packets_ok < packets:
    the reference's own packets are held to the same "carries the sent symbols" test."""
    from lora_sdr_amd import workloads as WL
    from oracle.oracle import Oracle, Ref
    impl, kind = (Ref(), "reference") if Ref.available() else (Oracle(), "port")
    B, N = host.shape[0], 1 << sf
    t0 = time.perf_counter()
    r = impl.demod_run_many(sf, host, mtu=nsyms, nthreads=threads, calls=True)
    cpu_s = time.perf_counter() - t0
    bad_ch, overflow = packets_vs_reference(gpu_packets, r, B)
    # the reference's packets under the test packets_ok applies to the device's
    ref_pk = []
    for c in range(B):
        at = 0
        for j in range(int(r["n_packets"][c])):
            n = int(r["pkt_lens"][c, j])
            ref_pk.append((c, 0, r["pkt_syms"][c, at:at + n]))
            at += n
    n_ref, ok_ref = WL.check_frame_packets(ref_pk, data, N, nsyms)
    out = {"oracle_kind": kind, "oracle_channels_checked": int(B), "oracle_channel_mismatches": int(bad_ch.sum()) + overflow,
           "oracle_packets": int(n_ref), "packets_not_ok": int(gpu_not_ok), "packets_not_ok_in_reference_too": int(n_ref - ok_ref),
           "oracle_cpu_s": r4(cpu_s)}
    # one traced pass: per call consumed + label kind
    dt = L.LoRaDemod(sf, n_channels=B, device=iq.device.index or 0)
    dt.set_mode(1)
    dt.setMTU(nsyms)
    dt.set_trace(True)
    dt.work(iq)
    thresh = np.float32(-30.0)                                          # LoRaDemod.cpp:72 (default, as in the timed passes)
    call_bad = calls_cmp = ch_bad = 0
    for c in range(B):
        t = dt.trace_array(c)
        n = int(r["n_calls"][c])
        st, co = t["state_before"], t["consumed"]
        cls = np.where(st == 0, np.where(co == 2 * N, 1, np.where(~(t["snr"] < thresh), 2, 0)),
                       np.where(st == 1, 3, np.where(st == 2, 0, np.where(st == 3, 4, 5)))).astype(np.uint8)
        if t.size != n:
            ch_bad += 1
            call_bad += abs(int(t.size) - n)
            continue
        d_ = int(((co != r["consumed"][c, :n]) | (cls != r["cls"][c, :n])).sum())
        call_bad += d_
        ch_bad += int(d_ != 0)
        calls_cmp += n
    out.update({"trace_calls_compared": int(calls_cmp), "trace_call_mismatches": int(call_bad), "trace_channel_mismatches": int(ch_bad),
                "trace_near_squelch": dt.near_threshold()[0], "trace_near_step": dt.near_threshold()[1]})
    dt.close()
    return out


def pothos_block(sf, host, nsyms, calls_expected, packets_expected, chunk_windows=128):
    """What a Pothos user gets from /lora/lora_demod_batch (lora_sdr_amd/pothos/LoRaDemodBatch.cpp, the product's reference-side binding,
    compiled against the fake Pothos of oracle/stub and linked with liblorahip.so: oracle/_ref/libloradrop.so): the level-3 workload
    handed over as ORDINARY HOST buffers that arrive 128 windows per channel at a time, a scheduler loop in C around work()
    (oracle/dropin_driver.cpp::loradrop_batch_bench), packets and signals posted like the reference block posts them. Debug ports off
    (the default) and on (the reference block's raw / dec / fft outputs, on a subset of the channels: they need 3 staging arrays per
    channel). Beside it the verbatim CPU block on the same samples with as many host threads as the block keeps busy."""
    from oracle.oracle import DropInBatch, Ref
    if not (DropInBatch.available() and Ref.available()):
        return {"error": "oracle/_ref/libloradrop.so did not travel"}
    B, n = host.shape
    chunk = chunk_windows << sf
    cpus_ = os.cpu_count() or 1                     # (lorahip_upload.cpp::uploadThreads' rule)
    up = int(os.environ.get("LORAHIP_UPLOAD_THREADS", "8" if cpus_ >= 32 else ("6" if cpus_ >= 16 else "3")))
    out = {"what": "LoRaDemodBatch.cpp over host buffers, %d windows per channel per arrival" % chunk_windows, "host_threads": 1 + up}
    for name, ports, nch in (("ports_off", False, B), ("ports_on", True, min(B, max(64, (1 << 21) >> sf)))):
        blk = DropInBatch(sf, nch, max_windows=chunk_windows + 2)
        blk.set("setMTU", nsyms)
        if ports:
            blk.set("setDebugPorts", 1)
        sub = host if nch == B else np.ascontiguousarray(host[:nch])
        blk.bench(sub, chunk)                                # first touch: contexts, staging, device buffers (every channel ends idle: the
        best = None                                          # streams close with silence, so the timed passes start from a clean receiver)
        for _ in range(2):
            r = blk.bench(sub, chunk)
            if best is None or r["seconds"] < best["seconds"]:
                best = r
        blk.close()
        calls = calls_expected * nch // B                   # work() calls of the reference block on these channels (equal per channel set)
        out[name] = {"channels": nch, "Msym_s": r4(calls / best["seconds"] / 1e6), "GB_s_in": r4(sub.size * 8 / best["seconds"] / 1e9),
                     "block_work_calls": best["works"], "packets": best["packets"], "signals": best["signals"], "seconds": r4(best["seconds"])}
        if not ports:
            out[name]["packets_expected"] = packets_expected
            # the same with the inputs taken from the block's OWN input buffer managers (getInputBufferManager, the counterpart of
            # LoRaDemod.cpp:346-357): pinned slabs, all ports' slabs in one allocation; every arrival is written into them by the source
            # (not timed: the upstream block's work) and a work() uploads them as one strided DMA (lorahip_demod_run_host_rows)
            try:
                blk = DropInBatch(sf, nch, max_windows=chunk_windows + 2)
                blk.set("setMTU", nsyms)
                blk.use_input_slabs(True)
                blk.bench(sub, chunk)
                best = None
                for _ in range(2):
                    r = blk.bench(sub, chunk)
                    if best is None or r["seconds"] < best["seconds"]:
                        best = r
                active = blk.input_slabs_active()
                blk.close()
                out["ports_off_pinned_input_slabs"] = {"channels": nch, "slabs_from_the_block": bool(active), "Msym_s": r4(calls / best["seconds"] / 1e6),
                                                       "GB_s_in": r4(sub.size * 8 / best["seconds"] / 1e9), "packets": best["packets"], "seconds": r4(best["seconds"])}
            except Exception as e:
                out["ports_off_pinned_input_slabs"] = {"error": repr(e)[:160]}
    # the CPU it replaces: the verbatim block, same samples, 1 thread and as many threads as the GPU block's host side uses
    ref = Ref("-O2")
    nst = min(B, 256)
    sample = np.ascontiguousarray(host[:nst]).reshape(-1)
    for thr in (1, 1 + up):
        t0 = time.perf_counter()
        c_ = ref.demod_bench(sf, sample, n, nst, thr, 1)
        out["cpu_reference_%d_thread%s_Msym_s" % (thr, "" if thr == 1 else "s")] = r4(c_ / (time.perf_counter() - t0) / 1e6)
    out["vs_cpu_same_threads"] = r4(out["ports_off"]["Msym_s"] / max(out["cpu_reference_%d_threads_Msym_s" % (1 + up)], 1e-9))
    return out


def section_level3(env, L, sf, threads=32):
    """B channels of the LoRaDemod block over whole frames through the streaming kernel (tools/bench_demod.py's workload)"""
    import numpy as np
    from lora_sdr_amd import workloads as WL
    torch = env.torch
    B, frames, nsyms = WL.LEVEL3_CHANNELS[sf], 4, 48
    ctx = L.Context(sf, device=env.local)
    iq, data = WL.frame_streams(ctx, B, frames, nsyms, sigma=0.05)
    d = L.LoRaDemod(sf, n_channels=B, device=env.local)
    d.set_mode(1)
    d.setMTU(nsyms)
    # pass 0: what the demodulator delivers (checked below); also allocates its staging buffers
    d.work(iq)
    near = d.near_threshold()
    ps, pn, pc = d.packets_device(clear=False)
    calls = d.work_calls()
    # frac_kernel counts 8*2^SF + 14 per work() call; FRAMESYNC / QUARTERCHIRP calls advance by less than the
    # window they read, so the same samples are counted by several calls. What the channels actually consumed, once:
    unique_bytes = int(d.consumed_all().sum()) * 8 + 14 * calls
    ch_, rd_, ln_, sy_ = d.packets_arrays()
    pk = list(zip(ch_.tolist(), rd_.tolist(), np.split(sy_, np.cumsum(ln_)[:-1]) if ch_.size else []))
    n_dev = int(ps.shape[0])

    def one_pass(to_host=False, together=True):
        d.clear_packets()
        d.activate()
        if together:
            env.barrier()                           # (several ranks: every rank's pass starts together; one rank: a device synchronise)
        else:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        d.work(iq)                                  # the streaming kernel + per-channel state back: packets stay on the device
        t1 = time.perf_counter()
        d.packets_device(clear=False)               # ... packed there into the batched decoder's input layout
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        t3 = t2
        if to_host:
            d.packets_arrays()                      # ... or drained to the host queue (what a host consumer pays)
            t3 = time.perf_counter()
        return t2 - t0, d.kernel_ms(), (t1 - t0) + (t3 - t2)
    # a receiver runs continuously: like the batch sections, time it with the device at its loaded clocks (a cold MI355X takes
    # ~40 ms of work to leave its idle clocks) -- passes back to back, no host drain in between
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.25:
        one_pass(together=False)                    # (the ranks' ramps take different numbers of passes: no barrier inside)
    best, kms_all = None, []
    for _ in range(5):
        r_ = one_pass()
        kms_all.append(r_[1])
        best = r_ if best is None else (min(best[0], r_[0]), min(best[1], r_[1]), r_[2])     # best wall clock, best device time
    best = (best[0], best[1], one_pass(to_host=True)[2])
    kms_all.sort()
    # the same launch with the block's signals kept (error / power / snr once per packet, LoRaDemod.cpp:267-269: the reference block
    # always emits them; one more pair of logarithms per packet and a 16-byte record)
    d.set_signals(True)
    # (the drain to the host just above let the device fall back towards its idle clocks: ramp again as before the first five passes --
    # without it this launch read 12 % slower than the one without signals at SF7, where a pass is 2 ms, and 0-3 % at SF11 / SF12)
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.25:
        one_pass(together=False)
    sig_kms = sorted(one_pass()[1] for _ in range(5))
    sig_pass = (None, sig_kms[0])
    n_signals = len(d.signals()[0])
    d.set_signals(False)
    d.clear_packets()
    # several ranks (bench.py --gpus N): every rank demodulates its own B channels; times are the slowest rank's, counts are sums
    best = env.max_over_ranks(*best)
    k_mean, k_median, k_sig, k_sig_median = env.max_over_ranks(sum(kms_all) / len(kms_all), kms_all[len(kms_all) // 2], sig_pass[1], sig_kms[len(sig_kms) // 2])
    (calls_all, unique_all) = env.sum_over_ranks(calls, unique_bytes)
    peak_all = HBM_PEAK_GBS * env.world
    # the RUNNING receiver: the same capture arrives in chunks of 128 (and of 8) windows. One call into the library per chunk
    # (lorahip_demod_receive): the append run -- every channel continues at its own read position, which lives on the device --, the
    # packets that completed (those that span chunks too) packed on the device, the queue cleared. No Python per channel.
    running = None
    try:
        cap_ = int(iq.shape[1])
        rows_ = d.receiver_rows(cap_packets=B * (frames + 1), stride=max(8, min(nsyms, 512)))

        n_sig_ = [0]                                 # signals delivered by the steps of the last pass (with set_signals + signal rows)

        def running_pass(chunk_windows):
            chunk = chunk_windows << sf
            d.clear_packets()
            d.rewind()
            d.activate()
            w = n_pk_ = n_work = calls_ = sg_ = 0
            env.barrier()
            t0 = time.perf_counter()
            while w < cap_:
                w = min(cap_, w + chunk)
                n_, k_ = d.receive(iq, w, rows_, async_=True)
                n_pk_ += n_
                calls_ += k_
                n_work += 1
                if with_sig_[0]:
                    sg_ += d.last_signals()
            torch.cuda.synchronize()
            (dt_,) = env.max_over_ranks(time.perf_counter() - t0)
            (calls_all_, pk_all_, n_sig_[0]) = env.sum_over_ranks(calls_, n_pk_, sg_)
            return dt_, calls_all_, pk_all_, n_work
        with_sig_ = [False]
        running = {}
        for cw in (128, 8):
            running_pass(cw)                        # sizes the buffers of the chunked shape
            rb = min((running_pass(cw) for _ in range(3)), key=lambda r_: r_[0])
            ent_ = {"chunk_windows": cw, "work_per_capture": rb[3], "ms_per_work": r4(rb[0] / rb[3] * 1e3), "Msym_s": r4(rb[1] / rb[0] / 1e6),
                    "frac": r4(rb[1] * L.bytes_per_symbol(sf) / rb[0] / 1e9 / (HBM_PEAK_GBS * env.world)), "work_calls": int(rb[1]), "packets": int(rb[2])}
            if cw == 128:
                running = ent_
                running["entry"] = "lorahip_demod_receive (C ABI): one call per chunk"
                running["near_squelch"], running["near_step"] = d.near_threshold()     # of the last pass (activate() resets the counters)
                running["same_packets_as_one_shot"] = bool(rb[2] == n_dev * env.world)
            else:
                running["chunk8"] = ent_
        # the same with the steps PIPELINED (async = 2: step k launched before step k-1's summary is read; packets one step late)
        def piped_pass(chunk_windows):
            chunk = chunk_windows << sf
            d.clear_packets()
            d.rewind()
            d.activate()
            w = n_pk_ = n_work = calls_ = sg_ = 0
            env.barrier()
            t0 = time.perf_counter()
            while w < cap_:
                w = min(cap_, w + chunk)
                # (the C entry itself: the rows are not read between the steps here, so the two stream hand-shakes with torch's stream
                # that LoRaDemod.receive adds for a Python consumer -- an event record and a wait each way -- are left out)
                n_, k_ = d.receive(iq, w, rows_, async_=2, order_with_torch=False)
                n_pk_ += n_
                calls_ += k_
                n_work += 1
                if with_sig_[0]:
                    sg_ += d.last_signals()
            n_, k_ = d.receive_flush(rows_)
            if with_sig_[0]:
                sg_ += d.last_signals()
            (dt_,) = env.max_over_ranks(time.perf_counter() - t0)
            (calls_all_, pk_all_, n_sig_[0]) = env.sum_over_ranks(calls_ + k_, n_pk_ + n_, sg_)
            return dt_, calls_all_, pk_all_, n_work
        for cw in (128, 8):
            piped_pass(cw)
            rb = min((piped_pass(cw) for _ in range(3)), key=lambda r_: r_[0])
            (running if cw == 128 else running["chunk8"])["pipelined"] = {
                "ms_per_work": r4(rb[0] / rb[3] * 1e3), "Msym_s": r4(rb[1] / rb[0] / 1e6),
                "frac": r4(rb[1] * L.bytes_per_symbol(sf) / rb[0] / 1e9 / (HBM_PEAK_GBS * env.world)), "packets": int(rb[2])}
        # RESIDENT steps (async = 3): one kernel launch stays on the device, a step is a 104-byte message and two words back; the kernel packs
        # the packets itself into the rows that came with the step (two sets, alternating).
        # (`depth`: how many steps the receiver may run ahead of the last report -- a step ends with its slowest workgroup, with more steps
        # in flight the fast ones work ahead; the caller cycles depth + 1 sets of rows)
        rows_set_ = [rows_] + [d.receiver_rows(cap_packets=B * (frames + 1), stride=max(8, min(nsyms, 512))) for _ in range(3)]

        def resident_pass(chunk_windows, depth_):
            chunk = chunk_windows << sf
            d.clear_packets()
            d.rewind()
            d.activate()
            w = n_pk_ = n_work = calls_ = k_ = 0
            was_ = False
            env.barrier()
            t0 = time.perf_counter()
            while w < cap_:
                w = min(cap_, w + chunk)
                n_, c_ = d.receive(iq, w, rows_set_[k_ % (depth_ + 1)], async_=3, order_with_torch=False, depth=depth_)
                n_pk_ += n_
                calls_ += c_
                n_work += 1
                k_ += 1
            was_ = d.resident_active()
            n_, c_ = d.receive_flush(rows_set_[k_ % (depth_ + 1)])
            (dt_,) = env.max_over_ranks(time.perf_counter() - t0)
            (calls_all_, pk_all_) = env.sum_over_ranks(calls_ + c_, n_pk_ + n_)
            return dt_, calls_all_, pk_all_, n_work, was_
        for cw, depth_, key_ in ((128, 1, "resident"), (8, 1, "resident"), (8, 3, "resident_depth3")):
            tgt_ = running if cw == 128 else running["chunk8"]
            try:
                resident_pass(cw, depth_)
                rb = min((resident_pass(cw, depth_) for _ in range(3)), key=lambda r_: r_[0])
                tgt_[key_] = {"ms_per_work": r4(rb[0] / rb[3] * 1e3), "Msym_s": r4(rb[1] / rb[0] / 1e6),
                              "frac": r4(rb[1] * L.bytes_per_symbol(sf) / rb[0] / 1e9 / (HBM_PEAK_GBS * env.world)), "packets": int(rb[2]), "kernel_resident": bool(rb[4]),
                              "steps_in_flight": depth_, "same_packets_as_one_shot": bool(rb[2] == n_dev * env.world)}
            except Exception as e:
                tgt_[key_] = {"error": repr(e)[:160]}
                try:
                    d.receive_flush(None)
                except Exception:
                    pass
        # The same steps delivering what the reference block delivers: packets AND the signals "error" / "power" / "snr" (LoRaDemod.cpp:
        # 267-269), into rows registered with lorahip_demod_receive_signal_rows (device memory: a consumer on the device, like the rows)
        d.set_signals(True)
        d.receiver_signal_rows(B * (frames + 2))
        with_sig_[0] = True
        for cw in (128, 8):
            tgt = running if cw == 128 else running["chunk8"]
            running_pass(cw)
            rb = min((running_pass(cw) for _ in range(3)), key=lambda r_: r_[0])
            ns_ = int(n_sig_[0])
            piped_pass(cw)
            rp = min((piped_pass(cw) for _ in range(3)), key=lambda r_: r_[0])
            tgt["with_signals"] = {"Msym_s": r4(rb[1] / rb[0] / 1e6), "frac": r4(rb[1] * L.bytes_per_symbol(sf) / rb[0] / 1e9 / (HBM_PEAK_GBS * env.world)),
                                   "packets": int(rb[2]), "signals": ns_, "signals_pipelined": int(n_sig_[0]),
                                   "pipelined": {"Msym_s": r4(rp[1] / rp[0] / 1e6), "frac": r4(rp[1] * L.bytes_per_symbol(sf) / rp[0] / 1e9 / (HBM_PEAK_GBS * env.world)),
                                                 "packets": int(rp[2])}}
        with_sig_[0] = False
        d.set_signals(False)
        d.receiver_signal_rows(0)
        d.rewind()
    except Exception as e:                          # a measurement beside the contract line: report, do not fail the bench
        running = {"error": repr(e)}
        try:
            d.receive_flush(None); d.set_signals(False); d.receiver_signal_rows(0); d.rewind()
        except Exception:
            pass
    # the same streams handed over as ordinary HOST buffers, one per channel (what a Pothos port gives the block): gathered through
    # the pinned double-buffered upload, then the streaming kernel -- PCIe-bound, reported beside the device-resident figures
    host = iq.cpu().numpy()                         # (B, samples): one buffer per channel for the C ABI (lorahip_demod_run)
    d.clear_packets(); d.activate()
    d.work(host)                                    # the device-side IQ array of the host path is allocated here
    d.clear_packets(); d.activate()
    t0 = time.perf_counter()
    d.work(host)
    from_host = time.perf_counter() - t0
    d.clear_packets()
    d.activate()
    n_pk, ok = WL.check_frame_packets(pk, data, 1 << sf, nsyms)
    # e2e = host wall clock from IQ in HBM to packets in the decoder's layout in HBM; e2e_host = the same to packets in host memory
    res = {"sf": sf, "channels": B * env.world, "channels_per_gpu": B, "work_calls": int(calls_all), "Msym_s_e2e": r4(calls_all / best[0] / 1e6), "e2e_ms": r4(best[0] * 1e3),
           "frac_e2e": r4(calls_all * L.bytes_per_symbol(sf) / best[0] / 1e9 / peak_all), "e2e_host_ms": r4(best[2] * 1e3),
           "kernel_us": r4(best[1] * 1e3), "Msym_s_kernel": r4(calls_all / (best[1] / 1e3) / 1e6),
           "frac_kernel": r4(calls_all * L.bytes_per_symbol(sf) / (best[1] / 1e3) / 1e9 / peak_all),
           # frac_kernel is the BEST of 5 launches (device time); the mean and the median of the same five, and the launch with signals kept
           "kernel_us_mean": r4(k_mean * 1e3), "kernel_us_median": r4(k_median * 1e3),
           "frac_kernel_mean": r4(calls_all * L.bytes_per_symbol(sf) / (k_mean / 1e3) / 1e9 / peak_all),
           "frac_kernel_median": r4(calls_all * L.bytes_per_symbol(sf) / (k_median / 1e3) / 1e9 / peak_all),
           "with_signals": {"kernel_us": r4(k_sig * 1e3), "frac_kernel": r4(calls_all * L.bytes_per_symbol(sf) / (k_sig / 1e3) / 1e9 / peak_all),
                            "frac_kernel_median": r4(calls_all * L.bytes_per_symbol(sf) / (k_sig_median / 1e3) / 1e9 / peak_all),
                            "signals_rank0": int(n_signals), "what": "best and median of 5 launches, like frac_kernel / frac_kernel_median"},
           "lanes_log2": d.stream_lanes(),
           "unique_stream_bytes": int(unique_all), "counted_bytes": int(calls_all * L.bytes_per_symbol(sf)),
           "frac_unique": r4(unique_all / (best[1] / 1e3) / 1e9 / peak_all),
           "packets": n_pk, "packets_device": n_dev, "packets_expected": B * frames, "packets_ok": ok, "staggered_starts": True,
           "from_host_ms": r4(from_host * 1e3), "from_host_GB_s": r4(iq.numel() * 8 / from_host / 1e9), "from_host_Msym_s": r4(calls / from_host / 1e6)}
    res["near_squelch"], res["near_step"] = near                       # decisions within float rounding of their boundary (pass 0)
    res["running"] = running
    if env.rank == 0:                                       # (several ranks: rank 0's own B channels, the others wait on the host)
        try:
            res.update(level3_parity(L, sf, iq, host, nsyms, (ch_, rd_, ln_, sy_), data, n_pk - ok, threads))
        except Exception as e:                      # (reported in its place; the other ranks are waiting at the barrier below: it must be reached)
            res["parity_error"] = repr(e)[:200]
        if env.world == 1 and sf in (7, 10, 12):
            try:
                res["pothos_block"] = pothos_block(sf, host, nsyms, calls, n_pk)
            except Exception as e:                  # beside the contract line: report, do not fail the bench
                res["pothos_block"] = {"error": repr(e)[:160]}
    env.host_barrier()
    del host
    d.close()
    ctx.close()
    del iq
    return res


def section_level3_scaling(env, L, sweeps=((7, (1024, 2048, 4096, 8192, 16384, 24576, 32768)), (8, (2048, 4096)), (9, (1024, 2048)), (10, (512, 1024)), (11, (1024, 2048, 4096)),
                                           (12, (256, 512, 1024, 2048, 4096))), both_grids=False, passes=4):
    """The streaming kernels against the channel count (whole LoRaDemod blocks, the level-3 workload): below the resident set (two
    wavefronts per SIMD: 16384 channels at SF7, 1024 at SF11, 512 at SF12) the device is not full; above it the dispatcher hands every
    free slot the next channel set (SF11: the resident workgroups walk channel after channel, lorahip_wide.hip). A grid that is not a
    whole number of resident sets runs its last partial set on half-empty SIMDs (24576 SF7 channels = 1.5 sets). (both_grids: also a
    persistent grid where it is not the default -- LORAHIP_STREAM_BLOCKS, profiling builds only: profiles/r04/s22_*.) Kernel time =
    HIP events around the launch, the best and the median of `passes` launches."""
    from lora_sdr_amd import workloads as WL
    torch = env.torch
    out = []
    for sf, counts in sweeps:
        ctx = L.Context(sf, device=env.local)
        for B in counts:
            iq, _ = WL.frame_streams(ctx, B, 4, 48, sigma=0.05)
            ent = {"sf": sf, "channels": B}
            for grid in (("library default", None), ("persistent", "512" if sf != 11 else "1024")) if both_grids else (("library default", None),):
                if both_grids:
                    if grid[1] is None:
                        os.environ.pop("LORAHIP_STREAM_BLOCKS", None)
                    else:
                        os.environ["LORAHIP_STREAM_BLOCKS"] = grid[1]
                d = L.LoRaDemod(sf, n_channels=B, device=env.local)
                d.set_mode(1); d.setMTU(48)
                d.work(iq)
                calls = d.work_calls()
                ent["lanes_log2"] = d.stream_lanes()           # lanes per channel the library chose for this channel count (SF7-9)
                t_ramp = time.perf_counter()
                while time.perf_counter() - t_ramp < 0.15:
                    d.clear_packets(); d.activate(); d.work(iq)
                kms = []
                for _ in range(passes):
                    d.clear_packets(); d.activate(); d.work(iq)
                    kms.append(d.kernel_ms())
                best = min(kms)
                d.close()
                key = "default" if grid[1] is None else grid[0]
                ent[key + "_Msym_s"] = r4(calls / (best / 1e3) / 1e6)
                ent[key + "_frac"] = r4(calls * L.bytes_per_symbol(sf) / (best / 1e3) / 1e9 / HBM_PEAK_GBS)
                ent[key + "_kernel_ms"] = r4(best)
                ent[key + "_kernel_ms_median"] = r4(sorted(kms)[len(kms) // 2])
                ent[key + "_frac_median"] = r4(calls * L.bytes_per_symbol(sf) / (sorted(kms)[len(kms) // 2] / 1e3) / 1e9 / HBM_PEAK_GBS)
            if both_grids:
                os.environ.pop("LORAHIP_STREAM_BLOCKS", None)
            out.append(ent)
            del iq
            torch.cuda.empty_cache()
        ctx.close()
    return out


def section_config5(env, L, a, threads):
    """BASELINE configs[4]: SF10, 8192 channels x 64 windows, AWGN at -10 dB SNR per sample (signal power 1, noise variance 10)"""
    import numpy as np
    sh = Shape(env, L, 10, 8192, 64, noise_sigma=5.0 ** 0.5, variant=a.variant)
    elapsed, kernel_ms = sh.measure(a.steps, a.warmup, 0.1)
    ser, off = sh.ser_vs_sent()
    res = {"sf": 10, "channels": 8192, "symbols": 64, "snr_dB": -10, "Msym_s": r4(sh.W * a.steps * env.world / elapsed / 1e6),
           "frac": r4(sh.W * L.bytes_per_symbol(10) / (kernel_ms / 1e3 / a.steps) / 1e9 / HBM_PEAK_GBS), "ser_gpu": ser, "bin_offset": off}
    if env.rank == 0:                                       # (several ranks: rank 0 checks its own 8192 x 64, the others wait on the host)
        try:
            from oracle.oracle import Oracle
            o = Oracle().detect_batch(10, sh.host_iq(), nthreads=threads)
            sent = sh.sym.cpu().numpy().view(np.uint16).astype(np.int64)
            res["ser_cpu"] = float(((o["sym"].astype(np.int64) - sent) % 1024 != off).mean())
            res["gpu_vs_cpu_index_mismatches"] = int((o["sym"] != sh.out["sym"].cpu().numpy().view(np.uint16)).sum())
            res["windows_checked"] = int(sh.W)
        except Exception as e:                              # (the other ranks wait at the barrier below: it must be reached)
            res["oracle_error"] = repr(e)[:200]
    env.host_barrier()
    (res["ser_gpu_max_over_ranks"],) = env.max_over_ranks(ser)
    sh.close()
    return res


def section_mixed(env, L, a, S=16, n_channels=16384, rccl_single=False):
    """BASELINE configs[3]: 16384 channels, SF(c) = 7 + c mod 6, S symbols each; byte-weighted contiguous shards
    (lora_sdr_amd/shard.py), one launch per SF bucket on its own HIP stream, symbols gathered to every rank at the end."""
    import numpy as np
    from lora_sdr_amd import workloads as WL
    from lora_sdr_amd.shard import gather_symbols
    torch = env.torch
    sfs = WL.mixed_sf_channels(n_channels)
    mine = L.shard_channels(sfs, env.world)[env.rank]
    total_bytes = sum(int((sfs == sf).sum()) * S * L.bytes_per_symbol(sf) for sf in range(7, 13))
    sent_all = WL.mixed_sent(sfs, S, env.dev)                        # the same "sent" symbols on every rank: the global truth
    # this rank's channels behind the C-level scheduler (lorahip_mixed_*): buckets by SF, a stream per bucket, event join
    mixed = L.MixedDetector(sfs[mine], device=env.local)
    mixed.set_variant(a.variant)
    parts, offsets, at, spans = [], np.zeros(mine.size, np.int64), 0, []
    for sf, first_row, n_ch in mixed.buckets:
        local = np.nonzero(sfs[mine] == sf)[0]                       # ascending: the order of the bucket's rows
        sym = sent_all[torch.from_numpy(mine[local]).to(env.dev)].to(torch.int16).reshape(-1).contiguous()
        gen = L.Context(sf, device=env.local)
        gen.use_torch_stream()
        parts.append(gen.synth_symbols(sym, ampl=1.0, noise_sigma=a.noise_sigma, seed=0x5EED1000 + sf).reshape(-1))
        torch.cuda.synchronize()
        gen.close()
        offsets[local] = at + np.arange(n_ch, dtype=np.int64) * (S << sf)
        spans.append((sf, first_row, n_ch, at))
        at += n_ch * (S << sf)
    iq = torch.cat(parts) if parts else torch.zeros(2, dtype=torch.float32, device=env.dev)
    del parts
    mixed.plan(offsets, S)
    out = mixed.new_outputs()
    local_ch = np.empty(mine.size, np.int64)
    local_ch[mixed.rows] = mine                                      # row -> global channel
    torch.cuda.synchronize()

    def step():
        mixed.detect(iq, out, sync=False)
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.2:
        step()
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        step()
    env.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    env.barrier()
    (elapsed,) = env.max_over_ranks(time.perf_counter() - t0)
    res = {"channels": n_channels, "symbols_per_channel": S, "sf_rule": "7 + c mod 6", "my_channels_rank0": int(mine.size),
           "Msym_s": r4(n_channels * S * a.steps / elapsed / 1e6), "ms_per_step": r4(elapsed * 1e3 / a.steps),
           "frac_byte_weighted": r4(total_bytes * a.steps / elapsed / 1e9 / (HBM_PEAK_GBS * env.world)),
           "iq_bytes_per_step": int(sum(int((sfs == sf).sum()) * S * (8 << sf) for sf in range(7, 13)))}
    # ---- end of run: the 2 B/symbol results to every rank (north_star: RCCL only as an embarrassingly parallel split) ----
    mixed.synchronize()
    local_sym = out["sym"]
    # every window of this rank's channels against the CPU oracle, like every other section (the ranks check their own shards side by side)
    try:
        from oracle.oracle import Oracle
        orc, bad, n_win, max_db = Oracle(), 0, 0, 0.0
        thr = max(1, min(32, (os.cpu_count() or 1) // env.world))
        for sf, first_row, n_ch, start in spans:
            host = iq[start:start + n_ch * (S << sf)].cpu().numpy()
            o = orc.detect_batch(sf, host, nthreads=thr)
            rows = slice(first_row, first_row + n_ch)
            bad += int((o["sym"] != out["sym"][rows].reshape(-1).cpu().numpy().view(np.uint16)).sum())
            fin = np.isfinite(o["power"])
            max_db = max(max_db, float(np.abs(out["power"][rows].reshape(-1).cpu().numpy() - o["power"])[fin].max()))
            n_win += n_ch * S
        bad_all, win_all = env.sum_over_ranks(bad, n_win)
        (db_all,) = env.max_over_ranks(max_db)
        res["oracle"] = {"windows": int(win_all), "index_mismatches": int(bad_all), "max_dB": r4(db_all)}
    except Exception as e:                                           # pragma: no cover - the checker may not have travelled
        res["oracle"] = {"error": str(e)[:100]}
    dist, made = env.dist, False
    try:
        if dist is None and not rccl_single:
            # one rank, no collective to run: the local result is the global one (bench.py --config mixed sets up a 1-rank
            # RCCL group instead, so that the gather itself is exercised on a single GPU: tests/test_gpu_bench.py)
            full = torch.zeros((n_channels, S), dtype=torch.int16, device=env.dev)
            full[torch.from_numpy(local_ch).to(env.dev)] = local_sym
            res["gather_backend"] = "none (1 rank)"
            res["symbol_errors_vs_sent"] = WL.mixed_errors(full, sfs, S)
            res["symbols_checked"] = n_channels * S
            raise StopIteration
        if dist is None:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=env.dev)
            made = True
        backend_cpu = env.dist is not None and env.backend != "nccl"
        t0 = time.perf_counter()
        full = gather_symbols(local_sym.cpu() if backend_cpu else local_sym, local_ch, n_channels)
        torch.cuda.synchronize()
        res["gather_ms"] = r4((time.perf_counter() - t0) * 1e3)
        res["gather_backend"] = "gloo" if backend_cpu else "nccl (RCCL), %d rank(s)" % dist.get_world_size()
        # check against the sent symbols (constant +1 bin of genChirp vs the demod table)
        bad = WL.mixed_errors(full.to(env.dev), sfs, S)
        res["symbol_errors_vs_sent"] = bad
        res["symbols_checked"] = n_channels * S
    except StopIteration:
        pass
    except Exception as e:                                           # pragma: no cover - environment dependent
        res["gather_backend"] = "failed: %s" % str(e)[:80]
    finally:
        if made:
            dist.destroy_process_group()
    res["scheduler"] = "lorahip_mixed_* (C ABI): %d SF buckets, one stream each" % len(mixed.buckets)
    mixed.close()
    return res


def section_mixed_level3(env, L, n_channels=16384, frames=1, nsyms=16, threads=32):
    """BASELINE configs[3] as RECEIVERS: 16384 channels, SF(c) = 7 + c mod 6, every channel a whole LoRaDemod block (frame sync,
    frequency estimate, packets) behind ONE level-3 handle per rank (lorahip_demod_create_mixed: one part per SF with its own stream
    and host thread, all parts of the device running side by side), the channels sharded over the ranks by lora_sdr_amd/shard.py (no
    data-path collective). Every channel carries `frames` frame(s) of `nsyms` data symbols; all of a rank's channels live in one
    device buffer (lorahip_demod_run_device_segments). Every channel's packets are compared with the verbatim CPU block."""
    from lora_sdr_amd import workloads as WL
    torch = env.torch
    sfs = WL.mixed_sf_channels(n_channels)
    mine = L.shard_channels(sfs, env.world)[env.rank]
    my_sf = sfs[mine]
    parts, first, cnt, at, datas, hosts = [], np.zeros(mine.size, np.int64), np.zeros(mine.size, np.uint64), 0, {}, {}
    for sf in range(7, 13):
        local = np.nonzero(my_sf == sf)[0]
        if local.size == 0:
            continue
        ctx = L.Context(sf, device=env.local)
        iq, data = WL.frame_streams(ctx, local.size, frames, nsyms, sigma=0.05, seed=3 + sf)
        ctx.close()
        n = int(iq.shape[1])
        first[local] = at + np.arange(local.size, dtype=np.int64) * n
        cnt[local] = n
        at += local.size * n
        parts.append(iq.reshape(-1))
        datas[sf] = (local, data, n)
    buf = torch.cat(parts)
    del parts
    d = L.LoRaDemod(channel_sf=my_sf, devices=[env.local])
    d.setMTU(nsyms)
    d.work_segments(buf, first, cnt)                        # pass 0: what the receivers deliver (checked below)
    calls = d.work_calls()
    ch_, rd_, ln_, sy_ = d.packets_arrays()
    calls_sf = {}
    for i in range(len(d.parts)):
        calls_sf[d.parts[i][1]] = 0
    best = None
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.2:
        d.clear_packets(); d.activate(); d.work_segments(buf, first, cnt)
    for _ in range(4):
        d.clear_packets(); d.activate()
        env.barrier()
        t0 = time.perf_counter()
        d.work_segments(buf, first, cnt)
        env.barrier()
        dt, km = env.max_over_ranks(time.perf_counter() - t0, d.kernel_ms())
        if best is None or dt < best[0]:
            best = (dt, km)
    d.clear_packets()
    # byte-weighted: every work() call counts the bytes of its SF
    bytes_rank = 0.0
    from oracle.oracle import Oracle, Ref
    impl, kind = (Ref(), "reference") if Ref.available() else (Oracle(), "port")
    bad = checked = n_pk = ok_pk = 0
    thr = max(1, min(threads, (os.cpu_count() or 1) // env.world))
    for sf, (local, data, n) in datas.items():
        host = buf[int(first[local[0]]):int(first[local[0]]) + local.size * n].reshape(local.size, n).cpu().numpy()
        r = impl.demod_run_many(sf, host, mtu=nsyms, nthreads=thr)
        bytes_rank += float(r["total_calls"]) * L.bytes_per_symbol(sf)
        sel = np.isin(ch_, local)
        remap = np.full(mine.size, -1, np.int64)
        remap[local] = np.arange(local.size)
        lens_sel = ln_[sel]
        starts = np.concatenate([[0], np.cumsum(ln_)])[:-1][sel]
        sy_sel = np.concatenate([sy_[a_:a_ + b_] for a_, b_ in zip(starts.tolist(), lens_sel.tolist())]) if lens_sel.size else np.zeros(0, np.int16)
        bad_ch, overflow = packets_vs_reference((remap[ch_[sel]].astype(np.int32), rd_[sel], lens_sel, sy_sel), r, local.size)
        bad += int(bad_ch.sum()) + overflow
        checked += local.size
        pk = list(zip(remap[ch_[sel]].tolist(), rd_[sel].tolist(), np.split(sy_sel, np.cumsum(lens_sel)[:-1]) if lens_sel.size else []))
        a_, b_ = WL.check_frame_packets(pk, data, 1 << sf, nsyms)
        n_pk += a_; ok_pk += b_
        del host
    calls_all, bad_all, checked_all, bytes_all, npk_all, ok_all = env.sum_over_ranks(calls, bad, checked, bytes_rank, n_pk, ok_pk)
    res = {"channels": n_channels, "sf_rule": "7 + c mod 6", "frames_per_channel": frames, "data_symbols_per_frame": nsyms,
           "object": "lorahip_demod_create_mixed: %d parts on rank 0 (one per SF, own stream + host thread)" % len(d.parts),
           "work_calls": int(calls_all), "e2e_ms": r4(best[0] * 1e3), "kernel_ms_slowest_part": r4(best[1]),
           "Msym_s_e2e": r4(calls_all / best[0] / 1e6),
           "frac_byte_weighted_e2e": r4(bytes_all / best[0] / 1e9 / (HBM_PEAK_GBS * env.world)),
           "oracle_kind": kind, "oracle_channels_checked": int(checked_all), "oracle_channel_mismatches": int(bad_all),
           "packets": int(npk_all), "packets_expected": n_channels * frames, "packets_ok": int(ok_all)}
    d.close()
    del buf
    torch.cuda.empty_cache()
    return res


# ------------------------------------------------------------------------------------------------------------------
def respawn_under_torchrun(a):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run would silently measure ONE rank: start the N ranks ourselves
    (one process per GPU over RCCL, the launch line of the module docstring) -- or fail, never report n_gpus: 1 for a --gpus N run"""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: --gpus %d without WORLD_SIZE: re-launching as %s\n" % (a.gpus, " ".join(cmd)))
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(a)                       # does not return
    env = Env(a)
    import lora_sdr_amd as L
    from lora_sdr_amd import workloads as WL
    single = a.sf is not None or a.moving or a.alias_windows
    sweep = not single and not a.no_sweep and a.config == "default"
    rank0 = env.rank == 0
    # The CPU legs (the reference timed on the host cores, the oracle checks) run on RANK 0, on its own shard, after the timed regions,
    # while the other ranks wait on the host (env.host_barrier): the line of a --gpus N run carries cpu_baseline and oracle like the
    # N = 1 line. Only what needs the whole box to itself (the Pothos block, the PCIe-inclusive rates) stays a single-rank measurement.
    solo = rank0 and env.world == 1

    if a.config == "mixed":
        m = section_mixed(env, L, a, rccl_single=True)
        if rank0:
            line = {"metric": METRIC, "value": m["Msym_s"], "unit": "Msym/s", "n_gpus": env.world, "steps": a.steps, "warmup": a.warmup,
                    "ms_per_step": m["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                    "data": "synthetic", "config": {"workload": "BASELINE configs[3]: 16384 channels, SF = 7 + c mod 6, 16 symbols each, "
                                                                "byte-weighted shards over %d rank(s)" % env.world},
                    "roofline": {"bound": "hbm", "frac": m["frac_byte_weighted"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "achieved": r4(m["frac_byte_weighted"] * HBM_PEAK_GBS), "traffic": None}, "mixed": m, "rccl_ranks": env.rccl_ranks}
        emit(env, line if rank0 else None)
        return

    sf0 = a.sf if a.sf is not None else 7
    ch_def, sy_def = WL.default_geometry(sf0)
    B, S = a.channels or ch_def, a.symbols or sy_def
    sh = Shape(env, L, sf0, B, S, a.noise_sigma, a.variant)
    if a.fine_gather:
        sh.ctx.set_fine_gather(True)
    elapsed, kernel_ms = sh.measure(a.steps, a.warmup, a.ramp_seconds, moving=a.moving, alias=a.alias_windows)
    ser, off = sh.ser_vs_sent(sh.out_moving if a.moving else None) if not a.alias_windows else (None, None)
    if ser is not None:
        (ser,) = env.max_over_ranks(ser)                # every rank's recovered symbols against the ones it sent: the worst rank
    line = None
    threads = min(os.cpu_count() or 1, 64)
    if rank0:
        launch_s = kernel_ms / 1e3 / a.steps
        line = {
            "metric": METRIC, "value": sh.W * a.steps * env.world / elapsed / 1e6, "unit": "Msym/s",
            "n_gpus": env.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed * 1e3 / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "batch %d channels SF=%d (N=%d FFT) x %d symbol windows per channel per step, per GPU" % (B, sf0, sh.N, S),
                       "sf": sf0, "channels_per_gpu": B, "symbols_per_channel": S, "iq_bytes_per_step_per_gpu": sh.W * sh.N * 8,
                       "noise_sigma": a.noise_sigma, "parallelism": "channels sharded, %d rank(s), no data-path collective" % env.world,
                       "kernel_variant": a.variant, "alias_windows": bool(a.alias_windows), "moving_fine_index": bool(a.moving),
                       "fine_gather": bool(a.fine_gather), "ramp_seconds": a.ramp_seconds},
            "symbol_error_rate_vs_sent": ser, "bin_offset": off,
            "roofline": roofline_obj(sf0, sh.W, launch_s, None if (a.moving or a.alias_windows) else traffic_for(sf0, a), L,
                                     shape="moving" if a.moving else "steady", default_geometry=a.channels is None and a.symbols is None and not a.alias_windows),
        }
        if env.rccl_ranks is not None:
            line["rccl_ranks"] = env.rccl_ranks
    # (rank 0's CPU legs: with several ranks the others are waiting at the host barrier below, which must be reached whatever happens here --
    # an exception is recorded in the line in place of the object, and one rank fails loudly at the end instead of all hanging)
    cpu_leg_error = None
    if rank0 and not a.no_cpu_baseline:
        try:
            n_streams = min(B, 512)
            host = sh.host_iq()[:n_streams * S * sh.N]
            cb = cpu_baseline(sf0, host, S * sh.N, n_streams, a.cpu_seconds)
            threads = cb["cores"]
            cb.update(host_cpu_info())
            # the other two flag sets beside -O2 (same sources; result-identical for finite input)
            cb["other_flags"] = [x for x in (cpu_baseline(sf0, host, S * sh.N, n_streams, 2.0, f, probe=False) for f in ("-O3 -fcx-limited-range", "-O3")) if x]
            for x in cb["other_flags"]:
                x.pop("sample", None); x.pop("kind", None); x.pop("unit", None)
            if env.world > 1:
                cb["where"] = "rank 0, on its own shard's IQ, the other %d rank(s) idle at a host barrier" % (env.world - 1)
            line["cpu_baseline"] = cb
        except Exception as e:
            if env.world == 1:
                raise
            cpu_leg_error = line["cpu_baseline"] = {"error": repr(e)[:200]}
    if rank0 and not a.alias_windows:
        try:
            line["oracle"] = sh.oracle_check(threads, moving=a.moving)
            if env.world > 1:
                line["oracle"]["where"] = "rank 0: every window of its own batch"
        except Exception as e:
            if env.world == 1:
                raise
            cpu_leg_error = line["oracle"] = {"error": repr(e)[:200]}
    env.host_barrier()
    if env.world > 1 and not a.alias_windows:
        every = sh.oracle_check_every_rank(moving=a.moving)
        if rank0:
            line["oracle"]["every_rank_sample"] = every
    if solo and not single:
        # the same path fed from HOST buffers (lorahip_detect_batch_host: stage, H2D, launch, results D2H): the PCIe-inclusive
        # rate of SURVEY.md section 8d -- reported beside `value`, never as it
        import numpy as np
        k = min(sh.W, 262144)
        part = sh.host_iq()[:k * sh.N]
        want = sh.out["sym"][:k].cpu().numpy().view(np.uint16)

        def host_rate(buf):
            sh.ctx.detect_batch(buf)                                # staging buffers allocated
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                r = sh.ctx.detect_batch(buf)
            dt = (time.perf_counter() - t0) / reps
            return {"Msym_s": r4(k / dt / 1e6), "GB_s": r4(k * sh.N * 8 / dt / 1e9), "same_symbols": bool(np.array_equal(r["sym"], want))}
        pin = L.pinned_empty(part.shape, part.dtype)                # lorahip_host_alloc: the DMA engine reads it directly
        pin[...] = part
        line["pcie_inclusive"] = dict(host_rate(part), windows=k, buffers="ordinary host memory (gathered through pinned double-buffered staging)",
                                      pinned=host_rate(pin))
        del pin
        # drop-in form (a) of INTEGRATION.md section 1: the reference's own LoRaDemod.cpp with its detector swapped for the level-1 shim
        # (one window per detect(): a launch and a PCIe round trip each), against the unpatched block, one host thread each, on the
        # same samples -- the honest figure for "change two lines and nothing else"
        try:
            from oracle.oracle import Ref
            if Ref.available("dropin") and Ref.available("-O2"):
                sps, ns = 512 * sh.N, 2
                sample = sh.host_iq()[:sps * ns]
                res = {}
                for name, flags in (("shim", "dropin"), ("cpu", "-O2")):
                    impl = Ref(flags)
                    impl.demod_bench(sf0, sample[:8 * sh.N], 8 * sh.N, 1, 1, 1)          # contexts, tables, first launch
                    t0 = time.perf_counter()
                    calls = impl.demod_bench(sf0, sample, sps, ns, 1, 1)
                    res[name] = (calls, time.perf_counter() - t0)
                line["pcie_inclusive"]["level1_shim"] = {
                    "what": "verbatim LoRaDemod.cpp + LoRaDetectorHip (one launch + PCIe round trip per detect()), 1 host thread",
                    "work_calls": int(res["shim"][0]), "us_per_detect": r4(res["shim"][1] / max(res["shim"][0], 1) * 1e6),
                    "Msym_s": r4(res["shim"][0] / res["shim"][1] / 1e6),
                    "cpu_reference_1_thread_Msym_s": r4(res["cpu"][0] / res["cpu"][1] / 1e6)}
        except Exception as e:                                       # pragma: no cover - the drop-in library may not have travelled
            line["pcie_inclusive"]["level1_shim"] = {"error": str(e)[:120]}

    if sweep:
        per_sf, moving, level3 = [], [], []
        for sf in range(7, 13):
            if sf != sf0:
                b2, s2 = WL.default_geometry(sf)
                cur = Shape(env, L, sf, b2, s2, a.noise_sigma, a.variant)
                e2, k2 = cur.measure(a.steps, a.warmup, a.ramp_seconds)
            else:
                cur, e2, k2 = sh, elapsed, kernel_ms
            ser2, off2 = cur.ser_vs_sent()
            ent = {"sf": sf, "channels": cur.B, "symbols": cur.S, "Msym_s": r4(cur.W * a.steps * env.world / e2 / 1e6),
                   "launch_us": r4(k2 * 1e3 / a.steps), "frac": r4(cur.W * L.bytes_per_symbol(sf) / (k2 / 1e3 / a.steps) / 1e9 / HBM_PEAK_GBS),
                   "traffic": traffic_for(sf, a), "ser_vs_sent": ser2, "valu_busy": (counters_for(sf) or {}).get("valu_busy"),
                   "lds_busy": (counters_for(sf) or {}).get("lds_busy"), "instr_per_sample": (counters_for(sf) or {}).get("instr_per_sample")}
            # the locked-receiver shape on the same IQ
            e3, k3 = cur.measure(a.steps, a.warmup, 0.1, moving=True)
            mv = {"sf": sf, "Msym_s": r4(cur.W * a.steps * env.world / e3 / 1e6), "launch_us": r4(k3 * 1e3 / a.steps),
                  "frac": r4(cur.W * L.bytes_per_symbol(sf) / (k3 / 1e3 / a.steps) / 1e9 / HBM_PEAK_GBS),
                  "valu_busy": (counters_for(sf, "moving") or {}).get("valu_busy"), "instr_per_sample": (counters_for(sf, "moving") or {}).get("instr_per_sample")}
            if rank0:                                           # (rank 0's CPU legs: the barrier below must be reached, see above)
                try:
                    if sf != sf0:
                        ent["oracle"] = cur.oracle_check(threads)
                        if not a.no_cpu_baseline:
                            n_streams = min(cur.B, 512)
                            cbs = cpu_baseline(sf, cur.host_iq()[:n_streams * cur.S * cur.N], cur.S * cur.N, n_streams, 2.0, probe=False)
                            ent["cpu_baseline"] = {"value": cbs["value"], "cores": cbs["cores"], "per_core": cbs["per_core"], "flags": "-O2"}
                    else:
                        ent["oracle"] = line["oracle"]
                        if "cpu_baseline" in line and "error" not in line["cpu_baseline"]:
                            ent["cpu_baseline"] = {k: line["cpu_baseline"][k] for k in ("value", "cores", "per_core")}
                            ent["cpu_baseline"]["flags"] = "-O2"
                    mv["oracle"] = cur.oracle_check(threads, moving=True)
                except Exception as e:
                    if env.world == 1:
                        raise
                    cpu_leg_error = ent["error"] = mv["error"] = repr(e)[:200]
            env.host_barrier()
            per_sf.append(ent)
            moving.append(mv)
            if cur is not sh:
                cur.close()
            del cur
            env.torch.cuda.empty_cache()
        sh.close()
        del sh
        env.torch.cuda.empty_cache()
        # every rank runs its own channels of every section (weak scaling, barriers around the timed regions, MAX over ranks for times,
        # sums for counts); the CPU-side checks that need the whole box (reference block on every channel) run on one rank alone
        # (sections beside the contract line: a failure is reported in its place, the line is still printed)
        for sf in range(7, 13):
            try:
                level3.append(section_level3(env, L, sf, threads))
            except Exception as e:
                level3.append({"sf": sf, "error": repr(e)[:200]})
            env.torch.cuda.empty_cache()
        try:
            c5 = section_config5(env, L, a, threads)
        except Exception as e:
            c5 = {"error": repr(e)[:200]}
        env.torch.cuda.empty_cache()
        try:
            mixed = section_mixed(env, L, a)
        except Exception as e:
            mixed = {"error": repr(e)[:200]}
        env.torch.cuda.empty_cache()
        try:
            mixed_l3 = section_mixed_level3(env, L, threads=threads)
        except Exception as e:                          # beside the contract line: report, do not fail the bench
            mixed_l3 = {"error": repr(e)[:200]}
        env.torch.cuda.empty_cache()
        scaling = None
        if env.world == 1:
            try:
                scaling = section_level3_scaling(env, L)
            except Exception as e:
                scaling = {"error": repr(e)[:200]}
        if rank0:
            line["per_sf"], line["moving"] = per_sf, moving
            line["level3"] = level3
            line["config5"] = c5
            line["mixed"] = mixed
            line["mixed_level3"] = mixed_l3
            if scaling is not None:
                line["level3_scaling"] = scaling
    emit(env, line if rank0 else None)
    if cpu_leg_error is not None:                           # the line is out (with the error in it); the run did not measure what it claims
        raise SystemExit("bench.py: a CPU leg on rank 0 failed: %s" % (cpu_leg_error,))


if __name__ == "__main__":
    main()
