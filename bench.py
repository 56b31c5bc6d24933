#!/usr/bin/env python
"""Benchmark of the MI355X LoRa demod hot path: Msymbols/s demodulated (dechirp + FFT + argmax).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--sf 7 --channels 4096 --symbols 256]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (one lorahip_detect_batch launch) over one batch of
synthetic IQ already resident in HBM: `channels` channels x `symbols` symbol windows of 2^SF
cf32 samples. Default workload = BASELINE.json configs[1]: 4096 channels SF7 (N=128 FFT),
256 windows per channel = 1 GiB of IQ per step. With N > 1 every rank demodulates its own
`channels` channels (independent units, no data-path collective): weak scaling; the value is
the whole-job aggregate.

Rank 0 prints ONE JSON line. Besides the contract fields it carries
  roofline      HBM roofline of the detect kernel: algorithmic bytes/launch (8*2^SF+14 per
                window) / average launch duration measured with HIP events
                recorded on the launch stream inside the C ABI (lorahip_timer_start/stop)
  cpu_baseline  the reference CPU path (oracle/_ref: the real LoRaDemod.cpp + kissfft, or the
                oracle's C port where that is absent) timed on this box's host cores on a
                bounded sample of the same IQ
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--ramp-seconds", type=float, default=0.3,
                    help="untimed launches before the warm-up steps until the GPU has left its idle clocks "
                         "(a cold MI355X needs ~40 ms of load to ramp: tools/ramp.py, profiles/r01/s4_clock_ramp.txt)")
    ap.add_argument("--sf", type=int, default=7)
    ap.add_argument("--channels", type=int, default=None, help="channels per GPU (default: 1 GiB of IQ per step)")
    ap.add_argument("--symbols", type=int, default=None, help="symbol windows per channel per step")
    ap.add_argument("--noise-sigma", type=float, default=0.5, help="AWGN per I/Q component (signal amplitude 1)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--moving", action="store_true",
                    help="diagnostic: per-window fine-tune error and start index (the DATASYMBOLS shape of a locked receiver: "
                         "LoRaDemod.cpp:160-162 moves the index every sample) instead of the launch-uniform steady state")
    ap.add_argument("--alias-windows", action="store_true",
                    help="diagnostic: every window reads window 0 (no HBM traffic): the compute-only time of the kernel")
    ap.add_argument("--traffic", type=float, default=None,
                    help="HBM bytes per launch from a separate rocprofv3 --pmc pass (profiles/), if known")
    return ap.parse_args()


def default_geometry(sf):
    # BASELINE.json configs: 4096 channels SF7, 1024 channels SF12; in between keep 1 GiB per step
    channels = {7: 4096, 8: 4096, 9: 2048, 10: 2048, 11: 1024, 12: 1024}.get(sf, 4096)
    symbols = (1 << 30) // (channels * (8 << sf))
    return channels, max(symbols, 1)


def cpu_baseline(sf, iq_host, samples_per_stream, n_streams, seconds):
    """Time the reference CPU path on a bounded sample of the same IQ. Returns the JSON object."""
    from oracle.oracle import Oracle, Ref
    cores = os.cpu_count() or 1
    if Ref.available():
        impl, kind, what = Ref(), "reference", "LoRaDemod.cpp+LoRaDetector.hpp+kissfft.hh compiled in place (g++ -O2, no FMA)"
    else:
        impl, kind, what = Oracle(), "port", "oracle/lora_oracle.c restatement (gcc -O2, no FMA)"
    # pick the thread count that is fastest on this box (SMT / allocator contention can make "all
    # hardware threads" slower), with ~1 s probes, then run that for ~`seconds` of wall time
    best = None
    for threads in sorted({min(cores, n_streams), max(1, cores // 2), max(1, cores // 4), max(1, cores // 8)}, reverse=True):
        threads = min(threads, n_streams)
        t0 = time.perf_counter()
        calls = impl.demod_bench(sf, iq_host, samples_per_stream, n_streams, threads, 1)
        dt = time.perf_counter() - t0
        rep = max(1, int(1.0 / max(dt, 1e-4)))
        t0 = time.perf_counter()
        calls = impl.demod_bench(sf, iq_host, samples_per_stream, n_streams, threads, rep)
        dt = time.perf_counter() - t0
        if best is None or calls / dt > best[0]:
            best = (calls / dt, threads, dt / rep)
    threads = best[1]
    repeat = max(1, int(seconds / max(best[2], 1e-4)))
    t0 = time.perf_counter()
    calls = impl.demod_bench(sf, iq_host, samples_per_stream, n_streams, threads, repeat)
    dt = time.perf_counter() - t0
    want = n_streams
    return {"value": calls / dt / 1e6, "unit": "Msym/s", "cores": threads, "kind": kind,
            "sample": "%d channels x %d samples of the same SF%d IQ, %d passes = %d work() calls (one dechirp+FFT+detect each) in %.1f s wall; %s"
                      % (want, samples_per_stream, sf, repeat, calls, dt, what),
            "host_cores_total": cores}


def main():
    a = parse()
    import torch
    import lora_sdr_amd as L
    from lora_sdr_amd.shard import bytes_per_symbol

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # test hooks (tools/gpu_session.sh "multi"): run several ranks on ONE GPU over gloo to exercise the N > 1 code path
    # where only a single device exists; a real multi-GPU run uses neither
    backend = os.environ.get("LORA_BENCH_BACKEND", "nccl")
    if os.environ.get("LORA_BENCH_ONE_DEVICE"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    sf = a.sf
    N = 1 << sf
    ch_def, sy_def = default_geometry(sf)
    B = a.channels or ch_def
    S = a.symbols or sy_def
    W = B * S

    ctx = L.Context(sf, device=local)
    ctx.set_variant(a.variant)
    ctx.use_torch_stream()

    # synthetic input, generated in HBM: random symbols per (channel, window), continuous stream per channel
    g = torch.Generator(device=dev)
    g.manual_seed(0x10AA + rank)
    sym = torch.randint(0, N, (W,), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
    iq = ctx.synth_symbols(sym, ampl=1.0, noise_sigma=a.noise_sigma, seed=0x5EED0000 + rank)
    out = dict(sym=torch.empty(W, dtype=torch.int16, device=dev), power=torch.empty(W, dtype=torch.float32, device=dev),
               powerAvg=torch.empty(W, dtype=torch.float32, device=dev), fIndex=torch.empty(W, dtype=torch.float32, device=dev))
    offsets = torch.zeros(W, dtype=torch.int64, device=dev) if a.alias_windows else None
    fine_err = fine_idx0 = None
    if a.moving:
        fine_err = (torch.rand(W, generator=g, device=dev) * 4.0 - 2.0).to(torch.float32)
        fine_idx0 = torch.randint(0, 128 * N, (W,), generator=g, device=dev, dtype=torch.int32)
    batch = ctx.make_batch(iq, W, out["sym"], out["power"], out["powerAvg"], out["fIndex"], chirp_sel_all=L.CHIRP_UP,
                           offsets=offsets, fine_err=fine_err, fine_idx0=fine_idx0)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # clock ramp: the first ~40 ms of load after idle run at reduced clocks (profiles/r01/s4_clock_ramp.txt);
    # keep launching (untimed) until that is over, then the W warm-up steps, then the timed K steps
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < a.ramp_seconds:
        for _ in range(10):
            ctx.detect_batch_raw(batch)
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        ctx.detect_batch_raw(batch)
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(a.steps):
        ctx.detect_batch_raw(batch)
    kernel_ms = ctx.timer_stop()          # HIP events on the launch stream, around exactly the K launches
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    # correctness of what was timed: recovered symbols vs sent (+ a cross-check of a slice vs the CPU oracle below)
    got = out["sym"].to(torch.int32) & 0xffff
    sent = sym.to(torch.int32) & 0xffff
    diff = (got - sent) % N
    bin_offset = int(torch.mode(diff).values)
    ser = float((diff != bin_offset).float().mean())
    # genChirp's phase ramp is one sample ahead of the demod's table: a window-aligned
    # symbol s lands in bin s+1; the frame sync of the real receiver removes that constant

    if rank == 0:
        total_syms = W * a.steps * world
        value = total_syms / elapsed / 1e6
        launch_s = kernel_ms / 1e3 / a.steps
        traffic = a.traffic
        if traffic is None and a.channels is None and a.symbols is None:
            # measured in a separate rocprofv3 --pmc pass of this same command (tools/gpu_session.sh pmc), committed
            try:
                t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["per_sf"][str(sf)]
                traffic = t.get("total_bytes")
            except Exception:
                traffic = None
        alg_bytes = W * bytes_per_symbol(sf)
        achieved = alg_bytes / launch_s / 1e9
        line = {
            "metric": "Msymbols/sec demodulated (dechirp+FFT+argmax)", "value": value, "unit": "Msym/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed * 1e3 / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "batch %d channels SF=%d (N=%d FFT) x %d symbol windows per channel per step, per GPU"
                                   % (B, sf, N, S), "sf": sf, "channels_per_gpu": B, "symbols_per_channel": S,
                       "iq_bytes_per_step_per_gpu": W * N * 8, "noise_sigma": a.noise_sigma,
                       "parallelism": "channels sharded, %d rank(s), no data-path collective" % world,
                       "kernel_variant": a.variant, "alias_windows": bool(a.alias_windows), "moving_fine_index": bool(a.moving),
                       "ramp_seconds": a.ramp_seconds},
            "symbol_error_rate_vs_sent": ser, "bin_offset": bin_offset,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "lorahip detect (dechirp+FFT+detect fused)", "launch_us": launch_s * 1e6,
                         "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_symbol": bytes_per_symbol(sf)},
        }
        if world == 1 and not a.no_cpu_baseline:
            n_streams = min(B, 512)
            host = iq[:n_streams * S * N].cpu().numpy()
            line["cpu_baseline"] = cpu_baseline(sf, host, S * N, n_streams, a.cpu_seconds)
            # the same slice through the oracle's batch checker: indices must agree exactly
            from oracle.oracle import Oracle
            k = min(W, 2048)
            o = Oracle().detect_batch(sf, host[:k * N], nthreads=os.cpu_count() or 1)
            line["oracle_index_mismatches_in_%d" % k] = int((o["sym"] != out["sym"][:k].cpu().numpy().view("uint16")).sum())
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
