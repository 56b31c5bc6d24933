"""Level-3 throughput: B channels of the LoRaDemod block over whole frames, streaming kernel (mode 1) vs host-driven
lock-step rounds (mode 2).   python tools/bench_demod.py --sf 7 --channels 8192 --frames 4 --nsyms 48 [--fine-gather]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lora_sdr_amd as L
from lora_sdr_amd import workloads as WL

ap = argparse.ArgumentParser()
ap.add_argument("--sf", type=int, default=7); ap.add_argument("--channels", type=int, default=8192)
ap.add_argument("--frames", type=int, default=4); ap.add_argument("--nsyms", type=int, default=48)
ap.add_argument("--sigma", type=float, default=0.05); ap.add_argument("--modes", default="1,2")
ap.add_argument("--fine-gather", action="store_true", help="A/B: read the fine-tune table in HBM (round-1 path)")
ap.add_argument("--ramp-seconds", type=float, default=0.25, help="work() passes back to back before the timed ones (device at its loaded clocks)")
ap.add_argument("--reps", type=int, default=4, help="work() passes over the same streams (the first one is the cold one)")
a = ap.parse_args()
sf, N, B = a.sf, 1 << a.sf, a.channels
ctx = L.Context(sf)
iq, data = WL.frame_streams(ctx, B, a.frames, a.nsyms, sigma=a.sigma)
print("SF%d: %d channels x %d samples (%.1f MB), %d frames of %d data symbols" % (sf, B, iq.shape[1], iq.numel() * 8 / 1e6, a.frames, a.nsyms))
for mode in [int(m) for m in a.modes.split(",")]:
    # one demodulator object, like a running block: the first work() also allocates its staging buffers (reported as
    # "cold"), the following ones reuse them. Packets are verified on the first pass.
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(mode); d.setMTU(a.nsyms)
    if a.fine_gather:
        d.set_fine_gather(True)
    times, kms = [], []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d.work(iq); cold = time.perf_counter() - t0                             # pass 0: checked below; allocates the staging buffers
    calls = d.work_calls(); pk = d.packets(); d.activate()
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < a.ramp_seconds:
        d.work(iq); d.clear_packets(); d.activate()
    for rep in range(max(2, a.reps)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rounds = d.work(iq)
        times.append(time.perf_counter() - t0); kms.append(d.kernel_ms())
        d.clear_packets()
        d.activate()
    dt = min(times)
    n, ok = WL.check_frame_packets(pk, data, N, a.nsyms)
    print("  mode %d%s: %.2f ms warm (%.1f ms cold), kernel %.3f ms, %d work() calls in %d rounds -> %.2f Msym/s end to end, %.2f Msym/s kernel; %d packets (expected %d), %d/%d carry the sent symbols (constant bin offset)"
          % (mode, " (table gather)" if a.fine_gather else "", dt * 1e3, cold * 1e3, min(kms), calls, rounds, calls / dt / 1e6,
             calls / (min(kms) / 1e3) / 1e6 if min(kms) > 0 else float("nan"), len(pk), B * a.frames, ok, n))
    d.close()
