"""Channeliser throughput: K channels, decimation D, L taps over n wideband samples.
    python tools/bench_chan.py --channels 8 --decim 8 --taps 64 --samples 67108864"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lora_sdr_amd as L

ap = argparse.ArgumentParser()
ap.add_argument("--channels", type=int, default=8); ap.add_argument("--decim", type=int, default=8)
ap.add_argument("--taps", type=int, default=64); ap.add_argument("--samples", type=int, default=1 << 26)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
ctx = L.Context(7)
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.view_as_complex(torch.randn((a.samples, 2), generator=g, device="cuda"))
freqs = (np.arange(a.channels) - 0.5 * (a.channels - 1)) * (0.8 / a.channels)
ch = L.Channelizer(ctx, freqs, a.decim, L.design_lowpass(a.decim, a.taps))
out = torch.empty((a.channels, a.samples // a.decim + 1), dtype=torch.complex64, device="cuda")
import time
t0 = time.time()
while time.time() - t0 < 0.4:                      # the clocks need ~40 ms of load to leave idle
    ch.run(x, out=out)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    ch.run(x, out=out)
e1.record(); torch.cuda.synchronize()
dt = e0.elapsed_time(e1) * 1e-3 / a.reps
groups = (a.channels + 7) // 8
flops = 8.0 * a.taps * groups * 8 * (a.samples / a.decim)            # complex MACs the kernel issues (padded channel groups)
useful = 8.0 * a.taps * a.channels * (a.samples / a.decim)
print("K=%d D=%d L=%d: %.3f ms per %d Msamples -> %.2f Gsamples/s in (%.1f GB/s), %.2f Gsamples/s out; %.1f TFLOP/s issued (%.1f useful) = %.0f%% of 157 TFLOP/s fp32"
      % (a.channels, a.decim, a.taps, dt * 1e3, a.samples >> 20, a.samples / dt / 1e9, 8 * a.samples / dt / 1e9,
         a.channels * (a.samples / a.decim) / dt / 1e9, flops / dt / 1e12, useful / dt / 1e12, 100 * flops / dt / 157e12))
