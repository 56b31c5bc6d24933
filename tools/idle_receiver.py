"""Level 3 on streams that carry no frames: every channel in FRAMESYNC for the whole run -- the state an idle receiver is in most of
the time. Noise only (squelched windows: consume N, reset), and a weak constant tone (unsquelched windows that never sync: consume
N - value and take fIndex every call). Prints kernel time and Msym/s next to the framed workload's.
    python tools/idle_receiver.py --sf 7 --channels 16384 --samples 37317"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lora_sdr_amd as L

ap = argparse.ArgumentParser()
ap.add_argument("--sf", type=int, default=7); ap.add_argument("--channels", type=int, default=16384)
ap.add_argument("--samples", type=int, default=37317); ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
sf, N, B, S = a.sf, 1 << a.sf, a.channels, a.samples
g = torch.Generator(device="cuda"); g.manual_seed(3)
noise = torch.view_as_complex(torch.randn((B, S, 2), device="cuda", generator=g) * 0.7071)
n = torch.arange(S, device="cuda", dtype=torch.float32)
tone = torch.polar(torch.full((S,), 3.0, device="cuda"), 2 * torch.pi * 0.1337 * n)
cases = {"noise only, threshold 3 dB above the noise-only snr": (noise, None), "noise only, threshold -100 dB (never squelched)": (noise, -100.0),
         "noise + a constant tone (unsquelched, never syncs)": (noise + tone[None, :], None)}
for name, (iq, thresh) in cases.items():
    iq = iq.contiguous()
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(48)
    if thresh is not None:
        d.setThreshold(thresh)
    d.work(iq); calls = d.work_calls(); npk = len(d.packets()); d.clear_packets(); d.activate()
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.25:
        d.work(iq); d.clear_packets(); d.activate()
    ts, ks = [], []
    for _ in range(a.reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        d.work(iq); ts.append(time.perf_counter() - t0); ks.append(d.kernel_ms()); d.clear_packets(); d.activate()
    print("SF%d %d ch x %d samples, %-55s %8d calls, %6d packets, kernel %.3f ms, e2e %.3f ms, %.1f Msym/s kernel" %
          (sf, B, S, name + ":", calls, npk, min(ks), min(ts) * 1e3, calls / min(ks) / 1e3), flush=True)
    d.close()
