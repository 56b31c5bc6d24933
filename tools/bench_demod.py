"""Level-3 throughput: B channels of the LoRaDemod block over whole frames, streaming kernel (mode 1) vs host-driven
lock-step rounds (mode 2).   python tools/bench_demod.py --sf 7 --channels 8192 --frames 4 --nsyms 48"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lora_sdr_amd as L

ap = argparse.ArgumentParser()
ap.add_argument("--sf", type=int, default=7); ap.add_argument("--channels", type=int, default=8192)
ap.add_argument("--frames", type=int, default=4); ap.add_argument("--nsyms", type=int, default=48)
ap.add_argument("--sigma", type=float, default=0.05); ap.add_argument("--modes", default="1,2")
a = ap.parse_args()
sf, N, B = a.sf, 1 << a.sf, a.channels
ctx = L.Context(sf)
sync = 0x12
g = torch.Generator(device="cuda"); g.manual_seed(1)
V = 64                                             # distinct channel contents, tiled over the B channels
per_frame = 10 + 2 + a.nsyms
data = torch.randint(0, N, (V, a.frames, a.nsyms), generator=g, device="cuda", dtype=torch.int32)
syms = torch.zeros((V, a.frames, per_frame), dtype=torch.int32, device="cuda")
syms[:, :, 10] = (sync >> 4) * 8; syms[:, :, 11] = (sync & 0xf) * 8                 # LoRaMod.cpp:150-169
syms[:, :, 12:] = data
up = ctx.synth_symbols(syms.reshape(-1).to(torch.int16)).reshape(V, a.frames, per_frame, N)
down = torch.conj(ctx.synth_symbols(torch.zeros(1, dtype=torch.int16, device="cuda")))      # LoRaMod.cpp:172-197
parts = [torch.zeros((V, N // 2 + 5), dtype=torch.complex64, device="cuda")]
for f in range(a.frames):
    parts += [up[:, f, :12].reshape(V, -1), down.repeat(V, 2), down[: N // 4].repeat(V, 1), up[:, f, 12:].reshape(V, -1),
              torch.zeros((V, 3 * N), dtype=torch.complex64, device="cuda")]
base = torch.cat(parts, dim=1)
iq = base.repeat((B + V - 1) // V, 1)[:B].contiguous()
iq = iq + a.sigma * torch.view_as_complex(torch.randn((B, iq.shape[1], 2), generator=g, device="cuda"))
iq = iq.contiguous()
torch.cuda.synchronize()
print("SF%d: %d channels x %d samples (%.1f MB), %d frames of %d data symbols" % (sf, B, iq.shape[1], iq.numel() * 8 / 1e6, a.frames, a.nsyms))
for mode in [int(m) for m in a.modes.split(",")]:
    # one demodulator object, like a running block: the first work() also allocates its staging buffers (reported as
    # "cold"), the following ones reuse them. Packets are verified on the first pass.
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(mode); d.setMTU(a.nsyms)
    times = []
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rounds = d.work(iq)
        times.append(time.perf_counter() - t0)
        if rep == 0:
            calls = d.work_calls(); pk = d.packets()
        else:
            d.packets()
        d.activate()
    dt = min(times[1:])
    ok = 0
    for ch, rd, s in pk[: 4 * V]:
        f = sum(1 for c2, r2, _ in pk[: 64 * V] if c2 == ch and r2 < rd)
        want = (data[ch % V, f].cpu().numpy() + 0) % N
        ok += int(len(s) == a.nsyms and np.array_equal((s.astype(np.int64) - want) % N, np.full(a.nsyms, (s[0] - want[0]) % N)))
    print("  mode %d: %.1f ms warm (%.1f ms cold), %d work() calls in %d rounds -> %.2f Msym/s; %d packets (expected %d), %d/%d checked packets carry the sent symbols (constant bin offset)"
          % (mode, dt * 1e3, times[0] * 1e3, calls, rounds, calls / dt / 1e6, len(pk), B * a.frames, ok, min(len(pk), 4 * V)))
    d.close()
