"""Symbol-error-rate sweep, all on the device: LoRaMod frames -> AWGN -> streaming LoRaDemod, one frame per channel.

    python tools/ser_sweep.py --sf 7 --channels 16384 --nsyms 32 --snr=-16:-4:2

SNR is quoted in the channel bandwidth (signal power ampl^2 over the complex noise power 2 sigma^2), the number LoRa
sensitivity tables use. Per SNR point the tool reports: frames found (a packet posted for the channel), frames whose length is
right, and the symbol error rate inside the found frames (against the sent symbols up to the frame's constant bin offset).
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lora_sdr_amd as L

ap = argparse.ArgumentParser()
ap.add_argument("--sf", type=int, default=7)
ap.add_argument("--channels", type=int, default=16384)
ap.add_argument("--nsyms", type=int, default=32)
ap.add_argument("--snr", default="-16:-4:2", help="lo:hi:step in dB, inclusive (write --snr=-16:-4:2: the value starts with a dash)")
ap.add_argument("--thresh", type=float, default=None, help="setThreshold(dB); default = the block's own default")
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()

sf, N, B = a.sf, 1 << a.sf, a.channels
lo, hi, step = [float(x) for x in a.snr.split(":")]
ctx = L.Context(sf)
g = torch.Generator(device="cuda"); g.manual_seed(a.seed)
sent = torch.randint(0, N, (B, a.nsyms), generator=g, device="cuda", dtype=torch.int32)
clean = ctx.mod_frames(sent.to(torch.int16), sync=0x12, ampl=1.0, padding=1, lead=N // 2 + 5, tail=3 * N)
sent_h = sent.cpu().numpy().astype(np.int64)
print("SF%d: %d channels x %d samples, %d data symbols per frame" % (sf, B, clean.shape[1], a.nsyms))
print("%8s %10s %10s %12s %10s" % ("SNR dB", "found", "len ok", "SER", "ms"))
d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(a.nsyms)
if a.thresh is not None:
    d.setThreshold(a.thresh)
snr = lo
while snr <= hi + 1e-9:
    sigma = float(np.sqrt(0.5 / 10.0 ** (snr / 10.0)))
    iq = ctx.add_awgn(clean.clone(), sigma, seed=a.seed + int(round(snr * 16)) + 4096)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d.activate(); d.work(iq)
    dt = time.perf_counter() - t0
    found = np.zeros(B, bool); len_ok = 0; errs = 0; total = 0
    for ch, _, s in d.packets():
        if found[ch]:
            continue                                   # a second (false) frame in the tail: count the first only
        found[ch] = True
        n = min(len(s), a.nsyms)
        if len(s) == a.nsyms:
            len_ok += 1
        if n == 0:
            continue
        diff = (s[:n].astype(np.int64) - sent_h[ch, :n]) % N
        off = np.bincount(diff, minlength=N).argmax()
        errs += int((diff != off).sum()) + (a.nsyms - n); total += a.nsyms
    print("%8.1f %10.4f %10.4f %12.3e %10.1f" % (snr, found.mean(), len_ok / B, errs / max(total, 1), dt * 1e3))
    snr += step
d.close()
