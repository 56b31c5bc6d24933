"""The whole receive chain on one GPU, stage by stage: encoder symbols (golden, from the verbatim LoRaEncoder.cpp) ->
batched modulator -> AWGN -> [wideband mix -> channeliser] -> streaming demodulator -> batched decoder -> payload bytes, checked
against what was fed to the encoder.   python tools/bench_chain.py --sf 7 --channels 8192 [--wideband 8]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lora_sdr_amd as L

ap = argparse.ArgumentParser()
ap.add_argument("--sf", type=int, default=7, choices=[7, 9, 10, 12]); ap.add_argument("--channels", type=int, default=8192)
ap.add_argument("--cr", default="4/8"); ap.add_argument("--sigma", type=float, default=0.3)
ap.add_argument("--wideband", type=int, default=0, help="K: put groups of K channels into one wideband stream at 16x the channel rate and channelise it back")
a = ap.parse_args()
sf, N, B = a.sf, 1 << a.sf, a.channels
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "codec_kat.npz"))
rdd = {"4/4": 0, "4/5": 1, "4/6": 2, "4/7": 3, "4/8": 4}[a.cr]
case = next(i for i in range(int(g["count"])) if tuple(int(v) for v in g["cfg_%d" % i][:4]) == (sf, 0, rdd, 1)
            and int(g["cfg_%d" % i][6]) == 1 and int(g["cfg_%d" % i][7]) == 0 and int(g["res_%d" % i][2]) == 0)
syms, data = g["syms_%d" % case], g["data_%d" % case]


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3


ctx = L.Context(sf)
tx = torch.from_numpy(np.tile(syms.astype(np.int16), (B, 1))).cuda()
d = L.LoRaDemod(sf, n_channels=B); d.setMTU(len(syms))
dec = L.LoRaDecoder(); dec.setSpreadFactor(sf); dec.setCodingRate(a.cr); dec.enableCrcc(True); dec.enableErrorCheck(True)
print("SF%d, %d channels, one frame of %d symbols (%d payload bytes, CR %s) each" % (sf, B, len(syms), len(data), a.cr))
for rep in range(3):
    iq, t_mod = timed(lambda: ctx.mod_frames(tx, padding=2, lead=N // 2 + 3, tail=3 * N))
    _, t_awgn = timed(lambda: ctx.add_awgn(iq, sigma=a.sigma, seed=5 + rep))
    t_chan = 0.0
    if a.wideband:
        K, D = a.wideband, 16
        assert B % K == 0
        freqs = (np.arange(K) - 0.5 * (K - 1)) * (0.8 / K)
        T = iq.shape[1]
        spec = torch.fft.fft(iq.view(B // K, K, T), dim=2)
        wide_spec = torch.zeros((B // K, K, T * D), dtype=torch.complex64, device="cuda")
        half = T // 2
        wide_spec[:, :, :half] = spec[:, :, :half]; wide_spec[:, :, -(T - half):] = spec[:, :, half:]
        up = torch.fft.ifft(wide_spec, dim=2) * D
        n = torch.arange(T * D, device="cuda", dtype=torch.float64)
        carriers = torch.exp(2j * np.pi * torch.from_numpy(freqs).cuda()[:, None] * n[None, :]).to(torch.complex64)
        wide = (up * carriers[None]).sum(dim=1).contiguous()                  # (B/K, T*D): the test input, not part of the chain
        del spec, wide_spec, up
        chans = [L.Channelizer(ctx, freqs, D, L.design_lowpass(D, 128, cutoff=0.6 / D))]
        iq, t_chan = timed(lambda: chans[0].run_captures(wide).view(B, T))     # all B/K captures in one launch
        chans[0].close()
    d.activate()
    _, t_dem = timed(lambda: d.work(iq))
    (ps, pn, pc), t_get = timed(lambda: d.packets_device())
    (out, out_len, dropped), t_dec = timed(lambda: dec.decode_batch(ps, pn))
    want = torch.from_numpy(data.astype(np.uint8)).cuda()
    ok = int(((out_len == len(data)) & (out[:, :len(data)] == want[None, :]).all(dim=1)).sum()) if out.shape[0] else 0
    tot = t_chan + t_dem + t_get + t_dec
    print("  run %d: modulate %.2f ms, awgn %.2f ms | channelise %.2f ms, demodulate %.3f ms, packets to the decoder's layout %.3f ms, decode %.3f ms"
          " -> %d/%d payloads correct; receive side %.3f ms = %.2f M frames/s, %.1f Msym/s, %.1f MB/s of payload"
          % (rep, t_mod, t_awgn, t_chan, t_dem, t_get, t_dec, ok, B, tot, B / tot / 1e3, B * len(syms) / tot / 1e3, B * len(data) / tot / 1e3), flush=True)
