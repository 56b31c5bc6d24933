"""What the reference's bit-exact operation graph costs: the opt-in contracted build of the batch kernels
(lorahip_set_variant(ctx, LORAHIP_VARIANT_FMA = 40): every complex multiply as one packed multiply + one packed FMA) against the
default kernels, on the steady-state shape at bench.py's geometry.
    python tools/fma_report.py [sf ...]
Per SF: launch time and roofline fraction of both builds (HIP events around 100 launches, alternated), the largest bin error
relative to the window's peak (north_star's tolerance is 1e-4), index mismatches on windows whose peak margin exceeds 1e-3 (must
be 0), on all signal windows, and on noise-only windows (no signal: the arg-max is a noise bin, ties are a matter of the last place).
The SQ_INSTS_VALU counts come from separate rocprofv3 --pmc passes (tools/gpu_r04.sh fma)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lora_sdr_amd as L
from lora_sdr_amd import workloads as WL

FMA = 40
sfs = [int(x) for x in sys.argv[1:]] or [7, 10, 12]
dev = torch.device("cuda", 0)
for sf in sfs:
    N = 1 << sf
    B, S = WL.default_geometry(sf)
    W = B * S
    ctx = L.Context(sf); ctx.use_torch_stream()
    g = torch.Generator(device=dev); g.manual_seed(77 + sf)
    sym = torch.randint(0, N, (W,), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
    iq = ctx.synth_symbols(sym, ampl=1.0, noise_sigma=0.5, seed=0xF3A + sf)
    out = {v: dict(sym=torch.empty(W, dtype=torch.int16, device=dev), power=torch.empty(W, dtype=torch.float32, device=dev),
                   powerAvg=torch.empty(W, dtype=torch.float32, device=dev), fIndex=torch.empty(W, dtype=torch.float32, device=dev)) for v in (0, FMA)}
    batch = {v: ctx.make_batch(iq, W, out[v]["sym"], out[v]["power"], out[v]["powerAvg"], out[v]["fIndex"]) for v in (0, FMA)}
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.3:
        for _ in range(10):
            ctx.detect_batch_raw(batch[0])
        torch.cuda.synchronize()
    best = {0: 1e9, FMA: 1e9}
    for rep in range(3):
        for v in (0, FMA):
            ctx.set_variant(v)
            for _ in range(10):
                ctx.detect_batch_raw(batch[v])
            ctx.timer_start()
            for _ in range(100):
                ctx.detect_batch_raw(batch[v])
            best[v] = min(best[v], ctx.timer_stop() / 100)
    torch.cuda.synchronize()
    frac = {v: W * L.bytes_per_symbol(sf) / (best[v] / 1e3) / 1e9 / 8000.0 for v in best}
    same_all = int((out[0]["sym"] != out[FMA]["sym"]).sum())
    dpow = float((out[0]["power"] - out[FMA]["power"]).abs().max())
    # bins of a sample of the windows, both builds
    k = min(W, (1 << 24) >> sf)
    bins = {}
    for v in (0, FMA):
        ctx.set_variant(v)
        bins[v] = ctx.detect_batch(iq[:k * N], want_fft=True)
    torch.cuda.synchronize()
    a, b = bins[0]["fft"], bins[FMA]["fft"]
    peak = a.abs().amax(dim=1, keepdim=True)
    rel = float(((a - b).abs() / peak).max())
    m2 = (a.real ** 2 + a.imag ** 2)
    top2 = torch.topk(m2, 2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1]) / top2[:, 0]
    clear = margin > 1e-3
    mism_clear = int((bins[0]["sym"][clear] != bins[FMA]["sym"][clear]).sum())
    # noise-only windows
    nz = (0.5 * torch.randn(k * N, 2, generator=g, device=dev)).view(-1)
    nz = torch.view_as_complex(nz.view(-1, 2).contiguous())
    res = {}
    for v in (0, FMA):
        ctx.set_variant(v)
        res[v] = ctx.detect_batch(nz)
    torch.cuda.synchronize()
    noise_mism = int((res[0]["sym"] != res[FMA]["sym"]).sum())
    print("SF%d: default %.1f us/launch (frac %.4f), contracted %.1f us/launch (frac %.4f): %+.1f %%; %d windows: index mismatches %d (signal windows, all), "
          "%d of %d with peak margin > 1e-3; max |dbin| / peak %.2e (%d windows); max |dpower| %.2e dB; noise-only windows: %d of %d indices differ"
          % (sf, best[0] * 1e3, frac[0], best[FMA] * 1e3, frac[FMA], 100.0 * (best[0] / best[FMA] - 1.0), W, same_all, mism_clear, int(clear.sum()), rel, k, dpow,
             noise_mism, k), flush=True)
    ctx.close()
    del iq, out, bins, res, nz
    torch.cuda.empty_cache()
