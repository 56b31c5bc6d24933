"""Round-2 kernel exploration on one GPU, one process: the locked-receiver batch shape (per-window fine-tune error and start
index) over kernel variants and the B-table swizzle; needs a library built with --all-variants.
    python tools/explore_r02.py moving | python tools/explore_r02.py stream (run under LORAHIP_STREAM_ALT=0/1/2)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lora_sdr_amd as L
from lora_sdr_amd import workloads as WL

what = sys.argv[1] if len(sys.argv) > 1 else "moving"
dev = torch.device("cuda", 0)


def moving(sf, variant, steps=60):
    N = 1 << sf
    B, S = WL.default_geometry(sf)
    W = B * S
    ctx = L.Context(sf)
    ctx.set_variant(variant)
    ctx.use_torch_stream()
    g = torch.Generator(device=dev); g.manual_seed(sf)
    sym = torch.randint(0, N, (W,), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
    iq = ctx.synth_symbols(sym, noise_sigma=0.5, seed=7)
    out = [torch.empty(W, dtype=torch.int16, device=dev)] + [torch.empty(W, dtype=torch.float32, device=dev) for _ in range(3)]
    fe = (torch.rand(W, generator=g, device=dev) * 4.0 - 2.0).to(torch.float32)
    fi = torch.randint(0, 128 * N, (W,), generator=g, device=dev, dtype=torch.int32)
    b = ctx.make_batch(iq, W, out[0], out[1], out[2], out[3], chirp_sel_all=L.CHIRP_UP, fine_err=fe, fine_idx0=fi)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
        for _ in range(10):
            ctx.detect_batch_raw(b)
        torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(steps):
        ctx.detect_batch_raw(b)
    ms = ctx.timer_stop()
    chk = int(out[0].to(torch.int32).sum())
    ctx.close()
    us = ms * 1e3 / steps
    print("SF%-2d moving variant %-2d: %8.1f us/launch  %8.1f Msym/s  frac %.3f   (checksum %d)"
          % (sf, variant, us, W / us, W * L.bytes_per_symbol(sf) / (us * 1e-6) / 8e12, chk), flush=True)


if what == "moving":
    # the shipped library knows 0 (default), 1 (generic) and 10; a --all-variants build also the round-1/2 A/B numbers
    full = {7: [0, 30, 13, 10], 8: [0, 30, 31, 11, 13, 10], 9: [0, 30, 31, 12, 13, 16, 10], 10: [0, 30, 31, 12, 13, 14, 16, 10], 11: [0, 10], 12: [0, 10]}
    plan = full if os.environ.get("EXPLORE_ALL") else {sf: [0, 10] for sf in range(6, 13)}
    for sf, vs in plan.items():
        for v in vs:
            moving(sf, v)
elif what == "stream":
    alt = os.environ.get("LORAHIP_STREAM_ALT", "0")
    for sf in (6, 7, 8, 9, 10, 11, 12):
        B = WL.LEVEL3_CHANNELS[sf]
        ctx = L.Context(sf)
        iq, data = WL.frame_streams(ctx, B, 4, 48, sigma=0.05)
        d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(48)
        best = None
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            d.work(iq)
            dt = time.perf_counter() - t0
            if rep == 0:
                calls = d.work_calls(); pk = d.packets()
                n, ok = WL.check_frame_packets(pk, data, 1 << sf, 48)
            else:
                d.packets()
                if best is None or d.kernel_ms() < best[1]:
                    best = (dt, d.kernel_ms())
            d.activate()
        print("SF%-2d stream alt %s: kernel %.3f ms  %8.1f Msym/s kernel (frac %.3f), e2e %.2f ms; %d/%d packets ok"
              % (sf, alt, best[1], calls / best[1] / 1e3, calls * L.bytes_per_symbol(sf) / (best[1] * 1e-3) / 8e12, best[0] * 1e3, ok, n), flush=True)
        d.close(); ctx.close(); del iq
        torch.cuda.empty_cache()
