"""HBM read ceiling on this box for the two access shapes (tools/, measurement aid)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lora_sdr_amd as L
ctx = L.Context(7); ctx.use_torch_stream()
lib = L.load()
n = 1 << 30
buf = torch.randn(n // 4, device="cuda")
for pattern in (0, 1):
    for bpc in (2, 4, 8, 16):
        for _ in range(3):
            lib.lorahip_membw_probe(ctx._h, C.c_void_p(buf.data_ptr()), n, pattern, bpc)
        ctx.timer_start()
        for _ in range(10):
            lib.lorahip_membw_probe(ctx._h, C.c_void_p(buf.data_ptr()), n, pattern, bpc)
        ms = ctx.timer_stop() / 10
        print("pattern %d blocks/CU %2d : %.1f GB/s" % (pattern, bpc, n / ms / 1e6))
