"""When and where the workgroups of a streaming launch (demodStreamWide, SF11 / SF12: one channel per workgroup) ran.
Needs the profiling build:   python tools/build_variant.py tl -DLORAHIP_WG_TIMELINE --only lorahip_wide.hip
    LORAHIP_LIB=lora_sdr_amd/liblorahip_tl.so python tools/wg_timeline.py <sf> <channels> [passes]
Per pass: the kernel's span, the spread of workgroup durations and start times, how many workgroups each compute unit ran, and the
time the compute units spent with fewer workgroups than they can hold (the tail of a grid of few, long workgroups)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lora_sdr_amd as L
from lora_sdr_amd import workloads as WL
from lora_sdr_amd import _lib

sf, B = int(sys.argv[1]), int(sys.argv[2])
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 4
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.lorahip_debug_wg_timeline.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.lorahip_debug_wg_waves.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
ctx = L.Context(sf)
iq, data = WL.frame_streams(ctx, B, 4, 48, sigma=0.05)
d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(48)
d.work(iq); d.clear_packets(); d.activate()
TICK_US = 0.01                                             # wall_clock64: 100 MHz
for p in range(passes):
    d.work(iq); torch.cuda.synchronize()
    km = d.kernel_ms(); d.clear_packets(); d.activate()
    tl = np.zeros((16384, 4), dtype=np.uint64)
    assert lib.lorahip_debug_wg_timeline(tl.ctypes.data, tl.nbytes) == 0
    tl = tl[:B]
    if os.environ.get("WG_TIMELINE_SAVE"):
        np.save("%s_sf%d_%d_pass%d.npy" % (os.environ["WG_TIMELINE_SAVE"], sf, B, p), tl)
        wv = np.zeros((16384, 4), dtype=np.uint32)
        assert lib.lorahip_debug_wg_waves(wv.ctypes.data, wv.nbytes) == 0
        np.save("%s_sf%d_%d_pass%d_waves.npy" % (os.environ["WG_TIMELINE_SAVE"], sf, B, p), wv[:B])
    t0, t1 = tl[:, 0].astype(np.float64) * TICK_US, tl[:, 1].astype(np.float64) * TICK_US
    org = t0.min(); t0 -= org; t1 -= org
    hw = (tl[:, 2] & np.uint64(0xffffffff)).astype(np.int64); xcc = (tl[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    unit = ((xcc * 8 + se) * 2 + sh) * 16 + cu             # a compute unit's identity
    dur = t1 - t0
    calls = tl[:, 3].astype(np.int64)
    print("pass %d: SF%d %d channels, kernel %.3f ms (events), workgroup span %.3f ms; %d distinct compute units, %d XCCs" %
          (p, sf, B, km, t1.max() / 1e3, len(np.unique(unit)), len(np.unique(xcc))))
    print("   workgroup duration us: min %.0f p10 %.0f median %.0f p90 %.0f max %.0f;  calls per channel %d..%d;  us per call: median %.2f" %
          (dur.min(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max(), calls.min(), calls.max(), np.median(dur / np.maximum(calls, 1))))
    late = np.sort(t0)
    print("   start times us: %d workgroups at < 50 us; then p25 %.0f median %.0f p75 %.0f last %.0f" %
          (int((t0 < 50).sum()), np.percentile(late, 25), np.median(late), np.percentile(late, 75), late[-1]))
    per_unit = np.bincount(unit, minlength=int(unit.max()) + 1); per_unit = per_unit[per_unit > 0]
    print("   workgroups per compute unit: min %d median %d max %d;  per XCC: %s" % (per_unit.min(), int(np.median(per_unit)), per_unit.max(),
                                                                                        np.bincount(xcc, minlength=8).tolist()))
    # resident workgroups over time (whole device), sampled
    ts = np.linspace(0, t1.max(), 41)
    res = [(int(((t0 <= x) & (t1 > x)).sum())) for x in ts]
    print("   resident workgroups at 0, 2.5 %%, ... of the span: %s" % res)
    full = max(res)
    area = float(np.trapezoid(np.minimum(res, full), ts)) / (full * t1.max())
    print("   mean residency %.3f of the peak %d;  sum of workgroup time %.1f ms = %.3f ms per slot at %d slots" % (area, full, dur.sum() / 1e3, dur.sum() / 1e3 / full, full))
    # durations by start order: do late starters take longer / shorter?
    order = np.argsort(t0)
    q = len(order) // 4
    print("   median duration by start-time quartile: %s" % [int(np.median(dur[order[i * q:(i + 1) * q]])) for i in range(4)])
d.close()
