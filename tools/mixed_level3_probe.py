"""bench.py's mixed_level3 section (BASELINE configs[3] as receivers behind one handle) by itself:  python tools/mixed_level3_probe.py
(LORAHIP_PART_PRIORITY=1: the parts' streams by priority, long windows first; LORA_PROBE_LANES=-1: every part on its 16-points-per-lane
geometry -- lorahip_demod_set_stream_lanes on the mixed handle -- instead of the lanes its own channel count would pick)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench
import lora_sdr_amd as L
env = bench.Env(bench.parse())
if os.environ.get("LORA_PROBE_LANES", "") != "":
    _init = L.LoRaDemod.__init__
    def _forced(self, *a, **k):
        _init(self, *a, **k)
        if k.get("channel_sf") is not None: self.set_stream_lanes(int(os.environ["LORA_PROBE_LANES"]))
    L.LoRaDemod.__init__ = _forced
for _ in range(2):
    r = bench.section_mixed_level3(env, L)
    print("mixed_level3: e2e %.3f ms, slowest part's kernel %.3f ms, %.1f Msym/s, %.4f of the byte-weighted roofline, oracle mismatches %s" % (
        r["e2e_ms"], r["kernel_ms_slowest_part"], r["Msym_s_e2e"], r["frac_byte_weighted_e2e"], r["oracle_channel_mismatches"]), flush=True)
