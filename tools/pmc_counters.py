"""profiles/counters.json from the SQ passes of `tools/gpu_r06.sh counters`: per shape (steady / moving) and SF the mean of every counter per launch
of the detect kernel and the utilisation figures bench.py replays in `roofline` (north_star: "LDS/VALU utilisation ... against gfx950 peak").

    python tools/pmc_counters.py gpurun_out [TAG]

Derived figures (all from counters of ONE pass, so that clock differences between passes cancel):
  valu_busy    share of the SIMDs' cycles in which a VALU instruction issues = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x SQ_WAVES / SIMDS.
               The batch kernels run persistent grids (every wavefront lives for the whole launch: SQ_WAVES = 3 x 1024 at SF7), so
               SQ_WAVE_CYCLES / SQ_WAVES is the launch's length in the counters' own unit (quad-cycles, MI355X_MICROARCH.md) and
               SQ_WAVES / SIMDS the wavefronts sharing a SIMD. SIMDS = 1024 (256 CUs x 4).
  wait_share   SQ_WAIT_ANY / SQ_WAVE_CYCLES: wave cycles parked on s_waitcnt / barriers
  lds_busy     SQ_LDS_IDX_ACTIVE / (CUS x launch cycles), launch cycles = 4 x SQ_WAVE_CYCLES / SQ_WAVES taken from the OTHER pass of the
               same shape (SQ_LDS_IDX_ACTIVE counts LDS-array cycles per CU; CUS = 256)
  lds_conflict SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  instr_per_sample   vector lane-instructions per IQ sample = SQ_INSTS_VALU x 64 / (windows x 2^SF)
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
tag = sys.argv[2] if len(sys.argv) > 2 else os.environ.get("TAG", "?")
SIMDS, CUS = 1024, 256
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lora_sdr_amd import workloads as WL
from lora_sdr_amd.build import kernel_digest

shapes = defaultdict(lambda: defaultdict(list))
kernels = {}
for d in sorted(glob.glob(os.path.join(root, "pmc_SQ*_*_sf*"))):
    m = re.match(r"pmc_SQ\w_(steady|moving)_sf(\d+)$", os.path.basename(d))
    if not m:
        continue
    key = (m.group(1), int(m.group(2)))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            k = row.get("Kernel_Name", "")
            if "lorahip::detect" not in k:
                continue
            kernels[key] = k.split("(")[0][:80]
            shapes[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"note": __doc__.split("Derived figures")[1].strip(), "sources_sha16": kernel_digest(), "session": tag, "simds": SIMDS, "cus": CUS,
       "command": "rocprofv3 --pmc <8 SQ counters> -- python bench.py --sf S [--moving] --steps 3 --warmup 1 --ramp-seconds 0 --no-cpu-baseline (two passes per shape)",
       "steady": {}, "moving": {}}
for (shape, sf), c in sorted(shapes.items()):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    B, S = WL.default_geometry(sf)
    W = B * S
    e = {"kernel": kernels.get((shape, sf)), "launches_averaged": min(len(v) for v in c.values()), "windows_per_launch": W,
         "counters": {k: round(v, 1) for k, v in sorted(m.items())}}
    wc, wv = m.get("SQ_WAVE_CYCLES"), m.get("SQ_WAVES")
    if wc and wv:
        e["waves_per_simd"] = round(wv / SIMDS, 3)
        if "SQ_ACTIVE_INST_VALU" in m:
            e["valu_busy"] = round(m["SQ_ACTIVE_INST_VALU"] / wc * wv / SIMDS, 4)
        if "SQ_ACTIVE_INST_LDS" in m:
            e["lds_issue_busy"] = round(m["SQ_ACTIVE_INST_LDS"] / wc * wv / SIMDS, 4)
        if "SQ_WAIT_ANY" in m:
            e["wait_share"] = round(m["SQ_WAIT_ANY"] / wc, 4)
        if "SQ_LDS_IDX_ACTIVE" in m:
            e["lds_busy"] = round(m["SQ_LDS_IDX_ACTIVE"] / (CUS * 4.0 * wc / wv), 4)
    if m.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 4)
    if "SQ_INSTS_VALU" in m:
        e["instr_per_sample"] = round(m["SQ_INSTS_VALU"] * 64.0 / (W * (1 << sf)), 3)
        e["valu_wave_instr_per_launch"] = round(m["SQ_INSTS_VALU"])
    out[shape][str(sf)] = e
    print("%-6s SF%-2d valu_busy %s wait %s lds_busy %s lds_conflict %s instr/sample %s  (%s)" % (
        shape, sf, e.get("valu_busy"), e.get("wait_share"), e.get("lds_busy"), e.get("lds_conflict"), e.get("instr_per_sample"), (e["kernel"] or "")[:50]))
if out["steady"] or out["moving"]:
    json.dump(out, open(os.path.join(root, "counters.json"), "w"), indent=1, sort_keys=True)
    print("wrote", os.path.join(root, "counters.json"))
else:
    print("no pmc_SQ*_<shape>_sf<N> directories under", root)
