"""Summarise rocprofv3 --pmc passes found under a directory: per kernel, per counter, mean value per dispatch.

    python tools/pmc_summary.py gpurun_out

FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; on gfx950 FETCH_SIZE counts a wide coalesced
streaming read at half its bytes (MI355X_MICROARCH.md, HBM section) -- the corrected figure is printed beside it.
"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(list)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")
                if "lorahip::detect" not in k:
                    continue
                acc[(k[:60], row.get("Counter_Name", "?"))].append(float(row.get("Counter_Value", "nan")))
    print("==", os.path.basename(d))
    for (k, c), v in sorted(acc.items()):
        mean = sum(v) / len(v)
        extra = ""
        if c == "FETCH_SIZE":
            extra = "  -> %.1f MB/dispatch as counted, %.1f MB with the gfx950 x2 correction" % (mean / 1024, 2 * mean / 1024)
        elif c == "WRITE_SIZE":
            extra = "  -> %.3f MB/dispatch as counted" % (mean / 1024)
        print("  %-60s %-12s n=%d mean=%.1f%s" % (k, c, len(v), mean, extra))


# ---- HBM traffic per launch of the detect kernel, for bench.py's roofline.traffic -----------------------
# FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 counts a wide coalesced streaming read at half
# its bytes, MI355X_MICROARCH.md "HBM"); WRITE_SIZE is taken as counted (uncalibrated for 2..4-byte scattered stores).
import json
import re
traffic = {}
for d in sorted(glob.glob(os.path.join(root, "pmc_*_sf*"))):
    m = re.match(r"pmc_(FETCH_SIZE|WRITE_SIZE)_sf(\d+)$", os.path.basename(d))
    if not m:
        continue
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if "lorahip::detect" in row.get("Kernel_Name", "") and row.get("Counter_Name") == m.group(1):
                vals.append(float(row["Counter_Value"]))
    if vals:
        kb = sum(vals) / len(vals)
        t = traffic.setdefault(m.group(2), {})
        if m.group(1) == "FETCH_SIZE":
            t["fetch_bytes"] = 2 * kb * 1024
        else:
            t["write_bytes"] = kb * 1024
for sf, t in traffic.items():
    if "fetch_bytes" in t and "write_bytes" in t:
        t["total_bytes"] = t["fetch_bytes"] + t["write_bytes"]
if traffic:
    out = os.path.join(root, "traffic.json")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from lora_sdr_amd.build import kernel_digest
    # stamped with the digest of the files the detect kernels are compiled from: bench.py replays these numbers only for that build
    json.dump({"note": "HBM bytes per launch of the detect kernel at bench.py's default geometry; rocprofv3 --pmc FETCH_SIZE (x2, gfx950) and WRITE_SIZE in separate passes",
               "sources_sha16": kernel_digest(), "session": os.environ.get("TAG", "?"),
               "per_sf": traffic}, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out)
