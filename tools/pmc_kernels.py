"""Per-kernel means of rocprofv3 --pmc passes: every counter of every pass directory matching a glob, for kernels whose
name contains a pattern.   python tools/pmc_kernels.py 'gpurun_out/l2_mov_sf7_*' detect [label]"""
import csv, glob, os, sys
from collections import defaultdict

pat, kern = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else pat
acc = defaultdict(list)
names = set()
for d in sorted(glob.glob(pat)):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            k = row.get("Kernel_Name", "?")
            if kern not in k:
                continue
            names.add(k.split("(")[0][:90])
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== %s   kernel(s): %s" % (label, "; ".join(sorted(names)) or "none matched '%s'" % kern))
for c in sorted(acc):
    v = acc[c]
    print("  %-40s n=%-3d mean per dispatch %16.1f" % (c, len(v), sum(v) / len(v)))
m = {c: sum(v) / len(v) for c, v in acc.items()}
def ratio(a, b, text):
    if a in m and b in m and m[b]:
        print("  -> %s = %.4f" % (text, m[a] / m[b]))
ratio("TCC_HIT_sum", "TCC_REQ_sum", "L2 hit / request")
ratio("TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "L1 -> L2 read requests per L1 cache access (L1 miss ratio)")
ratio("TCP_TOTAL_CACHE_ACCESSES_sum", "SQ_INSTS_VMEM_RD", "L1 cache-line accesses per vector-memory read instruction")
ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "wave cycles parked on s_waitcnt / barrier")
ratio("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "wave cycles issuing VALU")
