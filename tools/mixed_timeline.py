"""The six parts of bench.py's mixed_level3 pass on one device, side by side: from a rocprofv3 --kernel-trace of tools/mixed_level3_probe.py,
the streaming kernels of the LAST timed pass -- when each started and ended relative to the first one, and how many wavefront slots were
still wanted when the short ones had left (the pass is slot-bound, the end is the drain of the SF12 part).
    cd /tmp && rocprofv3 --kernel-trace -d /tmp/mx -o mx --output-format csv -- python $R/tools/mixed_level3_probe.py; python tools/mixed_timeline.py /tmp/mx"""
import csv, glob, os, re, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        k = r.get("Kernel_Name", "")
        if "demodStream" not in k:
            continue
        m = re.search(r"Cfg<(\d+)", k)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(m.group(1)) if m else 0, "Wide" in k, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort()
if not rows:
    raise SystemExit("no demodStream kernels in the trace")
# the k-th launch of every SF belongs to pass k (a pass launches one kernel per part)
from collections import defaultdict
by = defaultdict(list)
for r in rows: by[r[2]].append(r)
n = min(len(v) for v in by.values())
print("%d streaming launches, %d SFs, %d passes" % (len(rows), len(by), n))
for k in range(max(0, n - 3), n):
    p = [by[sf][len(by[sf]) - n + k] for sf in sorted(by)]
    t0 = min(x[0] for x in p); t1 = max(x[1] for x in p)
    print("pass %d of %.3f ms:" % (k, (t1 - t0) / 1e6))
    for s_, e, sf, wide, wg, grid in p:
        print("   SF%-2d %s workgroups %5d x %3d threads: start +%.3f ms, end +%.3f ms (%.3f ms)" % (sf, "wide" if wide else "    ", grid // max(wg, 1), wg, (s_ - t0) / 1e6, (e - t0) / 1e6, (e - s_) / 1e6))
