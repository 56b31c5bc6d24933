"""Average duration of the LAST k detect-kernel launches in a rocprofv3 kernel trace (= bench.py's timed steps;
the launches before them are the clock-ramp and warm-up ones).  python tools/trace_tail.py <dir> <k>"""
import csv, glob, os, sys
d, k = sys.argv[1], int(sys.argv[2])
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "lorahip::detect" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
tail = dur[-k:]
print("%s: %d launches traced; last %d (timed steps): avg %.2f us, min %.2f, max %.2f; all launches avg %.2f us"
      % (rows[-1]["Kernel_Name"][:70], len(dur), len(tail), sum(tail) / len(tail), min(tail), max(tail), sum(dur) / len(dur)))
