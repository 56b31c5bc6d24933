"""Readable register / scratch table of the kernels of a translation unit (after tools/kernel_resources.py left its .s in /tmp):
    python tools/kernel_table.py lorahip_wide.hip [name substring]"""
import re, subprocess, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
f = sys.argv[1]; key = sys.argv[2] if len(sys.argv) > 2 else ""
subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "kernel_resources.py"), f] + [a for a in sys.argv[3:]], capture_output=True)
txt = open("/tmp/_kr_%s.s" % f).read()
meta = txt[txt.index("amdhsa.kernels:"):]
worst = 0
for rec in meta.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", rec) or [None, "?"])[1]
    dem = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip().replace("lorahip::", "").replace("void ", "")
    if key not in dem: continue
    m = re.match(r"(\w+)<(\w+)<([^>]*)>(.*)>\(", dem)
    if not m:
        print("%-60s vgpr %3s scratch %4s" % (dem[:60], g("vgpr_count"), g("private_segment_fixed_size"))); continue
    cfg = m.group(3).split(", ")
    sc = int(g("private_segment_fixed_size")); worst = max(worst, sc)
    print("%-16s sf%-3s cfg[%s] flags%-14s vgpr %3s sgpr %3s scratch %4d B" % (m.group(1), cfg[0], " ".join(cfg[1:]), m.group(4), g("vgpr_count"), g("sgpr_count"), sc))
print("largest scratch:", worst, "B")
