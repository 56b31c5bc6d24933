"""Register / LDS / scratch use of the kernels of one translation unit (device-only assembly's AMDGPU metadata).
    python tools/kernel_resources.py lorahip_stream.hip [name substring] [extra hipcc flags...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lora_sdr_amd.build import FLAGS, CSRC
src = os.path.join(CSRC, sys.argv[1])
key = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
out = "/tmp/_kr_%s.s" % os.path.basename(src)
subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["--cuda-device-only", "-S", src, "-o", out], check=True)
txt = open(out).read()
meta = txt[txt.index("amdhsa.kernels:"):]
for rec in meta.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", rec) or [None, "?"])[1]
    dem = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    if key and key not in dem:
        continue
    dem = re.sub(r"lorahip::", "", dem)
    print("%-120s vgpr %3s agpr %3s sgpr %3s scratch %5s B" % (dem[:120], g("vgpr_count"), rec.split("\n")[0].strip(), g("sgpr_count"), g("private_segment_fixed_size")))
