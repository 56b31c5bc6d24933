"""Another build of the library beside the shipped one, for A/B measurements inside one GPU session (tools/gpu_ab.sh, LORAHIP_LIB):
    python tools/build_variant.py <name> [-DFLAG ...] [--only file.hip ...]
-> lora_sdr_amd/liblorahip_<name>.so (git-ignored; travels to the GPU box). Objects under lora_sdr_amd/build_<name>/; translation
units that are not named with --only are taken from the shipped build's objects when the flags do not concern them."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lora_sdr_amd.build import FLAGS, SOURCES, CSRC, HERE, OBJDIR, build_lib
name = sys.argv[1]
extra = [a for a in sys.argv[2:] if a.startswith("-")]
only = [a for a in sys.argv[2:] if not a.startswith("-")]
build_lib()                                     # the shipped objects are current
objdir = os.path.join(HERE, "build_" + name)
os.makedirs(objdir, exist_ok=True)
jobs, objs = [], []
for s in SOURCES:
    if only and s not in only:
        objs.append(os.path.join(OBJDIR, os.path.splitext(s)[0] + ".o"))
        continue
    obj = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
    objs.append(obj)
    jobs.append((s, subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-c", os.path.join(CSRC, s), "-o", obj])))
bad = [s for s, p in jobs if p.wait() != 0]
if bad:
    raise SystemExit("hipcc failed: " + ", ".join(bad))
lib = os.path.join(HERE, "liblorahip_%s.so" % name)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib], check=True)
print(lib, "flags", " ".join(extra), "recompiled", [s for s, _ in jobs])
