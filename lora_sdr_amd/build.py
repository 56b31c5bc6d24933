"""Build liblorahip.so in-tree with hipcc for gfx950 (no GPU needed: cross-compiles).

    python -m lora_sdr_amd.build        # or: from lora_sdr_amd.build import build_lib

The whole library is compiled with -ffp-contract=off: the FFT kernels promise the exact
fp32 operation graph of the reference (no FMA), and the host-side tables must round like
the reference's.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblorahip.so")
SOURCES = ["lorahip_kernels.hip", "lorahip_fast.hip", "lorahip_wide.hip", "lorahip_stream.hip", "lorahip_codec.hip", "lorahip_chan.hip", "lorahip_api.cpp", "lorahip_tables.cpp",
           "lorahip_demod.cpp", "lorahip_mixed.cpp", "lorahip_upload.cpp"]
HEADERS = ["lorahip_internal.h", "lorahip_device.h", "lorahip_fft.h", "lorahip_fastcore.h", "lorahip_framemachine.h", "lorahip_fine.h", os.path.join("..", "..", "include", "lorahip.h")]
OBJDIR = os.path.join(HERE, "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-result", "-x", "hip"]


def _mtime(path):
    return os.path.getmtime(path) if os.path.exists(path) else 0.0


def build_lib(force=False, verbose=False, extra=()):
    """Compile every HIP/C++ source (one object per translation unit, in parallel) and link
    lora_sdr_amd/liblorahip.so; returns its path. Only stale objects are rebuilt."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_t = max([_mtime(os.path.join(CSRC, h)) for h in HEADERS] + [_mtime(os.path.abspath(__file__))])
    jobs, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or extra or _mtime(obj) < max(_mtime(src), hdr_t):
            cmd = [hipcc] + FLAGS + list(extra) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((s, subprocess.Popen(cmd)))
    failed = [s for s, p in jobs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for " + ", ".join(failed))
    if jobs or _mtime(LIB) < max(_mtime(o) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    # --all-variants: also compile the round-1 A/B kernel variants (profiling only; the shipped library carries the default,
    # the generic kernel and one alternative per SF)
    build_lib(force="--force" in sys.argv, verbose=True, extra=("-DLORAHIP_ALL_VARIANTS",) if "--all-variants" in sys.argv else ())
    print(LIB)
