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
SOURCES = ["lorahip_kernels.hip", "lorahip_fast.hip", "lorahip_api.cpp", "lorahip_tables.cpp", "lorahip_demod.cpp"]
HEADERS = ["lorahip_internal.h", "lorahip_device.h", os.path.join("..", "..", "include", "lorahip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-result", "-x", "hip"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False, extra=()):
    """Compile every HIP/C++ source into lora_sdr_amd/liblorahip.so; returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + list(extra) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv, verbose=True)
    print(LIB)
