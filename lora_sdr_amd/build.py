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
SOURCES = ["lorahip_kernels.hip", "lorahip_fast.hip", "lorahip_wide.hip", "lorahip_stream.hip", "lorahip_stream_lanes.hip", "lorahip_stream_pairs.hip", "lorahip_resident.hip", "lorahip_codec.hip", "lorahip_chan.hip", "lorahip_api.cpp", "lorahip_tables.cpp",
           "lorahip_demod.cpp", "lorahip_mixed.cpp", "lorahip_upload.cpp", "lorahip_rx.cpp", "lorahip_fma_fast.hip", "lorahip_fma_wide.hip"]
HEADERS = ["lorahip_internal.h", "lorahip_device.h", "lorahip_fft.h", "lorahip_fastcore.h", "lorahip_framemachine.h", "lorahip_streamkernel.h", "lorahip_residentproto.h", "lorahip_streamcfg.h", "lorahip_fine.h", os.path.join("..", "..", "include", "lorahip.h")]
INCLUDED_SOURCES = {"lorahip_fma_fast.hip": ["lorahip_fast.hip"], "lorahip_fma_wide.hip": ["lorahip_wide.hip"]}
OBJDIR = os.path.join(HERE, "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-result", "-x", "hip"]


def _mtime(path):
    return os.path.getmtime(path) if os.path.exists(path) else 0.0


def source_digest():
    """sha256 over every source and header the library is built from (what a build log entry is stamped with)"""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


DETECT_KERNEL_FILES = ["lorahip_fast.hip", "lorahip_wide.hip", "lorahip_kernels.hip", "lorahip_fastcore.h", "lorahip_fft.h", "lorahip_device.h", "lorahip_fine.h"]


def kernel_digest():
    """sha256 over the files the batch detect kernels are compiled from: what a committed counter measurement (profiles/traffic.json)
    is stamped with, so that bench.py replays it only while those kernels are the ones it times"""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(DETECT_KERNEL_FILES):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def build_lib(force=False, verbose=False, extra=()):
    """Compile every HIP/C++ source (one object per translation unit, in parallel) and link
    lora_sdr_amd/liblorahip.so; returns its path. Only stale objects are rebuilt unless force (or LORAHIP_FORCE_REBUILD=1 in the
    environment): then every object is removed first and all translation units are compiled from clean. What was done is printed
    and appended to lora_sdr_amd/build/build_log.jsonl (mode, translation units compiled, seconds, source digest)."""
    import json
    import time
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    force = bool(force) or os.environ.get("LORAHIP_FORCE_REBUILD", "") not in ("", "0")
    os.makedirs(OBJDIR, exist_ok=True)
    t_start = time.time()
    if force:
        for f in os.listdir(OBJDIR):
            if f.endswith(".o") or f.endswith(".stamp"):
                os.remove(os.path.join(OBJDIR, f))
        if os.path.exists(LIB):
            os.remove(LIB)
    # an object is current when its stamp holds the digest of its source, every header and the flags: by CONTENT, not by
    # modification time (a checkout or a copy to another box may reorder mtimes)
    import hashlib
    hh = hashlib.sha256()
    for name in sorted(HEADERS):
        with open(os.path.join(CSRC, name), "rb") as f:
            hh.update(name.encode() + b"\0" + f.read())
    hh.update(" ".join(FLAGS + list(extra)).encode())
    jobs, objs, stamps = [], [], {}
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        h1 = hh.copy()
        with open(src, "rb") as f:
            h1.update(f.read())
        for dep in INCLUDED_SOURCES.get(s, ()):             # a translation unit that #includes another one's source
            with open(os.path.join(CSRC, dep), "rb") as f:
                h1.update(f.read())
        want = h1.hexdigest()
        have = open(obj + ".stamp").read().strip() if os.path.exists(obj + ".stamp") and os.path.exists(obj) else ""
        if force or have != want:
            cmd = [hipcc] + FLAGS + list(extra) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            if os.path.exists(obj + ".stamp"):
                os.remove(obj + ".stamp")
            stamps[obj] = want
            jobs.append((s, subprocess.Popen(cmd)))
    failed = [s for s, p in jobs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for " + ", ".join(failed))
    for obj, want in stamps.items():
        with open(obj + ".stamp", "w") as f:
            f.write(want + "\n")
    linked = False
    if jobs or _mtime(LIB) < max(_mtime(o) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        linked = True
    rec = {"mode": "clean" if force else "incremental", "compiled": [s for s, _ in jobs], "of": len(SOURCES), "linked": linked,
           "seconds": round(time.time() - t_start, 1), "sources_sha16": source_digest(), "flags": FLAGS + list(extra), "time": int(t_start)}
    print("lorahip build: %s, %d of %d translation units compiled%s in %.1f s (sources %s)"
          % (rec["mode"], len(jobs), len(SOURCES), ", linked" if linked else ", library up to date", rec["seconds"], rec["sources_sha16"]), flush=True)
    try:
        with open(os.path.join(OBJDIR, "build_log.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    return LIB


HOST_SOURCES = [s_ for s_ in SOURCES if s_.endswith(".cpp")]
SANITIZERS = {"address": ("asan", ["-fsanitize=address"]), "thread": ("tsan", ["-fsanitize=thread"]), "undefined": ("ubsan", ["-fsanitize=undefined"])}


def sanitizer_runtime(kind):
    """the shared sanitizer runtime that has to be preloaded into an uninstrumented python (LD_PRELOAD): gcc's for address / undefined,
    clang's for thread (see build_sanitized)"""
    if kind == "thread":
        import glob
        hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so"))
        return hits[-1] if hits else None
    tag = {"address": "libasan.so", "undefined": "libubsan.so"}[kind]
    out = subprocess.run([os.environ.get("CXX", "g++"), "-print-file-name=" + tag], capture_output=True, text=True).stdout.strip()
    return os.path.realpath(out) if os.path.isabs(out) else None


def build_sanitized(kind, verbose=False):
    """SURVEY.md section 5's sanitizer pass: ANOTHER build of the library beside the shipped one -- lora_sdr_amd/liblorahip_<asan|tsan|ubsan>.so
    (git-ignored; loaded through LORAHIP_LIB) -- whose HOST translation units (the stateful C++ with worker threads: lorahip_demod.cpp,
    lorahip_rx.cpp, lorahip_upload.cpp, lorahip_mixed.cpp, lorahip_api.cpp, lorahip_tables.cpp) are compiled with -fsanitize=<kind>,
    -O1 -g and frame pointers; the kernel objects are the shipped build's. The host units are plain C++ over the HIP runtime API, so
    address / undefined are compiled with g++ and run against gcc's runtimes: ROCm's clang ASan runtime intercepts
    hsa_amd_memory_pool_allocate for its device-side ASan and aborts in a process that holds a GPU without xnack+ (profiles/r05/
    s7_*: "allocator is trying to allocate 0x400000 bytes" inside libamdhip64). thread is compiled with hipcc's clang and runs against
    ITS runtime: gcc 11's ThreadSanitizer does not know this kernel's address-space layout ("unexpected memory mapping"). Also reached as
    LORAHIP_SANITIZE=address|thread|undefined python -m lora_sdr_amd.build. Run with LD_PRELOAD=<sanitizer_runtime(kind)>
    LORAHIP_LIB=<the result> (tools/gpu_sanitize.sh)."""
    if kind not in SANITIZERS:
        raise ValueError("LORAHIP_SANITIZE: one of " + ", ".join(sorted(SANITIZERS)))
    build_lib(verbose=verbose)                         # the shipped objects are current
    name, sflags = SANITIZERS[kind]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cxx = os.environ.get("CXX", "g++")
    objdir = os.path.join(HERE, "build_" + name)
    os.makedirs(objdir, exist_ok=True)
    clang = kind == "thread"
    if clang:
        cxx = hipcc
        flags = [f for f in FLAGS if f != "-O3"] + ["-O1", "-g", "-fno-omit-frame-pointer", "-shared-libsan"] + sflags
    else:
        flags = ["-std=c++17", "-O1", "-g", "-fPIC", "-fno-omit-frame-pointer", "-ffp-contract=off", "-fno-fast-math", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"] + sflags
    jobs, objs = [], []
    for s in SOURCES:
        if s not in HOST_SOURCES:
            objs.append(os.path.join(OBJDIR, os.path.splitext(s)[0] + ".o"))
            continue
        obj = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        cmd = [cxx] + flags + ["-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append((s, subprocess.Popen(cmd)))
    bad = [s for s, p in jobs if p.wait() != 0]
    if bad:
        raise RuntimeError("%s failed for %s" % (cxx, ", ".join(bad)))
    lib = os.path.join(HERE, "liblorahip_%s.so" % name)
    # (the sanitizer's own symbols stay undefined in the library: the preloaded runtime provides them)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + (["-shared-libsan"] + sflags if clang else []) + objs + ["-o", lib], check=True)
    print("lorahip sanitized build (%s): %s; host units %s (%s); preload %s" % (kind, lib, ", ".join(HOST_SOURCES), "clang" if clang else "g++", sanitizer_runtime(kind)), flush=True)
    return lib


if __name__ == "__main__":
    if os.environ.get("LORAHIP_SANITIZE") or "--sanitize" in sys.argv:
        kind = os.environ.get("LORAHIP_SANITIZE") or sys.argv[sys.argv.index("--sanitize") + 1]
        print(build_sanitized(kind, verbose=True))
        sys.exit(0)
    # --all-variants: also compile the round-1 A/B kernel variants (profiling only; the shipped library carries the default,
    # the generic kernel and one alternative per SF)
    # --define NAME: a measurement build (e.g. LORAHIP_RESIDENT_STAMPS: the resident receiver's per-wavefront time stamps); the shipped
    # library is the one built WITHOUT it -- build it again afterwards
    extra = ["-DLORAHIP_ALL_VARIANTS"] if "--all-variants" in sys.argv else []
    extra += ["-D" + sys.argv[i + 1] for i, a in enumerate(sys.argv[:-1]) if a == "--define"]
    build_lib(force="--force" in sys.argv, verbose=True, extra=tuple(extra))
    print(LIB)
