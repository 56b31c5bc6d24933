"""CPU: the C-ABI shared library loads, exports every symbol include/lorahip.h declares, its
host-side tables are bit-identical to the oracle's, and it fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, bits

import lora_sdr_amd as L
from lora_sdr_amd import _lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lorahip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lorahip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "liblorahip.so does not export %s" % n
    # and the Python binding knows all of them
    assert sorted(_lib.SIGNATURES) == names


def test_struct_layout_matches_header():
    # struct lorahip_batch: 16 pointer-sized fields incl. one int32 padded to 8
    assert C.sizeof(_lib.Batch) == 16 * 8
    assert C.sizeof(_lib.WorkResult) == 72
    assert C.sizeof(_lib.DemodPorts) == 8 * 8


def test_version_and_errors():
    lib = L.load()
    assert lib.lorahip_version() == 4
    assert lib.lorahip_selfcheck() == 0, lib.lorahip_last_error()     # every kernel's LDS exchange layout is injective
    assert lib.lorahip_strerror(0) == b"ok"
    assert lib.lorahip_strerror(-5) == b"device is not gfx950"
    assert lib.lorahip_host_tables(0, None, None, None, None) == -1
    assert lib.lorahip_host_tables(13, None, None, None, None) == -1
    assert lib.lorahip_create(None, 0, 7) == -1
    assert lib.lorahip_detect_batch(None, None) == -1
    # level 3 without an object: an error code or a neutral value, never a crash (no compute call without a GPU)
    assert lib.lorahip_demod_run_device_segments(None, None, None, None, None) == -1
    assert lib.lorahip_demod_consumed_all(None, None) == -1
    assert lib.lorahip_demod_last_launches(None) == 0
    assert lib.lorahip_demod_kernel_ms(None) == 0.0


def test_round4_entries_refuse_bad_arguments_and_fail_loudly_without_a_gpu():
    """lorahip_demod_create_mixed / _receive / _run_device_append / signals / part accessors: NULL and out-of-range arguments are
    LORAHIP_E_INVALID (never a crash), and without a gfx950 device the mixed object cannot be made (no CPU path behind it)"""
    import torch
    lib = L.load()
    h = C.c_void_p()
    sfs = np.array([7, 8, 9], np.int32)
    dev = np.array([0], np.int32)
    assert lib.lorahip_demod_create_mixed(None, dev.ctypes.data, 1, sfs.ctypes.data, 3) == -1
    assert lib.lorahip_demod_create_mixed(C.byref(h), None, 1, sfs.ctypes.data, 3) == -1
    assert lib.lorahip_demod_create_mixed(C.byref(h), dev.ctypes.data, 0, sfs.ctypes.data, 3) == -1
    assert lib.lorahip_demod_create_mixed(C.byref(h), dev.ctypes.data, 1, sfs.ctypes.data, 0) == -1
    bad = np.array([7, 13], np.int32)
    assert lib.lorahip_demod_create_mixed(C.byref(h), dev.ctypes.data, 1, bad.ctypes.data, 2) == -1 and not h.value
    if not torch.cuda.is_available():
        rc = lib.lorahip_demod_create_mixed(C.byref(h), dev.ctypes.data, 1, sfs.ctypes.data, 3)
        assert rc in (-2, -5) and not h.value, rc                  # LORAHIP_E_NODEVICE / _ARCH: loud, nothing half-made
    assert lib.lorahip_demod_receive(None, None, 0, 0, None, None, None) == -1
    assert lib.lorahip_demod_run_device_append(None, None, 0, 0, None) == -1
    assert lib.lorahip_demod_rewind(None) == -1
    assert lib.lorahip_demod_set_signals(None, 1) == -1
    assert lib.lorahip_demod_num_signals(None) == 0
    assert lib.lorahip_demod_num_parts(None) == 0 and lib.lorahip_demod_num_channels(None) == 0
    assert lib.lorahip_demod_part(None, 0, None, None, None, None) == -1
    assert lib.lorahip_demod_part_handle(None, 0) is None
    assert lib.lorahip_demod_set_variant(None, 0) == -1
    assert lib.lorahip_demod_set_stream_grid(None, 0) == -1
    assert lib.lorahip_set_variant(None, 40) == -1
    assert C.sizeof(_lib.PacketRows) == 7 * 8          # struct lorahip_packet_rows: 6 pointer-sized fields + 2 x int32


@pytest.mark.parametrize("sf", range(6, 13))
def test_host_tables_match_oracle(oracle, sf):
    up, down, fine, tw = L.host_tables(sf)
    ou, od, of = oracle.tables(sf)
    ot = oracle.twiddles(1 << sf)
    assert np.array_equal(bits(up), bits(ou))
    assert np.array_equal(bits(down), bits(od))
    assert np.array_equal(bits(fine), bits(of))
    assert np.array_equal(bits(tw), bits(ot))
    assert fine[0] != 1.0          # accumulator is pre-incremented (LoRaDemod.cpp:111): fine[0] != 1


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert L.device_count() == 0
    with pytest.raises(L.LoraHipError):
        L.Context(7)
    with pytest.raises(L.LoraHipError):
        L.LoRaDetector(1024)
    with pytest.raises(L.LoraHipError):
        L.LoRaDemod(10)
    with pytest.raises(L.LoraHipError):
        L.MixedDetector([7, 8, 12])
    with pytest.raises(L.LoraHipError):
        L.MixedDetectorMulti([7, 8, 12], [0, 0])
    with pytest.raises(MemoryError):                  # pinned memory comes from the HIP runtime: none without a device
        L.pinned_empty((16,), np.complex64)


def test_product_never_touches_oracle():
    """the product package must not import / link / call anything under oracle/"""
    pkg = os.path.join(ROOT, "lora_sdr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no CPU", ""), "%s mentions oracle" % f
                assert "/root/reference" not in src


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_c_shard_plan_equals_shard_py(world):
    """lorahip_shard_plan (what lorahip_mixed_create_multi splits the channels over the devices of one process with, SURVEY.md
    section 8e) is the rule of lora_sdr_amd/shard.py (the one-process-per-GPU form): BASELINE configs[3], ragged buckets, one SF,
    fewer channels than shards. Host-only: no GPU needed."""
    from lora_sdr_amd.shard import shard_channels
    rng = np.random.default_rng(world)
    cases = [7 + np.arange(16384) % 6, np.array([12, 7, 7]), np.full(1000, 10), rng.integers(6, 13, 777), np.array([9]),
             np.concatenate([np.full(5, 7), np.full(3, 12)])]
    for sfs in cases:
        plan = L.shard_plan(sfs, world)
        parts = shard_channels(sfs, world)
        want = np.empty(len(sfs), np.int32)
        for r, p in enumerate(parts):
            want[p] = r
        assert np.array_equal(plan, want)
    assert L.shard_plan([], world).size == 0
    lib = L.load()
    assert lib.lorahip_shard_plan(None, 0, 0, None) == -1                 # no shards
    bad = np.array([7, 0], np.int32); out = np.zeros(2, np.int32)
    assert lib.lorahip_shard_plan(bad.ctypes.data, 2, 2, out.ctypes.data) == -1
    assert lib.lorahip_mixed_create_multi(None, None, 0, None, 0) == -1


@pytest.mark.gpu
def test_cpp_detector_shim_detects_on_the_gpu(gpu, tmp_path):
    """the success branch of the test below, in the -m gpu set: the C++ shim class finds the DC tone in bin 0"""
    assert gpu.cuda.is_available()
    test_cpp_detector_shim_compiles_links_and_fails_loudly(tmp_path)


def test_cpp_detector_shim_compiles_links_and_fails_loudly(tmp_path):
    """include/LoRaDetectorHip.hpp (the LoRaDetector<float> drop-in of INTEGRATION.md §1) builds with a
    plain host compiler against the C ABI; without a gfx950 device its constructor throws."""
    import shutil
    import subprocess
    import torch
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no g++")
    src = tmp_path / "shim.cpp"
    src.write_text(r'''
#include "LoRaDetectorHip.hpp"
#include <cstdio>
int main()
{
    try {
        LoRaDetectorHip<float> det(1024);
        for (size_t i = 0; i < 1024; i++) det.feed(i, std::complex<float>(1.0f, 0.0f));   // DC tone -> bin 0
        float power, powerAvg, fIndex;
        const size_t idx = det.detect(power, powerAvg, fIndex);
        std::printf("index %zu power %g\n", idx, power);
        return idx == 0 ? 0 : 2;
    } catch (const std::exception &e) { std::printf("threw: %s\n", e.what()); return 3; }
}
''')
    exe = tmp_path / "shim"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run([cxx, "-std=c++11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-llorahip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        assert r.returncode == 3 and "threw" in r.stdout, r.stdout + r.stderr


def test_header_is_plain_c_and_a_c_program_links(tmp_path):
    """include/lorahip.h is a C header: a C99 translation unit (gcc -std=c99 -pedantic -Werror) that touches every level of the ABI --
    the round-5 entry points included -- compiles, links against liblorahip.so and, without a GPU, is told so by return codes (no
    exception, no abort crosses the boundary)."""
    import shutil
    import subprocess
    import torch
    cc = shutil.which("gcc")
    if cc is None:
        pytest.skip("no gcc")
    src = tmp_path / "cabi.c"
    src.write_text(r'''
#include "lorahip.h"
#include <stdio.h>
#include <string.h>
int main(void)
{
    lorahip_ctx *ctx = NULL;
    lorahip_demod *d = NULL;
    lorahip_decoder_cfg cfg;
    lorahip_packet_rows rows;
    int64_t first[2] = { 0, 0 };
    size_t count[2] = { 0, 0 };
    uint16_t syms[16]; int32_t nsyms[1] = { 16 }, out_len[1], dropped[1];
    uint8_t out[48];
    int rc;
    memset(&cfg, 0, sizeof cfg); cfg.struct_size = sizeof cfg; cfg.sf = 7; cfg.rdd = 4; cfg.interleaving = 1; cfg.explicit_hdr = 1; cfg.data_length = 8;
    memset(&rows, 0, sizeof rows); rows.struct_size = sizeof rows;
    memset(syms, 0, sizeof syms);
    if (lorahip_version() != 4) return 10;
    rc = lorahip_create(&ctx, 0, 7);
    printf("create %d (%s)\n", rc, lorahip_strerror(rc));
    if (rc != LORAHIP_OK)
    {
        /* no device: every entry refuses politely */
        if (lorahip_decode_packets_host(NULL, &cfg, syms, 16, nsyms, 1, out, 48, out_len, dropped) != LORAHIP_E_INVALID) return 11;
        if (lorahip_demod_run_host_rows(NULL, NULL, 0, first, count, NULL) != LORAHIP_E_INVALID) return 12;
        if (lorahip_demod_stream_wait(NULL, NULL) != LORAHIP_E_INVALID || lorahip_demod_stream_follow(NULL, NULL) != LORAHIP_E_INVALID) return 13;
        if (lorahip_demod_set_stream_lanes(NULL, 0) != LORAHIP_E_INVALID || lorahip_demod_set_record_capacity(NULL, 0) != LORAHIP_E_INVALID) return 14;
        return rc == LORAHIP_E_NODEVICE ? 3 : 4;
    }
    rc = lorahip_demod_create(&d, 0, 7, 2);
    if (rc != LORAHIP_OK) return 20;
    if (lorahip_demod_stream_lanes(d) < 3) return 21;
    if (lorahip_demod_run_host_rows(d, NULL, 0, first, count, NULL) != LORAHIP_OK) return 22;       /* two channels with nothing: a run over nothing */
    if (lorahip_decode_packets_host(ctx, &cfg, syms, 16, nsyms, 1, out, 48, out_len, dropped) != LORAHIP_OK) return 23;
    printf("decoded %d dropped %d lanes 2^%d max symbols %d\n", (int)out_len[0], (int)dropped[0], lorahip_demod_stream_lanes(d), lorahip_decode_max_symbols());
    lorahip_demod_destroy(d);
    lorahip_destroy(ctx);
    return 0;
}
''')
    exe = tmp_path / "cabi"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run([cc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-llorahip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        assert r.returncode == 3, (r.returncode, r.stdout + r.stderr)


@pytest.mark.gpu
def test_c_program_runs_on_the_gpu(gpu, tmp_path):
    """the success branch of the test above, in the -m gpu set"""
    assert gpu.cuda.is_available()
    test_header_is_plain_c_and_a_c_program_links(tmp_path)
