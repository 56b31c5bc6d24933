#!/bin/bash
# SURVEY.md section 5's sanitizer pass on the GPU box: the host side of the library (the stateful C++ with worker threads) built with
# -fsanitize=address / thread / undefined (python -m lora_sdr_amd.build --sanitize <kind>, beside the shipped library), the level-3 /
# upload / mixed / drop-in tests and a randomised soak run under each, the reports collected.
#   gpurun --timeout 1500 -- 'TAG=s9 KINDS="address thread undefined" SOAK=40 bash tools/gpu_sanitize.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; TAG=${TAG:-san}
export TMPDIR=/tmp
TESTS=${TESTS:-"tests/test_gpu_receiver.py tests/test_gpu_dropin.py tests/test_gpu_upload.py tests/test_gpu_mixed.py tests/test_gpu_lanes.py"}
for kind in ${KINDS:-address thread}; do
  case $kind in address) n=asan;; thread) n=tsan;; *) n=ubsan;; esac
  lib=$R/lora_sdr_amd/liblorahip_$n.so
  rt=$(python -c "from lora_sdr_amd.build import sanitizer_runtime; print(sanitizer_runtime('$kind'))")
  [[ -f $lib ]] || { echo "$kind: $lib not built"; continue; }
  # (a dlopen that goes through a sanitizer's interceptor no longer searches the RUNPATH of the library that asked: torch's lazily
  # loaded libraries are found through LD_LIBRARY_PATH instead)
  export LD_LIBRARY_PATH=$(python -c "import os, torch; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))" 2>/dev/null):/opt/rocm/lib:${LD_LIBRARY_PATH:-}
  log=$O/${TAG}_${n}
  # alloc_dealloc_mismatch=0: uninstrumented C++ libraries in the process (torch, the kernel objects) bind operator new / delete and
  # malloc / free to different providers once a sanitizer runtime is preloaded -- reports about THEM, not about the host units
  export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0:halt_on_error=0:alloc_dealloc_mismatch=0:log_path=${log}_report
  export TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0:suppressions=$R/tools/tsan.supp:log_path=${log}_report
  export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=${log}_report
  rm -f ${log}_report.*
  # the compiled drop-in (oracle/_ref/libloradrop.so) asks for "liblorahip.so": give it the sanitized build too
  mkdir -p /tmp/san_$n && ln -sf $lib /tmp/san_$n/liblorahip.so && export LD_LIBRARY_PATH=/tmp/san_$n:$LD_LIBRARY_PATH
  # a preloaded runtime resolves __cxa_throw when the process starts, and python has no C++ runtime then: preload that too (the
  # compiled Pothos block throws on bad arguments, and the tests ask it to)
  rt="$rt $(g++ -print-file-name=libstdc++.so.6)"
  RUN=""
  echo "== $kind: tests"
  LD_PRELOAD=$rt LORAHIP_LIB=$lib timeout ${SAN_TEST_TIMEOUT:-700} $RUN python -m pytest $TESTS -m gpu -q ${SAN_PYTEST_ARGS:--x} -p no:cacheprovider > ${log}_pytest.txt 2>&1
  echo "exit $?" >> ${log}_pytest.txt; tail -4 ${log}_pytest.txt
  echo "== $kind: soak"
  LD_PRELOAD=$rt LORAHIP_LIB=$lib timeout $(( ${SOAK:-40} + 200 )) $RUN python tools/soak_level3.py ${SOAK:-40} ${SOAK_SEED:-9000} > ${log}_soak.txt 2>&1
  echo "exit $?" >> ${log}_soak.txt; tail -3 ${log}_soak.txt
  # the reports (one file per process that had something to say)
  ls ${log}_report.* > /dev/null 2>&1 && { for f in ${log}_report.*; do echo "--- $f"; grep -E "ERROR|WARNING|SUMMARY|runtime error" $f | sort | uniq -c | sort -rn | head -20; done; } || echo "$kind: no sanitizer report files (nothing reported)"
done
