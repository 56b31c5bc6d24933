#!/bin/bash
# A/B of library builds inside ONE GPU session (box-to-box differences are ~3 %, larger than most kernel changes):
#   gpurun --timeout 900 -- 'LIBS="A C cur" SFS="7 10 12" bash tools/gpu_ab.sh'
# lora_sdr_amd/liblorahip_<name>.so are other builds of the library (LORAHIP_LIB, lora_sdr_amd/_lib.py); "cur" = liblorahip.so
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
one() {  # one <lib name> <label> <cmd...>
  local lib=$1 label=$2; shift 2
  local path=$R/lora_sdr_amd/liblorahip_$lib.so; [[ $lib == cur ]] && path=$R/lora_sdr_amd/liblorahip.so
  LORAHIP_LIB=$path "$@" 2>/dev/null | python -c "
import json, sys
lab = sys.argv[1]
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line); print('%-28s %8.1f Msym/s frac %.3f launch %.1f us' % (lab, d['value'], d['roofline']['frac'], d['roofline']['launch_us']))
    elif 'mode 1' in line:
        import re
        m = re.search(r'kernel ([0-9.]+) ms.*-> ([0-9.]+) Msym/s end to end', line); print('%-28s kernel %s ms, e2e %s Msym/s' % (lab, m.group(1), m.group(2)))
" "$label"
}
for rep in 1 2; do
  for sf in ${SFS:-7 10 12}; do
    case $sf in 7) CH=16384;; 8|9) CH=8192;; 10) CH=4096;; 11) CH=2048;; *) CH=1024;; esac
    for lib in ${LIBS:-A cur}; do
      [[ -n "${NOSTEADY:-}" ]] || one $lib "SF$sf steady  [$lib]" python bench.py --sf $sf --no-cpu-baseline
      [[ -n "${NOMOVING:-}" ]] || one $lib "SF$sf moving  [$lib]" python bench.py --sf $sf --no-cpu-baseline --moving
      [[ -n "${NOL3:-}" ]] || one $lib "SF$sf level3  [$lib]" python tools/bench_demod.py --sf $sf --channels $CH --modes 1 --reps 6
    done
  done
done
