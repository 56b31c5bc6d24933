#!/bin/bash
# Steady-state and per-window-settings variants again, now that the wavefronts rotate the priority (an --all-variants build):
#   python tools/build_variant.py allv -DLORAHIP_ALL_VARIANTS lorahip_fast.hip lorahip_wide.hip
#   gpurun -- 'bash tools/retune_r04.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export LORAHIP_LIB=$R/lora_sdr_amd/liblorahip_allv.so
vars() { case $1 in 7) echo "0 4 15 17 3";; 8) echo "0 15 17 30";; 9) echo "0 11 12 16 24";; 10) echo "0 15 16 30";; 11) echo "0 6 7 12 30";; *) echo "0 6 7 14 30";; esac; }
for sf in ${SFS:-7 8 9 10 11 12}; do
  for v in $(vars $sf); do
    for shape in "" ${MOVING:+--moving}; do
      timeout 120 python bench.py --sf $sf --variant $v --no-cpu-baseline $shape > $O/retune.json 2> $O/retune.err
      python - $sf $v "$shape" <<'EOP'
import json, sys
try:
    d = json.loads(open("/root/repo/gpurun_out/retune.json").read().strip().splitlines()[-1])
    print("SF%s variant %-3s %-9s %8.1f Msym/s frac %.4f launch %.1f us  index mismatches %s" % (sys.argv[1], sys.argv[2], sys.argv[3] or "steady", d["value"], d["roofline"]["frac"], d["roofline"]["launch_us"], d["oracle"]["index_mismatches"]))
except Exception as e:
    print("SF", sys.argv[1], "variant", sys.argv[2], sys.argv[3], "FAILED", e)
EOP
    done
  done
done
