#!/bin/bash
# quick A/B session: moving shape every SF + the streaming kernel at three SFs + the fine-path parity tests
#   gpurun --timeout 900 -- 'TAG=s15 bash tools/gpu_quick.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; TAG=${TAG:-q}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_demod.py -x -q -m gpu 2>&1 | tail -3
for sf in ${SFS:-7 8 9 10 11 12}; do
  timeout 200 python bench.py --sf $sf --no-cpu-baseline --moving > $O/${TAG}_moving_sf$sf.json 2> $O/${TAG}_moving_sf$sf.err
  python - $O/${TAG}_moving_sf$sf.json $sf <<'EOP'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("SF%s moving %8.1f Msym/s frac %.3f launch %.1f us" % (sys.argv[2], d["value"], d["roofline"]["frac"], d["roofline"]["launch_us"]))
except Exception as e:
    print("SF", sys.argv[2], "FAILED", e)
EOP
done
for sf in ${L3SFS:-7 10 12}; do
  case $sf in 7) CH=16384;; 8|9) CH=8192;; 10) CH=4096;; *) CH=1024;; esac
  timeout 200 python tools/bench_demod.py --sf $sf --channels $CH --modes 1 > $O/${TAG}_level3_sf$sf.txt 2>&1
  tail -1 $O/${TAG}_level3_sf$sf.txt
done
if [[ -n "${STEADY:-}" ]]; then
  for sf in $STEADY; do
    timeout 200 python bench.py --sf $sf --no-cpu-baseline > $O/${TAG}_steady_sf$sf.json 2> $O/${TAG}_steady_sf$sf.err
    python - $O/${TAG}_steady_sf$sf.json $sf <<'EOP'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("SF%s steady %8.1f Msym/s frac %.3f launch %.1f us" % (sys.argv[2], d["value"], d["roofline"]["frac"], d["roofline"]["launch_us"]))
except Exception as e:
    print("SF", sys.argv[2], "FAILED", e)
EOP
  done
fi
