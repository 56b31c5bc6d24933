#!/bin/bash
# round-4 GPU sessions: gpurun --timeout N -- 'TAG=s2 bash tools/gpu_r04.sh tests level3 receive ...'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; TAG=${TAG:-r04}
export TMPDIR=/tmp
what="$*"
l3ch() { case $1 in 6|7) echo 16384;; 8|9) echo 8192;; 10) echo 4096;; 11) echo 2048;; *) echo 1024;; esac; }
if [[ $what == *tests* ]]; then
  timeout 600 python -m pytest ${TESTS:-tests} -m gpu -x -q > $O/${TAG}_pytest.txt 2>&1; tail -4 $O/${TAG}_pytest.txt
fi
if [[ $what == *level3* ]]; then
  for sf in ${L3SFS:-7 8 9 10 11 12}; do
    timeout 200 python tools/bench_demod.py --sf $sf --channels $(l3ch $sf) --modes 1 > $O/${TAG}_level3_sf$sf.txt 2>&1
    tail -1 $O/${TAG}_level3_sf$sf.txt | cut -c1-260
  done
fi
if [[ $what == *receive* ]]; then
  for sf in ${RXSFS:-7}; do
    LORAHIP_DEMOD_TIMING=1 timeout 300 python tools/receive_breakdown.py $sf > $O/${TAG}_receive_sf$sf.txt 2> $O/${TAG}_receive_sf$sf.err
    cat $O/${TAG}_receive_sf$sf.txt; tail -3 $O/${TAG}_receive_sf$sf.err
  done
fi
if [[ $what == *moving* ]]; then
  for sf in ${MVSFS:-7 8 9 10 11 12}; do
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
fi
if [[ $what == *steady* ]]; then
  for sf in ${STSFS:-7 10 12}; do
    timeout 200 python bench.py --sf $sf --no-cpu-baseline ${STEADY_ARGS:-} > $O/${TAG}_steady_sf$sf.json 2> $O/${TAG}_steady_sf$sf.err
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
if [[ $what == *bench* ]]; then
  timeout 600 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err; tail -c 600 $O/${TAG}_bench_default.err
  python tools/bench_digest.py $O/${TAG}_bench_default.json
fi
if [[ $what == *smoke* ]]; then
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.txt 2>&1; tail -3 $O/${TAG}_smoke.txt
fi
if [[ $what == *fma* ]]; then
  timeout 300 python tools/fma_report.py ${FMASFS:-7 10 12} > $O/${TAG}_fma_report.txt 2> $O/${TAG}_fma_report.err; cat $O/${TAG}_fma_report.txt; tail -2 $O/${TAG}_fma_report.err
  for sf in ${FMASFS:-7 10 12}; do
    for v in 0 40; do
      ( cd /tmp && timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY -d $O/${TAG}_pmc_fma_sf${sf}_v$v -o pmc --output-format csv -- \
          python $R/bench.py --sf $sf --variant $v --steps 3 --warmup 1 --ramp-seconds 0 --no-cpu-baseline > $O/${TAG}_pmc_fma_sf${sf}_v$v.log 2>&1 )
      python tools/pmc_kernels.py "$O/${TAG}_pmc_fma_sf${sf}_v$v" detect "SF$sf variant $v" | tee -a $O/${TAG}_fma_counters.txt
    done
  done
fi
