#!/bin/bash
# Round-3 GPU sessions: gpurun --timeout N -- 'bash tools/gpu_r03.sh <what...>'; everything lands under gpurun_out/r03_<tag>/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
TAG=${TAG:-s1}; O=$R/gpurun_out/r03_$TAG; mkdir -p $O
export TMPDIR=/tmp
what="$*"
if [[ $what == *tests* ]]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -15 $O/pytest_gpu.log
fi
if [[ $what == *loopback* ]]; then
  timeout 600 python tools/loopback_probe.py 21 22 27 28 > $O/loopback_probe.txt 2>&1; tail -20 $O/loopback_probe.txt
fi
if [[ $what == *bench* ]]; then
  timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench exit $?"; tail -c 6000 $O/bench_default.json; tail -25 $O/bench_default.err
fi
if [[ $what == *quicktests* ]]; then
  timeout 900 python -m pytest tests/test_gpu_demod.py tests/test_gpu_dropin.py tests/test_gpu_codec.py tests/test_gpu_bench.py -m gpu -x -q > $O/pytest_quick.log 2>&1; echo "pytest exit $?" >> $O/pytest_quick.log; tail -4 $O/pytest_quick.log
fi
if [[ $what == *ab* ]]; then
  bash tools/gpu_ab.sh 2>&1 | tee $O/ab.txt
fi
if [[ $what == *ldsconf* ]]; then
  # LDS bank conflicts and LDS / VALU activity of the locked-receiver shape, the steady state and the streaming kernel
  for sf in ${CSF:-7 10 12}; do
    case $sf in 7) CH=16384;; 8|9) CH=8192;; 10) CH=4096;; 11) CH=2048;; *) CH=1024;; esac
    for shape in moving steady stream; do
      case $shape in
        moving) CMD="python $R/bench.py --sf $sf --steps 3 --warmup 1 --ramp-seconds 0 --no-cpu-baseline --moving"; KEY=detect;;
        steady) CMD="python $R/bench.py --sf $sf --steps 3 --warmup 1 --ramp-seconds 0 --no-cpu-baseline"; KEY=detect;;
        stream) CMD="python $R/tools/bench_demod.py --sf $sf --channels $CH --modes 1 --reps 2 --ramp-seconds 0"; KEY=demodStream;;
      esac
      i=0
      for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
                 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_WAVES"; do
        i=$((i+1))
        ( cd /tmp && timeout 300 rocprofv3 --pmc $set -d $O/pmc_${shape}_sf${sf}_$i -o pmc --output-format csv -- $CMD > $O/pmc_${shape}_sf${sf}_$i.log 2>&1 )
      done
      python tools/pmc_kernels.py $O/pmc_${shape}_sf${sf}_1 $KEY "== $shape SF$sf (LDS)" | tee -a $O/ldsconf.txt
      python tools/pmc_kernels.py $O/pmc_${shape}_sf${sf}_2 $KEY "== $shape SF$sf (issue)" | tee -a $O/ldsconf.txt
    done
  done
fi
if [[ $what == *e2e* ]]; then
  for sf in ${ESF:-7 8}; do LORAHIP_DEMOD_TIMING=1 timeout 300 python tools/e2e_breakdown.py --sf $sf 2>&1 | tail -14 | tee -a $O/e2e_breakdown.txt; done
fi
if [[ $what == *geo64* ]]; then
  # the two-phase geometries of 64 points per lane (variants 40 / 41 of an --all-variants build) against the defaults
  for sf in 10 11 12; do
    for v in 0 25 26; do
      LORAHIP_LIB=$R/lora_sdr_amd/liblorahip_AV.so timeout 200 python bench.py --sf $sf --variant $v --no-cpu-baseline > $O/geo64_sf${sf}_v$v.json 2> $O/geo64_sf${sf}_v$v.err
      python - <<EOF2 | tee -a $O/geo64.txt
import json
try:
    d = json.loads(open("$O/geo64_sf${sf}_v$v.json").read().strip().splitlines()[-1])
    print("SF$sf variant $v:", round(d["value"], 1), "Msym/s frac", round(d["roofline"]["frac"], 3), "launch_us", d["roofline"]["launch_us"], "index mismatches vs oracle", d["oracle"]["index_mismatches"])
except Exception as e:
    print("SF$sf variant $v failed:", open("$O/geo64_sf${sf}_v$v.err").read().strip().splitlines()[-1][:200])
EOF2
    done
  done
fi
if [[ $what == *smoke* ]]; then
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.txt
fi
