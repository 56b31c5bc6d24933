#!/bin/bash
# Round-2 GPU sessions (one gpurun call each):  gpurun --timeout N -- 'bash tools/gpu_r02.sh <section> [...]'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
what="${*:-tests}"
export TMPDIR=/tmp
TAG=${TAG:-x}

SETS=( "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum"
       "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum"
       "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"
       "TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
       "SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVES" )

pmc_run() {   # pmc_run <outprefix> <cmd...>: one rocprofv3 pass per counter set
  local pre=$1; shift
  local i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    ( cd /tmp && timeout 300 rocprofv3 --pmc $set -d $O/${pre}_set$i -o pmc --output-format csv -- "$@" > $O/${pre}_set$i.log 2>&1 )
  done
}

if [[ $what == *tests* ]]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> $O/pytest_gpu.log
  tail -5 $O/pytest_gpu.log
fi

if [[ $what == *l2* ]]; then
  # L1/L2/SQ counters of the locked-receiver batch shape (bench.py --moving) and of the streaming demodulator
  for sf in ${L2SF:-7 10 12}; do
    timeout 200 python bench.py --sf $sf --no-cpu-baseline --moving > $O/${TAG}_moving_sf$sf.json 2> $O/${TAG}_moving_sf$sf.err
    pmc_run ${TAG}_l2_mov_sf$sf python $R/bench.py --sf $sf --moving --steps 3 --warmup 1 --ramp-seconds 0 --no-cpu-baseline
    python tools/pmc_kernels.py "$O/${TAG}_l2_mov_sf${sf}_set*" detect "bench.py --sf $sf --moving" | tee $O/${TAG}_l2_mov_sf$sf.txt
    case $sf in 7) CH=16384;; 8|9) CH=8192;; 10) CH=4096;; *) CH=1024;; esac
    timeout 200 python tools/bench_demod.py --sf $sf --channels $CH --modes 1 > $O/${TAG}_level3_sf$sf.txt 2>&1
    pmc_run ${TAG}_l2_str_sf$sf python $R/tools/bench_demod.py --sf $sf --channels $CH --modes 1
    python tools/pmc_kernels.py "$O/${TAG}_l2_str_sf${sf}_set*" demodStream "tools/bench_demod.py --sf $sf --channels $CH (streaming kernel)" | tee $O/${TAG}_l2_str_sf$sf.txt
    tail -2 $O/${TAG}_level3_sf$sf.txt
  done
fi

PMCA="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
PMCB="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES"
pmc1() {   # pmc1 <outprefix> <counters> <cmd...>: ONE rocprofv3 pass
  local pre=$1 set=$2; shift 2
  ( cd /tmp && timeout 300 rocprofv3 --pmc $set -d $O/${pre} -o pmc --output-format csv -- "$@" > $O/${pre}.log 2>&1 )
}
line() {   # line <json file> <label>
  python - "$1" "$2" <<'EOP'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-34s %8.1f Msym/s  frac %.3f  launch %.1f us  ser %s" % (sys.argv[2], d["value"], d["roofline"]["frac"], d["roofline"]["launch_us"], d.get("symbol_error_rate_vs_sent")))
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-600:])
EOP
}

if [[ $what == *after* ]]; then
  # the locked-receiver shape with the split tables (default) and with the round-1 table gather (A/B), then the L1/L2 counters
  for sf in ${L2SF:-7 10 12}; do
    timeout 200 python bench.py --sf $sf --no-cpu-baseline --moving > $O/${TAG}_moving_sf$sf.json 2> $O/${TAG}_moving_sf$sf.err
    line $O/${TAG}_moving_sf$sf.json "SF$sf moving, split tables"
    timeout 200 python bench.py --sf $sf --no-cpu-baseline --moving --fine-gather > $O/${TAG}_moving_gather_sf$sf.json 2> $O/${TAG}_moving_gather_sf$sf.err
    line $O/${TAG}_moving_gather_sf$sf.json "SF$sf moving, table gather"
  done
  for sf in ${PMCSF:-7 10}; do
    pmc1 ${TAG}_l2_mov_sf${sf}_setA "$PMCA" python $R/bench.py --sf $sf --moving --steps 3 --warmup 1 --ramp-seconds 0 --no-cpu-baseline
    python tools/pmc_kernels.py "$O/${TAG}_l2_mov_sf${sf}_set*" detect "bench.py --sf $sf --moving" | tee $O/${TAG}_l2_mov_sf$sf.txt
    case $sf in 7) CH=16384;; 8|9) CH=8192;; 10) CH=4096;; *) CH=1024;; esac
    timeout 200 python tools/bench_demod.py --sf $sf --channels $CH --modes 1 > $O/${TAG}_level3_sf$sf.txt 2>&1
    tail -1 $O/${TAG}_level3_sf$sf.txt
    pmc1 ${TAG}_l2_str_sf${sf}_setA "$PMCA" python $R/tools/bench_demod.py --sf $sf --channels $CH --modes 1
    python tools/pmc_kernels.py "$O/${TAG}_l2_str_sf${sf}_set*" demodStream "tools/bench_demod.py --sf $sf --channels $CH (streaming kernel)" | tee $O/${TAG}_l2_str_sf$sf.txt
  done
  pmc1 ${TAG}_l2_mov_sf7_setB "$PMCB" python $R/bench.py --sf 7 --moving --steps 3 --warmup 1 --ramp-seconds 0 --no-cpu-baseline
  python tools/pmc_kernels.py "$O/${TAG}_l2_mov_sf7_set*" detect "bench.py --sf 7 --moving" | tee $O/${TAG}_l2_mov_sf7.txt
fi

if [[ $what == *movvar* ]]; then
  # A/B of option sets on the locked-receiver shape (needs a library built with --all-variants)
  for spec in ${MOVVARS:-7:0 7:3 7:5 7:10 10:0 10:16 10:10 12:0 12:2 12:10}; do
    sf=${spec%%:*}; v=${spec##*:}
    timeout 200 python bench.py --sf $sf --no-cpu-baseline --moving --variant $v > $O/${TAG}_movvar_sf${sf}_v$v.json 2> $O/${TAG}_movvar_sf${sf}_v$v.err
    line $O/${TAG}_movvar_sf${sf}_v$v.json "SF$sf moving variant $v"
  done
fi

if [[ $what == *explore* ]]; then
  timeout 600 python tools/explore_r02.py moving > $O/${TAG}_explore_moving.txt 2>&1; cat $O/${TAG}_explore_moving.txt | grep -v amdgpu.ids
  for alt in ${STREAM_ALTS:-0}; do
    LORAHIP_STREAM_ALT=$alt timeout 300 python tools/explore_r02.py stream > $O/${TAG}_explore_stream_alt$alt.txt 2>&1; grep "^SF" $O/${TAG}_explore_stream_alt$alt.txt
  done
fi

if [[ $what == *final* ]]; then
  # the evidence profiles/r02 keeps: rocprofv3 kernel stats + the average of the timed steps for the three shapes bench.py reports
  # (steady state, locked receiver, streaming demodulator), HBM traffic counters for roofline.traffic
  for sf in ${FSF:-7 8 9 10 11 12}; do
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_sf$sf -o sf$sf --output-format csv -- \
        python $R/bench.py --sf $sf --steps 200 --warmup 20 --no-cpu-baseline > $O/${TAG}_prof_sf$sf.log 2>&1 )
    python tools/trace_tail.py $O/${TAG}_prof_sf$sf 200 | tee $O/${TAG}_sf${sf}_timed_steps.txt
    find $O/${TAG}_prof_sf$sf -name '*kernel_stats.csv' | head -1 | xargs -r -I{} cp {} $O/${TAG}_sf${sf}_kernel_stats.csv
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_mov_sf$sf -o sf$sf --output-format csv -- \
        python $R/bench.py --sf $sf --moving --steps 200 --warmup 20 --no-cpu-baseline > $O/${TAG}_prof_mov_sf$sf.log 2>&1 )
    python tools/trace_tail.py $O/${TAG}_prof_mov_sf$sf 200 | tee $O/${TAG}_moving_sf${sf}_timed_steps.txt
    case $sf in 7) CH=16384;; 8|9) CH=8192;; 10) CH=4096;; 11) CH=2048;; *) CH=1024;; esac
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_str_sf$sf -o sf$sf --output-format csv -- \
        python $R/tools/bench_demod.py --sf $sf --channels $CH --modes 1 > $O/${TAG}_level3_sf$sf.txt 2>&1 )
    find $O/${TAG}_prof_str_sf$sf -name '*kernel_stats.csv' | head -1 | xargs -r head -4 | cut -c1-200 | tee $O/${TAG}_level3_sf${sf}_kernel_stats.txt
    tail -1 $O/${TAG}_level3_sf$sf.txt
  done
  for sf in ${FSF:-7 8 9 10 11 12}; do
    for c in FETCH_SIZE WRITE_SIZE; do
      ( cd /tmp && timeout 300 rocprofv3 --pmc $c -d $O/pmc_${c}_sf$sf -o pmc --output-format csv -- \
          python $R/bench.py --sf $sf --steps 5 --warmup 1 --ramp-seconds 0 --no-cpu-baseline > $O/pmc_${c}_sf$sf.log 2>&1 )
    done
  done
  python tools/pmc_summary.py $O > $O/${TAG}_pmc_summary.txt 2>&1
  tail -25 $O/${TAG}_pmc_summary.txt
fi

if [[ $what == *bench* ]]; then
  timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
  echo "bench exit $?"; tail -c 3000 $O/${TAG}_bench.json; tail -5 $O/${TAG}_bench.err
fi
