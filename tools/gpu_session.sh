#!/bin/bash
# One gpurun call: GPU parity tests, bench per SF, rocprofv3 kernel stats and PMC passes.
# Everything lands under gpurun_out/ (scratch); summaries worth judging are copied to profiles/.
#   gpurun --timeout 1500 -- 'bash tools/gpu_session.sh [tests] [bench] [prof] [pmc] [membw]'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
what="${*:-tests bench prof pmc membw}"
export TMPDIR=/tmp

if [[ $what == *tests* ]]; then
  timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> $O/pytest_gpu.log
  tail -5 $O/pytest_gpu.log
fi

if [[ $what == *bench* ]]; then
  timeout 600 python bench.py > $O/bench_sf7.json 2> $O/bench_sf7.err
  tail -c 2500 $O/bench_sf7.json
  for sf in 8 9 10 11 12; do
    timeout 300 python bench.py --sf $sf --cpu-seconds 3 > $O/bench_sf$sf.json 2> $O/bench_sf$sf.err
    python - <<EOF
import json
try:
    d = json.loads(open("$O/bench_sf$sf.json").read().strip().splitlines()[-1])
    print("SF$sf", round(d["value"], 1), "Msym/s frac", round(d["roofline"]["frac"], 3), "cpu", d.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print("SF$sf bench failed", e)
EOF
  done
fi

if [[ $what == *membw* ]]; then
  timeout 300 python tools/membw.py > $O/membw.log 2>&1
  cat $O/membw.log
fi

if [[ $what == *prof* ]]; then
  for sf in ${PSF:-7 12}; do
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_sf$sf -o sf$sf --output-format csv -- \
        python $R/bench.py --sf $sf --steps 300 --warmup 20 --no-cpu-baseline > $O/prof_sf$sf.log 2>&1 )
    find $O/prof_sf$sf -name '*kernel_stats.csv' | head -1 | xargs -r head -3
    python tools/trace_tail.py $O/prof_sf$sf 300 | tee $O/prof_sf$sf.timed.txt; tail -c 600 $O/prof_sf$sf.log
  done
fi

if [[ $what == *pmc* ]]; then
  for sf in ${PMCSF:-7 8 9 10 11 12}; do
    for c in FETCH_SIZE WRITE_SIZE; do
      ( cd /tmp && timeout 600 rocprofv3 --pmc $c -d $O/pmc_${c}_sf$sf -o pmc --output-format csv -- \
          python $R/bench.py --sf $sf --steps 5 --warmup 1 --ramp-seconds 0 --no-cpu-baseline > $O/pmc_${c}_sf$sf.log 2>&1 )
    done
  done
  python tools/pmc_summary.py $O > $O/pmc_summary.txt 2>&1
  cat $O/pmc_summary.txt
fi

if [[ $what == *variants* ]]; then
  for sf in ${VSF:-11 12}; do
    for v in ${VARS:-0 2 3 4 5}; do
      timeout 200 python bench.py --sf $sf --variant $v --no-cpu-baseline > $O/var_sf${sf}_v$v.json 2> $O/var_sf${sf}_v$v.err
      python - <<EOF2
import json
try:
    d = json.loads(open("$O/var_sf${sf}_v$v.json").read().strip().splitlines()[-1])
    print("SF$sf variant $v:", round(d["value"], 1), "Msym/s frac", round(d["roofline"]["frac"], 3), "ser", d["symbol_error_rate_vs_sent"])
except Exception as e:
    print("SF$sf variant $v failed", e)
EOF2
    done
  done
fi

if [[ $what == *sq* ]]; then
  rocprofv3 -L > $O/counters_list.txt 2>&1
  for sf in ${SQSF:-7 12}; do
    i=0
    for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
               "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
               "GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
      i=$((i+1))
      ( cd /tmp && timeout 300 rocprofv3 --pmc $set -d $O/pmc_SQ${i}_sf$sf -o pmc --output-format csv -- \
          python $R/bench.py --sf $sf --steps 3 --warmup 1 --ramp-seconds 0 --no-cpu-baseline > $O/pmc_SQ${i}_sf$sf.log 2>&1 )
    done
  done
  python tools/pmc_summary.py $O > $O/pmc_summary_sq.txt 2>&1
  cat $O/pmc_summary_sq.txt
fi

if [[ $what == *chan* ]]; then
  CH="${CHARGS:---channels 8 --decim 8 --taps 64}"
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/chan_prof -o chan --output-format csv -- \
      python $R/tools/bench_chan.py $CH > $O/chan_prof.log 2>&1 )
  grep "^K=" $O/chan_prof.log
  find $O/chan_prof -name "*kernel_stats.csv" | head -1 | xargs head -6
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
             "GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
    i=$((i+1))
    ( cd /tmp && timeout 300 rocprofv3 --pmc $set -d $O/chan_SQ$i -o pmc --output-format csv -- \
        python $R/tools/bench_chan.py $CH --reps 2 > $O/chan_SQ$i.log 2>&1 )
  done
  python - <<EOF2
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$O/chan_SQ*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "channelize" in r.get("Kernel_Name", ""):
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot):
    print("%-24s %16.0f per launch (%d launches)" % (k, tot[k] / n[k], n[k]))
EOF2
fi

if [[ $what == *tailtest* ]]; then
  timeout 120 tools/tailtest.bin > $O/tailtest.log 2>&1; cat $O/tailtest.log
fi

if [[ $what == *quick* ]]; then
  for sf in ${QSF:-7 8 9 10 11 12}; do
    timeout 200 python bench.py --sf $sf --no-cpu-baseline > $O/q_sf$sf.json 2> $O/q_sf$sf.err
    python - <<EOF2
import json
try:
    d = json.loads(open("$O/q_sf$sf.json").read().strip().splitlines()[-1])
    print("SF$sf:", round(d["value"], 1), "Msym/s frac", round(d["roofline"]["frac"], 3), "launch_us", round(d["roofline"]["launch_us"],1), "ser", d["symbol_error_rate_vs_sent"])
except Exception as e:
    print("SF$sf failed", e); print(open("$O/q_sf$sf.err").read()[-800:])
EOF2
  done
fi

if [[ $what == *alias* ]]; then
  for sf in ${QSF:-7 10 12}; do
    for extra in "" "--alias-windows"; do
      timeout 200 python bench.py --sf $sf --no-cpu-baseline $extra > $O/alias_sf$sf.json 2> $O/alias_sf$sf.err
      python - <<EOF2
import json
try:
    d = json.loads(open("$O/alias_sf$sf.json").read().strip().splitlines()[-1])
    print("SF$sf $extra:", round(d["value"], 1), "Msym/s frac", round(d["roofline"]["frac"], 3), "launch_us", round(d["roofline"]["launch_us"],1))
except Exception as e:
    print("SF$sf failed", e); print(open("$O/alias_sf$sf.err").read()[-800:])
EOF2
    done
  done
fi

if [[ $what == *multi* ]]; then
  # the N > 1 path of bench.py on the one GPU that is here: 2 ranks over gloo, both on device 0 (halved batch so both fit)
  LORA_BENCH_BACKEND=gloo LORA_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 2 --steps 100 --warmup 10 --channels 2048 > $O/multi2.json 2> $O/multi2.err
  tail -c 1200 $O/multi2.json; tail -3 $O/multi2.err
fi

if [[ $what == *moving* ]]; then
  for sf in ${QSF:-7 10 12}; do
    timeout 200 python bench.py --sf $sf --no-cpu-baseline --moving > $O/mov_sf$sf.json 2> $O/mov_sf$sf.err
    python - <<EOF2
import json
try:
    d = json.loads(open("$O/mov_sf$sf.json").read().strip().splitlines()[-1])
    print("SF$sf moving fine index:", round(d["value"], 1), "Msym/s, launch_us", round(d["roofline"]["launch_us"],1))
except Exception as e:
    print("SF$sf failed", e); print(open("$O/mov_sf$sf.err").read()[-600:])
EOF2
  done
fi
