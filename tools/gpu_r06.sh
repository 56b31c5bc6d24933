#!/bin/bash
# round-6 GPU sessions: gpurun --timeout N -- 'TAG=s1 bash tools/gpu_r06.sh tests tworank level3 ...'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; TAG=${TAG:-r06}
export TMPDIR=/tmp
what="$*"
l3ch() { case $1 in 6|7) echo 16384;; 8|9) echo 8192;; 10) echo 4096;; 11) echo 2048;; *) echo 1024;; esac; }
if [[ $what == *tests* ]]; then
  timeout ${TEST_TIMEOUT:-900} python -m pytest ${TESTS:-tests} -m gpu -x -q ${PYTEST_ARGS:-} > $O/${TAG}_pytest.txt 2>&1; tail -${TEST_TAIL:-6} $O/${TAG}_pytest.txt
fi
if [[ $what == *tworank* ]]; then
  # the --gpus 2 code path on one device over gloo: the line must carry cpu_baseline + oracle like the N = 1 line
  LORA_BENCH_BACKEND=gloo LORA_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 ${TWORANK_ARGS:-} > $O/${TAG}_bench_two_ranks_one_device_gloo.json 2> $O/${TAG}_bench_two_ranks.err
  tail -c 400 $O/${TAG}_bench_two_ranks.err
  python tools/bench_digest.py $O/${TAG}_bench_two_ranks_one_device_gloo.json
fi
if [[ $what == *eightrank* ]]; then
  # the --gpus 8 code path (what the driver's SCALE run launches) on the ONE device of this box over gloo: a code-path check -- the line's
  # size, its keys and the wall time (rank 0 runs every CPU leg; the driver's limit is 1800 s) -- not a scaling measurement
  t0=$(date +%s)
  LORA_BENCH_BACKEND=gloo LORA_BENCH_ONE_DEVICE=1 timeout 1500 python bench.py --gpus 8 --steps 20 --warmup 5 > $O/${TAG}_bench_eight_ranks_one_device_gloo.json 2> $O/${TAG}_bench_eight_ranks.err
  echo "eight ranks on one device: rc $? wall $(( $(date +%s) - t0 )) s" | tee $O/${TAG}_bench_eight_ranks_wall.txt
  tail -c 300 $O/${TAG}_bench_eight_ranks.err
  python tools/bench_digest.py $O/${TAG}_bench_eight_ranks_one_device_gloo.json | head -30
fi
if [[ $what == *level3* ]]; then
  for sf in ${L3SFS:-7 8 9 10 11 12}; do
    timeout 200 python tools/bench_demod.py --sf $sf --channels ${L3CH:-$(l3ch $sf)} --modes 1 > $O/${TAG}_level3_sf$sf.txt 2>&1
    tail -1 $O/${TAG}_level3_sf$sf.txt | cut -c1-260
  done
fi
if [[ $what == *scaling* ]]; then
  timeout 600 python tools/level3_scaling.py ${SCALING_ARGS:-} > $O/${TAG}_level3_scaling.txt 2>&1; cat $O/${TAG}_level3_scaling.txt
fi
if [[ $what == *moving* ]]; then
  for sf in ${MVSFS:-7 8 9 10 11 12}; do
    timeout 200 python bench.py --sf $sf --no-cpu-baseline --moving > $O/${TAG}_moving_sf$sf.json 2> $O/${TAG}_moving_sf$sf.err
    python - $O/${TAG}_moving_sf$sf.json $sf <<'EOP'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("SF%s moving %8.1f Msym/s frac %.3f launch %.1f us oracle %s" % (sys.argv[2], d["value"], d["roofline"]["frac"], d["roofline"]["launch_us"], d.get("oracle", {}).get("index_mismatches")))
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
  timeout 700 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err; tail -c 600 $O/${TAG}_bench_default.err
  python tools/bench_digest.py $O/${TAG}_bench_default.json
fi
if [[ $what == *smoke* ]]; then
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.txt 2>&1; tail -3 $O/${TAG}_smoke.txt
fi
if [[ $what == *pmcl3* ]]; then
  # SQ counters of the streaming kernel at SF7 by channel count / lanes per channel: separate --pmc passes, no traces
  : > $O/${TAG}_sq_counters_level3_sf7.txt
  for cfg in ${PMCL3:-"4096 -1" "4096 0" "16384 -1" "2048 -1" "2048 0"}; do
    set -- $cfg; cnt=$1; lanes=$2
    d=$O/${TAG}_pmc_l3_sf7_${cnt}_lanes${lanes}
    ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU -d ${d}_issue -o pmc --output-format csv -- \
        python $R/tools/level3_scaling.py --sf 7 --counts $cnt --lanes $lanes --passes 2 > ${d}.log 2>&1 )
    ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD -d ${d}_lds -o pmc --output-format csv -- \
        python $R/tools/level3_scaling.py --sf 7 --counts $cnt --lanes $lanes --passes 2 >> ${d}.log 2>&1 )
    python tools/pmc_kernels.py "${d}_*" demodStream "SF7 level 3, $cnt channels, lanes $lanes" | tee -a $O/${TAG}_sq_counters_level3_sf7.txt
    grep "^SF7" ${d}.log | tail -1 | tee -a $O/${TAG}_sq_counters_level3_sf7.txt
    rm -rf ${d}_issue ${d}_lds
  done
fi
if [[ $what == *final* ]]; then
  # the evidence profiles/r06 keeps (as tools/gpu_r02.sh final): rocprofv3 kernel stats + the average of the timed steps for the shapes
  # bench.py reports (steady state, locked receiver, streaming demodulator), HBM traffic counters for roofline.traffic
  for sf in ${FSF:-7 8 9 10 11 12}; do
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_sf$sf -o sf$sf --output-format csv -- \
        python $R/bench.py --sf $sf --steps 200 --warmup 20 --no-cpu-baseline > $O/${TAG}_prof_sf$sf.log 2>&1 )
    python tools/trace_tail.py $O/${TAG}_prof_sf$sf 200 | tee $O/${TAG}_sf${sf}_timed_steps.txt
    find $O/${TAG}_prof_sf$sf -name '*kernel_stats.csv' | head -1 | xargs -r -I{} cp {} $O/${TAG}_sf${sf}_kernel_stats.csv
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_mov_sf$sf -o sf$sf --output-format csv -- \
        python $R/bench.py --sf $sf --moving --steps 200 --warmup 20 --no-cpu-baseline > $O/${TAG}_prof_mov_sf$sf.log 2>&1 )
    python tools/trace_tail.py $O/${TAG}_prof_mov_sf$sf 200 | tee $O/${TAG}_moving_sf${sf}_timed_steps.txt
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_str_sf$sf -o sf$sf --output-format csv -- \
        python $R/tools/bench_demod.py --sf $sf --channels $(l3ch $sf) --modes 1 > $O/${TAG}_level3_sf$sf.txt 2>&1 )
    find $O/${TAG}_prof_str_sf$sf -name '*kernel_stats.csv' | head -1 | xargs -r head -4 | cut -c1-200 | tee $O/${TAG}_level3_sf${sf}_kernel_stats.txt
    tail -1 $O/${TAG}_level3_sf$sf.txt
    rm -rf $O/${TAG}_prof_sf$sf $O/${TAG}_prof_mov_sf$sf $O/${TAG}_prof_str_sf$sf
  done
  rm -rf $O/pmc_FETCH_SIZE_sf* $O/pmc_WRITE_SIZE_sf*
  for sf in ${FSF:-7 8 9 10 11 12}; do
    for c in FETCH_SIZE WRITE_SIZE; do
      ( cd /tmp && timeout 300 rocprofv3 --pmc $c -d $O/pmc_${c}_sf$sf -o pmc --output-format csv -- \
          python $R/bench.py --sf $sf --steps 5 --warmup 1 --ramp-seconds 0 --no-cpu-baseline > $O/pmc_${c}_sf$sf.log 2>&1 )
    done
  done
  python tools/pmc_summary.py $O > $O/${TAG}_pmc_summary.txt 2>&1
  tail -25 $O/${TAG}_pmc_summary.txt
  rm -rf $O/pmc_FETCH_SIZE_sf* $O/pmc_WRITE_SIZE_sf*
fi
if [[ $what == *counters* ]]; then
  # north_star's "LDS/VALU utilisation against gfx950 peak" for the batch kernels of THIS tree: two SQ passes per shape
  # and SF (8 SQ slots per pass), counters only -- no trace domains beside --pmc. -> gpurun_out/counters.json (tools/pmc_counters.py),
  # installed as profiles/counters.json and replayed by bench.py as roofline.valu_busy / lds_busy / instr_per_sample
  rm -rf $O/pmc_SQ*_sf*
  for sf in ${CSF:-7 8 9 10 11 12}; do
    for shape in steady moving; do
      mv=""; [[ $shape == moving ]] && mv="--moving"
      ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES \
          -d $O/pmc_SQA_${shape}_sf$sf -o pmc --output-format csv -- python $R/bench.py --sf $sf $mv --steps 3 --warmup 1 --ramp-seconds 0 --no-cpu-baseline > $O/pmc_SQA_${shape}_sf$sf.log 2>&1 )
      ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
          -d $O/pmc_SQB_${shape}_sf$sf -o pmc --output-format csv -- python $R/bench.py --sf $sf $mv --steps 3 --warmup 1 --ramp-seconds 0 --no-cpu-baseline > $O/pmc_SQB_${shape}_sf$sf.log 2>&1 )
    done
  done
  TAG=$TAG python tools/pmc_counters.py $O $TAG | tee $O/${TAG}_sq_counters_batch_kernels.txt
  rm -rf $O/pmc_SQ*_sf*
fi
if [[ $what == *abfine* ]]; then
  # the two declined micro-steps of the moving-index kernels, BUILT (tools/build_variant.py pow2 / skew) and measured as
  # in-session A/B pairs: 5 alternations of `bench.py --sf S --moving` per library, the launch time of each; then, per library, the parity
  # tests that cover the moving index on that build, and the SQ counters of the moving kernel at SF11 / SF12
  : > $O/${TAG}_ab_fine_moving.txt
  for rep in 1 2 3 4 5; do
    for sf in ${ABSF:-10 11 12}; do
      for lib in ${ABLIBS:-cur pow2 skew}; do
        path=$R/lora_sdr_amd/liblorahip_$lib.so; [[ $lib == cur ]] && path=$R/lora_sdr_amd/liblorahip.so
        LORAHIP_LIB=$path timeout 200 python bench.py --sf $sf --no-cpu-baseline --moving 2>/dev/null | python -c "
import json, sys
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('rep %s SF%s %-5s moving %8.1f Msym/s frac %.4f launch %.2f us oracle mismatches %s' % (sys.argv[1], sys.argv[2], sys.argv[3], d['value'], d['roofline']['frac'], d['roofline']['launch_us'], d.get('oracle', {}).get('index_mismatches')))
" $rep $sf $lib | tee -a $O/${TAG}_ab_fine_moving.txt
      done
    done
  done
  python - $O/${TAG}_ab_fine_moving.txt <<'EOP' | tee -a $O/${TAG}_ab_fine_moving.txt
import re, sys, statistics
rows = {}
for ln in open(sys.argv[1]):
    m = re.match(r"rep (\d+) SF(\d+) (\w+)\s+moving\s+([0-9.]+) Msym/s frac ([0-9.]+) launch ([0-9.]+) us", ln)
    if m:
        rows.setdefault((int(m.group(2)), m.group(3)), []).append(float(m.group(6)))
print("median launch us of the 5 alternations, and against cur:")
for (sf, lib), v in sorted(rows.items()):
    base = statistics.median(rows.get((sf, "cur"), v))
    print("  SF%d %-5s %.2f us  (%+.2f %% time vs cur)  [%s]" % (sf, lib, statistics.median(v), (statistics.median(v) / base - 1) * 100, " ".join("%.1f" % x for x in v)))
EOP
  for lib in ${ABLIBS:-cur pow2 skew}; do
    path=$R/lora_sdr_amd/liblorahip_$lib.so; [[ $lib == cur ]] && path=$R/lora_sdr_amd/liblorahip.so
    LORAHIP_LIB=$path timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_demod.py -m gpu -x -q -k "fine or moving or recurrence" > $O/${TAG}_ab_fine_parity_$lib.txt 2>&1
    echo "parity [$lib]: $(tail -1 $O/${TAG}_ab_fine_parity_$lib.txt)" | tee -a $O/${TAG}_ab_fine_moving.txt
    for sf in 11 12; do
      d=$O/${TAG}_pmc_abfine_${lib}_sf$sf
      ( cd /tmp && LORAHIP_LIB=$path timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVES -d $d -o pmc --output-format csv -- \
          python $R/bench.py --sf $sf --moving --steps 3 --warmup 1 --ramp-seconds 0 --no-cpu-baseline > $d.log 2>&1 )
      python tools/pmc_kernels.py "$d" detect "moving SF$sf [$lib]" | tee -a $O/${TAG}_ab_fine_moving.txt
      rm -rf $d
    done
  done
fi
if [[ $what == *custom* ]]; then
  bash -c "$CUSTOM" > $O/${TAG}_custom.txt 2>&1; tail -${CUSTOM_TAIL:-40} $O/${TAG}_custom.txt
fi
