#!/bin/bash
# SQ instruction counters of the final kernels (one rocprofv3 --pmc pass per workload): steady, moving, streaming at one SF
#   gpurun --timeout 900 -- 'TAG=s28 SF=7 bash tools/gpu_pmc_final.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; TAG=${TAG:-p}; SF=${SF:-7}; export TMPDIR=/tmp
PMC="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVES"
case $SF in 7) CH=16384;; 8|9) CH=8192;; 10) CH=4096;; 11) CH=2048;; *) CH=1024;; esac
run() { local pre=$1; shift; ( cd /tmp && timeout 300 rocprofv3 --pmc $PMC -d $O/${TAG}_$pre -o pmc --output-format csv -- "$@" > $O/${TAG}_$pre.log 2>&1 ); }
run steady_sf$SF python $R/bench.py --sf $SF --steps 3 --warmup 1 --ramp-seconds 0 --no-cpu-baseline
python tools/pmc_kernels.py "$O/${TAG}_steady_sf$SF" detect "bench.py --sf $SF (steady state, final kernels)" | tee $O/${TAG}_sq_steady_sf$SF.txt
run moving_sf$SF python $R/bench.py --sf $SF --moving --steps 3 --warmup 1 --ramp-seconds 0 --no-cpu-baseline
python tools/pmc_kernels.py "$O/${TAG}_moving_sf$SF" detect "bench.py --sf $SF --moving (final kernels)" | tee $O/${TAG}_sq_moving_sf$SF.txt
run stream_sf$SF python $R/tools/bench_demod.py --sf $SF --channels $CH --modes 1 --reps 2 --ramp-seconds 0
python tools/pmc_kernels.py "$O/${TAG}_stream_sf$SF" demodStream "tools/bench_demod.py --sf $SF --channels $CH (streaming kernel, final)" | tee $O/${TAG}_sq_stream_sf$SF.txt
