#!/bin/bash
# Level 3 end to end against kernel time as the run gets longer (frames per channel): the fixed cost per run (launch, state D2H, host
# scan, packet packing) against kernel time.   gpurun -- 'TAG=sNN bash tools/run_length.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r03_${TAG:-s31}; mkdir -p $O; : > $O/run_length.txt
for cfg in "7 16384 4" "7 16384 8" "7 16384 16" "7 16384 32" "8 8192 4" "8 8192 16" "10 4096 4" "10 4096 16" "12 1024 4" "12 1024 16"; do
  set -- $cfg
  timeout 300 python tools/bench_demod.py --sf $1 --channels $2 --frames $3 --modes 1 --reps 5 2>&1 | grep -v amdgpu.ids | tee -a $O/run_length.txt
done
