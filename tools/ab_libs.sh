#!/bin/bash
# A/B of two builds of the library on ONE box: level-3 kernel time per SF, alternating.  LIBS="build_ab/liblorahip_x.so" TAG=sNN bash tools/ab_libs.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r03_${TAG:-ab}; mkdir -p $O; : > $O/ab_libs.txt
for rep in 1 2; do
for lib in "" ${LIBS:-}; do
  for cfg in "7 16384" "10 4096" "12 1024"; do set -- $cfg
    echo "== lib ${lib:-current} SF$1" >> $O/ab_libs.txt
    LORAHIP_LIB=${lib:+$R/$lib} timeout 120 python tools/bench_demod.py --sf $1 --channels $2 --frames 4 --modes 1 --reps 5 2>&1 | grep "mode 1" | cut -c1-120 >> $O/ab_libs.txt
  done
done
done
