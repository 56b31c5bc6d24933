#!/bin/bash
# the streaming kernel against the stride of the channels' symbol rows (LORAHIP_SYM_PAD, lorahip_demod.cpp::runStream)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r03_${TAG:-stride}; mkdir -p $O; : > $O/row_stride.txt
for rep in 1 2; do
for pad in ${PADS:--257 -256 -193 -129 -1 0 255}; do
    echo "== pad $pad SF7" >> $O/row_stride.txt
    LORAHIP_SYM_PAD=$pad timeout 100 python tools/bench_demod.py --sf 7 --channels 16384 --frames 4 --modes 1 --reps 5 2>&1 | grep "mode 1" | cut -c1-120 >> $O/row_stride.txt
done
done
