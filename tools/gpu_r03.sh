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
