#!/bin/bash
# sample clocks / power while any command runs:  bash tools/power_trace_cmd.sh <tag> <command...>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
tag=$1; shift
( cd $R && "$@" > $O/ptc_$tag.log 2>&1 ) &
pid=$!
: > $O/ptc_$tag.smi
for i in $(seq 1 60); do
  kill -0 $pid 2>/dev/null || break
  echo "t=$i $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Power' | sed -E 's/GPU\[0\]\s*: //; s/clock level: //; s/Current Socket Graphics Package //; s/=+ Power Consumption =+//' | tr '\n' ' ')" >> $O/ptc_$tag.smi
  sleep 0.5
done
wait $pid
tail -2 $O/ptc_$tag.log
sed -E 's/\s+/ /g' $O/ptc_$tag.smi | cut -c1-160
