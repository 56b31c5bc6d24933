#!/bin/bash
# sample clocks / power while a long bench runs:  bash tools/power_trace.sh <tag> <bench args...>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
tag=$1; shift
( cd $R && python bench.py "$@" --no-cpu-baseline > $O/pt_$tag.json 2> $O/pt_$tag.err ) &
pid=$!
: > $O/pt_$tag.smi
for i in $(seq 1 60); do
  kill -0 $pid 2>/dev/null || break
  echo "t=$i $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|mclk|fclk|Power' | sed -E 's/GPU\[0\]\s*: //; s/clock level: //; s/Current Socket Graphics Package //' | tr '\n' ' ')" >> $O/pt_$tag.smi
  sleep 0.5
done
wait $pid
python - <<EOF2
import json,re
d = json.loads(open("$O/pt_$tag.json").read().strip().splitlines()[-1])
print("$tag:", round(d["value"],1), "Msym/s, launch_us", round(d["roofline"]["launch_us"],1))
for l in open("$O/pt_$tag.smi").read().splitlines():
    print("   ", re.sub(r"\s+", " ", l)[:300])
EOF2
