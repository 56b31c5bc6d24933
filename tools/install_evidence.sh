#!/bin/bash
# Copy one GPU session's evidence (tools/gpu_r02.sh tests final bench, TAG=sN) from gpurun_out/ into profiles/r02/, replacing the
# previous final set:   bash tools/install_evidence.sh s13 s11
set -e
NEW=$1; OLD=${2:-}
cd "$(dirname "$0")/.."
for sf in 7 8 9 10 11 12; do
  cp gpurun_out/${NEW}_sf${sf}_timed_steps.txt gpurun_out/${NEW}_moving_sf${sf}_timed_steps.txt gpurun_out/${NEW}_sf${sf}_kernel_stats.csv gpurun_out/${NEW}_level3_sf${sf}_kernel_stats.txt profiles/r02/
  grep -v "rocprofv3\|^[WE]2026\|amdgpu.ids" gpurun_out/${NEW}_level3_sf$sf.txt > profiles/r02/${NEW}_level3_sf$sf.txt
done
cp gpurun_out/${NEW}_pmc_summary.txt profiles/r02/${NEW}_pmc_fetch_write_summary.txt
cp gpurun_out/traffic.json profiles/traffic.json
grep "^{" gpurun_out/${NEW}_bench.json > profiles/r02/${NEW}_bench_default.json
tail -3 gpurun_out/pytest_gpu.log > profiles/r02/${NEW}_pytest_gpu_tail.txt
if [ -n "$OLD" ] && [ "$OLD" != "$NEW" ]; then
  git rm -q -f --ignore-unmatch profiles/r02/${OLD}_sf*_timed_steps.txt profiles/r02/${OLD}_moving_sf*_timed_steps.txt profiles/r02/${OLD}_sf*_kernel_stats.csv \
      profiles/r02/${OLD}_level3_sf*.txt profiles/r02/${OLD}_pmc_fetch_write_summary.txt profiles/r02/${OLD}_bench_default.json profiles/r02/${OLD}_pytest_gpu_tail.txt
  # only references into profiles/r02 move on (r01 has files with the same session prefixes)
  sed -i -E "s#(profiles/r01/)${OLD}_#\\1@@KEEP@@_#g; s/${OLD}_/${NEW}_/g; s#@@KEEP@@_#${OLD}_#g" profiles/r02/README.md DESIGN.md README.md
fi
