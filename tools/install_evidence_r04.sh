#!/bin/bash
# Copy one GPU session's evidence (TAG=fN bash tools/gpu_r02.sh final; TAG=fNsq bash tools/gpu_r03.sh ldsconf) from gpurun_out/ into
# profiles/r04/ and profiles/traffic.json:   bash tools/install_evidence_r04.sh f4
set -e
NEW=$1
cd "$(dirname "$0")/.."
mkdir -p profiles/r04
for sf in 7 8 9 10 11 12; do
  cp gpurun_out/${NEW}_sf${sf}_timed_steps.txt gpurun_out/${NEW}_moving_sf${sf}_timed_steps.txt gpurun_out/${NEW}_sf${sf}_kernel_stats.csv gpurun_out/${NEW}_level3_sf${sf}_kernel_stats.txt profiles/r04/
  grep -v "rocprofv3\|^[WE]2026\|amdgpu.ids" gpurun_out/${NEW}_level3_sf$sf.txt > profiles/r04/${NEW}_level3_sf$sf.txt
done
cp gpurun_out/${NEW}_pmc_summary.txt profiles/r04/${NEW}_pmc_fetch_write_summary.txt
cp gpurun_out/traffic.json profiles/traffic.json
if [ -f gpurun_out/r03_${NEW}sq/ldsconf.txt ]; then cp gpurun_out/r03_${NEW}sq/ldsconf.txt profiles/r04/${NEW}_sq_counters_sf11_sf12.txt; fi
