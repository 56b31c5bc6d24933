#!/bin/bash
# Copy one GPU session's evidence (TAG=fN bash tools/gpu_r06.sh final tests bench smoke) from gpurun_out/ into profiles/r06/ and
# profiles/traffic.json:   bash tools/install_evidence_r06.sh f9
set -e
NEW=$1
cd "$(dirname "$0")/.."
mkdir -p profiles/r06
for sf in 7 8 9 10 11 12; do
  cp gpurun_out/${NEW}_sf${sf}_timed_steps.txt gpurun_out/${NEW}_moving_sf${sf}_timed_steps.txt gpurun_out/${NEW}_sf${sf}_kernel_stats.csv gpurun_out/${NEW}_level3_sf${sf}_kernel_stats.txt profiles/r06/
  grep -v "rocprofv3\|^[WE]2026\|amdgpu.ids" gpurun_out/${NEW}_level3_sf$sf.txt > profiles/r06/${NEW}_level3_sf$sf.txt
done
cp gpurun_out/${NEW}_pmc_summary.txt profiles/r06/${NEW}_pmc_fetch_write_summary.txt
cp gpurun_out/traffic.json profiles/traffic.json
[ -f gpurun_out/${NEW}_bench_default.json ] && cp gpurun_out/${NEW}_bench_default.json profiles/r06/${NEW}_bench_default.json
[ -f gpurun_out/${NEW}_bench_default.json ] && python tools/bench_digest.py gpurun_out/${NEW}_bench_default.json > profiles/r06/${NEW}_bench_digest.txt
[ -f gpurun_out/${NEW}_pytest.txt ] && tail -3 gpurun_out/${NEW}_pytest.txt > profiles/r06/${NEW}_pytest_tail.txt
[ -f gpurun_out/${NEW}_smoke.txt ] && grep -v amdgpu.ids gpurun_out/${NEW}_smoke.txt > profiles/r06/${NEW}_smoke.txt
echo installed $NEW
[ -f gpurun_out/counters.json ] && cp gpurun_out/counters.json profiles/counters.json
[ -f gpurun_out/${NEW}_sq_counters_batch_kernels.txt ] && cp gpurun_out/${NEW}_sq_counters_batch_kernels.txt profiles/r06/
echo "installed counters of $NEW"
