cd $GRAFT_REPO_ROOT
TAG=r04_s5 bash tools/gpu_r04.sh tests
export LORAHIP_LIB=$PWD/lora_sdr_amd/liblorahip_allv.so
for rep in 1 2; do
for cfg in "10 0" "10 30" "10 31" "11 0" "11 31" "11 32" "12 0" "12 32" "12 33"; do set -- $cfg
  timeout 100 python bench.py --sf $1 --moving --variant $2 --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SF$1 moving variant $2: %.1f Msym/s frac %.4f launch %.1f us ser %s' % (d['value'], d['roofline']['frac'], d['roofline']['launch_us'], d['symbol_error_rate_vs_sent']))" | tee -a gpurun_out/r04_s5_moving_variants.txt
done; done
