R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2 3 4 5; do
  for lib in cur m10a m10c; do
    path=$R/lora_sdr_amd/liblorahip_$lib.so; [[ $lib == cur ]] && path=$R/lora_sdr_amd/liblorahip.so
    LORAHIP_LIB=$path timeout 200 python bench.py --sf 10 --no-cpu-baseline --moving 2>/dev/null | python -c "
import json, sys
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('rep %s SF10 %-5s moving %8.1f Msym/s frac %.4f launch %.2f us oracle mismatches %s' % (sys.argv[1], sys.argv[2], d['value'], d['roofline']['frac'], d['roofline']['launch_us'], d.get('oracle', {}).get('index_mismatches')))
" $rep $lib
  done
done
