#!/bin/bash
# A/B of the streaming kernels' scheduling (alternating wavefront priority, lorahip_framemachine.h::alternatePriority; persistent grid),
# one GPU session. Builds (tools/build_variant.py <name> <flags> lorahip_wide.hip lorahip_stream.hip lorahip_demod.cpp):
#   base  -DLORAHIP_PRIO_ALTERNATE=0 -DLORAHIP_STREAM_PERSIST   run with LORAHIP_STREAM_BLOCKS=-1: no priority, one workgroup per channel set (rounds 1-3)
#   ship  a copy of the shipped liblorahip.so
#   pNN   -DLORAHIP_PRIO_ALTERNATE=NN [-DLORAHIP_WG_TIMELINE]; ppNN: the same with -DLORAHIP_STREAM_PERSIST (adds the persistent-grid leg)
#   gpurun -- 'LIBS="base ship" bash tools/ab_prio_r04.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
chans() { case $1 in 7) echo 16384,32768;; 8) echo 8192,16384;; 9) echo 8192,16384;; 10) echo 4096,8192;; 11) echo 2048,4096;; *) echo 1024,2048;; esac; }
for l in ${LIBS:-base ship}; do
  export LORAHIP_LIB=$R/lora_sdr_amd/liblorahip_$l.so
  unset LORAHIP_STREAM_BLOCKS
  if [[ $l == base ]]; then export LORAHIP_STREAM_BLOCKS=-1; fi
  echo "=== $l"
  for sf in ${SFS:-7 8 9 10 11 12}; do
    both=""; if [[ $l == pp* ]]; then both="--both"; fi
    timeout 300 python tools/level3_scaling.py $both --passes ${PASSES:-8} $sf $(chans $sf) 2>&1 | grep "^{"
  done
  if [[ -n "${TIMELINE:-}" ]]; then
    if [[ $l == pp* ]]; then export LORAHIP_STREAM_BLOCKS=1024; fi
    timeout 120 python tools/wg_timeline.py 11 2048 2 2>&1 | grep "pass\|duration\|residency"
    if [[ $l == pp* ]]; then export LORAHIP_STREAM_BLOCKS=512; fi
    timeout 120 python tools/wg_timeline.py 12 1024 2 2>&1 | grep "pass\|duration\|residency"
  fi
done
