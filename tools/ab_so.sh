#!/bin/bash
# A/B of two builds of the library inside ONE gpurun call (boxes differ by several per cent; only same-call figures compare):
# runs `python <script> <args>` with cvpr2020_manet_amd/libmanet_hip.so ("new") and with libmanet_hip_old.so ("old") swapped in,
# alternating N times.   usage: tools/ab_so.sh N script.py [args...]
N=$1; shift
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp cvpr2020_manet_amd/libmanet_hip.so /tmp/ab_new.so; cp cvpr2020_manet_amd/libmanet_hip_old.so /tmp/ab_old.so
for i in $(seq $N); do
  for v in new old; do
    cp /tmp/ab_$v.so cvpr2020_manet_amd/libmanet_hip.so
    echo "== $v (round $i)"; python "$@" 2>&1 | grep -v amdgpu.ids
  done
done
cp /tmp/ab_new.so cvpr2020_manet_amd/libmanet_hip.so
