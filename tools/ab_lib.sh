#!/bin/bash
# A/B of two builds of the library in one gpurun call: runs bench.py with libmanet_hip.so, then with the alternative .so
# swapped in (cvpr2020_manet_amd/libmanet_hip_old.so), alternating.   usage: tools/ab_lib.sh "<bench args>" [rounds]
ARGS=$1; N=${2:-2}
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp cvpr2020_manet_amd/libmanet_hip.so /tmp/new.so; cp cvpr2020_manet_amd/libmanet_hip_old.so /tmp/old.so
for i in $(seq $N); do
  for v in new old; do
    cp /tmp/$v.so cvpr2020_manet_amd/libmanet_hip.so
    python bench.py $ARGS --steps 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(l['value'],1),'fps kern_ms', round(l['roofline']['kernel_ms'],4))"
  done
done
cp /tmp/new.so cvpr2020_manet_amd/libmanet_hip.so
