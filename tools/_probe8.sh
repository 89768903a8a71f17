R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "" b c d; do
  for L in blobs random; do
    if [ -n "$v" ]; then export MANET_LIB_VARIANT=$R/cvpr2020_manet_amd/csrc/build_$v/libmanet_hip.so; else unset MANET_LIB_VARIANT; fi
    echo "== variant '$v' labels $L"
    timeout -k 5 300 bash tools/kstat_cmd.sh lv_${v}_$L "local_fused_kernel<12, 2>" -- tools/local_volume_bench.py --d 12 --labels $L
  done
done
unset MANET_LIB_VARIANT
timeout -k 5 120 python tools/local_timeline_vol.py 12 blobs
