timeout -k 5 300 python -m pytest tests/test_gpu_local.py -x -q 2>&1 | tail -2
timeout -k 5 120 python tools/local_timeline_vol.py 12 blobs
timeout -k 5 120 python tools/local_timeline_vol.py 12 random
for L in blobs random; do
  timeout -k 5 400 bash tools/kstat_cmd.sh lv_$L "local_fused_kernel<12, 2>" -- tools/local_volume_bench.py --d 12 --labels $L
done
timeout -k 5 400 bash tools/kstat_cmd.sh lv_d4 "local_fused_kernel<4, 2>" -- tools/local_volume_bench.py --d 4 --ids 5
timeout -k 5 400 bash tools/kstat_cmd.sh lv_720 "local_fused_kernel<12, 2>" -- tools/local_volume_bench.py --d 12 --height 720 --width 1280 --ids 6 --pairs 30
