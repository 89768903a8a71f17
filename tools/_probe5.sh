for L in blobs random; do
  timeout -k 5 400 bash tools/kstat_cmd.sh lv_$L "local_fused|fill" -- tools/local_volume_bench.py --d 12 --labels $L
done
timeout -k 5 400 bash tools/kstat_cmd.sh lv_d4 "local_fused|fill" -- tools/local_volume_bench.py --d 4 --ids 5
