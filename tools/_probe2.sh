mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_local.py -x -q -k "stored_volume or stored_volumes" > gpurun_out/probe2_tests.log 2>&1
tail -5 gpurun_out/probe2_tests.log
for args in "--d 12" "--d 4 --ids 5" "--d 12 --height 720 --width 1280 --ids 6 --pairs 30"; do
  timeout 300 python tools/local_volume_bench.py $args >> gpurun_out/probe2_bench.log 2>&1
done
cat gpurun_out/probe2_bench.log
