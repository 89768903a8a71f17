export MANET_TUNING=1 GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
timeout 600 python tools/head_overlap_probe.py > gpurun_out/probe1.log 2>&1
for t in "" "11=128" "11=128,12=17" "12=17"; do
  echo "== MANET_TUNE_INIT=$t" >> gpurun_out/probe1_e2e.log
  MANET_TUNE_INIT=$t timeout 300 python examples/propagate_clip.py --frames 31 --fused-mask-step --rounds 3 --two-streams --bank roi >> gpurun_out/probe1_e2e.log 2>&1
done
tail -30 gpurun_out/probe1.log; cat gpurun_out/probe1_e2e.log
