mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_e2e_bank.py tests/test_intvos_module.py -x -q -m gpu > gpurun_out/probe3_tests.log 2>&1
tail -8 gpurun_out/probe3_tests.log
timeout 300 python examples/propagate_clip.py --frames 31 --fused-mask-step --rounds 3 --two-streams --bank roi --session 8 --stages > gpurun_out/probe3_e2e.log 2>&1
timeout 300 python examples/propagate_clip.py --frames 31 --fused-mask-step --rounds 3 --two-streams --bank roi --session 8 --no-local-volumes >> gpurun_out/probe3_e2e.log 2>&1
cat gpurun_out/probe3_e2e.log
