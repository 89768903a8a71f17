export GPU_MAX_HW_QUEUES=8
timeout -k 5 900 python -m pytest tests/test_e2e_bank.py tests/test_bench_cli.py tests/test_seg_head.py tests/test_frame_prepare.py tests/test_producer_layout.py -x -q -m gpu 2>&1 | tail -6
timeout -k 5 600 python bench.py > gpurun_out/bench_r06a.log 2>&1
tail -1 gpurun_out/bench_r06a.log | cut -c1-3800
