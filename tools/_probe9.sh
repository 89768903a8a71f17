export GPU_MAX_HW_QUEUES=8
timeout -k 5 900 python -m pytest tests/test_e2e_bank.py tests/test_bench_cli.py -x -q -m gpu 2>&1 | tail -4
timeout -k 5 300 python examples/propagate_clip.py --frames 32 --fused-mask-step --rounds 3 --two-streams --bank roi --session 8 --json | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:r[k] for k in ('eager_frames_per_s','first_round_frames_per_s','two_streams_frames_per_s','session_frames_per_s','session_frames_per_s_rounds_alone','local_volumes','session_ms_per_round')})"
