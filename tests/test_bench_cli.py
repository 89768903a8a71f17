"""`python bench.py --gpus N` must start N ranks by itself (VERDICT r1 weak #3 / ADVICE): checked here on
CPU -- the ranks cannot run without a GPU, but they must have been STARTED (each refuses loudly) and the
parent must pass the failure on instead of printing a 1-GPU line."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_spawns_n_ranks_and_propagates_failure():
    env = dict(os.environ, MANET_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    text = r.stdout + r.stderr
    import torch
    if torch.cuda.is_available():  # on a GPU box the dry run completes: one JSON line with n_gpus 2
        assert r.returncode == 0 and '"n_gpus": 2' in r.stdout
    else:
        assert r.returncode != 0
        assert text.count("bench.py needs an MI355X") == 2  # both ranks were started
        assert '"metric"' not in r.stdout


def test_mismatched_world_size_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr)
