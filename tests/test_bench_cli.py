"""`python bench.py --gpus N` must start N ranks by itself (VERDICT r1 weak #3 / ADVICE): checked here on
CPU -- the ranks cannot run without a GPU, but they must have been STARTED (each refuses loudly) and the
parent must pass the failure on instead of printing a 1-GPU line."""
import os
import subprocess
import sys

import pytest

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
        # (the launcher tears the other rank down as soon as one fails: one or both refusals make it to the log -- counting two
        # was a race; that the launcher ran at all says the ranks were spawned)
        assert r.returncode != 0
        assert text.count("bench.py needs an MI355X") >= 1 and "torch.distributed" in text
        assert '"metric"' not in r.stdout


@pytest.mark.gpu
def test_two_rank_flow_on_one_gpu():
    """the whole N > 1 flow of bench.py (self-spawned ranks, byte-slab bank exchange, sharded frames, max-over-ranks
    timing) on the 1-GPU box: two ranks share the device, the collective goes over gloo.  K > 8 frames per rank
    exercises the cycled-frame view of the exchange."""
    env = dict(os.environ, MANET_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "1",
                        "--cfg", "3", "--compute", "bf16"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 10 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["cpu_baseline"] is None
    # the line says what the exchange of the timed region really was (a driver-run N-rank line shows backend nccl, world N)
    col = line["collective"]
    assert col["world"] == 2 and col["backend"] == "gloo" and col["ownership"] == "round_robin"
    assert col["slab_bytes"] > 0 and col["gathered_bytes"] == 2 * col["slab_bytes"] and col["allgather_ms"] > 0
    assert col["bank_slots_per_rank"] == 3  # cfg3: a 5-frame bank, shipped once, dealt round robin to the 2 ranks


@pytest.mark.gpu
def test_two_rank_strong_scaling_flow_on_one_gpu():
    """`--scaling strong`: a fixed 64-frame clip sharded over the ranks (32 frames each at N = 2), same exchange"""
    env = dict(os.environ, MANET_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "1",
                        "--cfg", "3", "--compute", "bf16", "--scaling", "strong"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["clip_frames"] == 64 and line["config"]["frames_per_gpu"] == 32
    assert line["collective"]["world"] == 2


@pytest.mark.gpu
def test_eight_rank_configs3_partition_on_one_gpu():
    """BASELINE configs[3] at its real shape: a 64-frame 480p clip sharded 8 frames per rank over 8 ranks --
    sharing the one GPU here, the collective over gloo; on an 8-GPU node the same command runs over RCCL -- 5-frame bank dealt
    round robin (one slot per rank), ONE all-gather in the timed region"""
    env = dict(os.environ, MANET_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--scaling", "strong", "--cfg", "2",
                        "--no-cpu-baseline", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    import json
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["value"] > 0 and line["steps"] == 8
    assert line["config"]["clip_frames"] == 64 and line["config"]["frames_per_gpu"] == 8
    col = line["collective"]
    assert col["world"] == 8 and col["backend"] == "gloo" and col["ownership"] == "round_robin"
    assert col["bank_slots_per_rank"] == 1 and col["gathered_bytes"] == 8 * col["slab_bytes"]
    # slab = one bank slot (embedding + labels) + the halo frame, fp32 at the 480p grid
    assert col["slab_bytes"] == 4 * (100 * 120 * 214 + 120 * 214) + 4 * 100 * 120 * 214
    assert len(r.stdout.strip().splitlines()[-1]) < 4096


def test_mismatched_world_size_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr)


def _run_json(cmd, env, timeout=900):
    import json
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
@pytest.mark.parametrize("world,frames", [(2, 9), (3, 5), (8, 64), (8, 5)])
def test_clip_parallel_propagation_masks_bit_equal_to_one_rank(world, frames):
    """the multi-GPU split that speeds up what test.py does.  N ranks (sharing the one GPU here, gloo)
    extract the embeddings of their frame blocks (one all-gather assembles the clip), compute the normalised + merged global
    maps of their blocks, ONE collective ships them to the chain ranks: rank 0 runs the forward half of the propagation
    (local match -> head -> mask, frame by frame), rank 1 the backward half at the same time and ships its masks to rank 0;
    rank 0 also runs the plain 1-rank loop on the same embeddings: the masks must be the same bits."""
    env = dict(os.environ, MANET_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    res = _run_json([sys.executable, os.path.join(ROOT, "examples", "propagate_clip.py"), "--gpus", str(world), "--frames",
                     str(frames), "--fused-mask-step", "--json"], env)
    assert res["world"] == world and res["frames"] == frames and res["backend"] == "gloo"
    assert res["masks_bit_equal_to_single_rank"] is True
    col = res["collective"]
    L = 120 * 214 * 3
    assert res["chain_ranks"] == [0, 1]
    assert col["world"] == world and col["dst"] is None and col["slab_bytes"] == -(-frames // world) * L * 4
    assert res["parallel_frames_per_s"] > 0 and res["single_rank_frames_per_s"] > 0
    # the flow says what bounds it -- per propagated frame the sharded part (global match), the sequential
    # chain (local match on the stored volume + head + mask step) and the speed-up no number of ranks exceeds
    assert col["chain_ranks"] == 2 and col["chain_us_per_frame"] > 0 and col["sharded_us_per_frame"] > 0
    assert col["amdahl_ceiling"] == pytest.approx(2 * (col["sharded_us_per_frame"] + col["chain_us_per_frame"])
                                                  / col["chain_us_per_frame"], rel=1e-6)
    assert 2.0 < col["amdahl_ceiling"] < 20.0 and col["measured_speedup"] > 0
    # the chain rank stored the window-distance volumes of its direction's frame pairs, once per clip
    per_pair = 240 * 107520 / 1e6
    assert res["rank0_local_volume_MB"] == pytest.approx((frames - 1 - frames // 2) * per_pair, rel=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("frames", [9, 4])
def test_two_stream_round_masks_equal_the_one_stream_loop(frames):
    """examples/propagate_clip.py --two-streams: the forward and the backward half of the chain issued alternately on two HIP
    streams of the one GPU (per-stream workspaces, disjoint frames of the shared memories): the same masks"""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    res = _run_json([sys.executable, os.path.join(ROOT, "examples", "propagate_clip.py"), "--frames", str(frames),
                     "--fused-mask-step", "--two-streams", "--rounds", "2", "--json"], env)
    assert res["two_streams_masks_equal_eager"] is True and res["two_streams_frames_per_s"] > 0


@pytest.mark.gpu
def test_two_stream_round_on_a_cold_model():
    """the two-stream round FIRST (nothing cached: the bank is built by IntVOS.prepare_bank before the streams fork, the
    memories exist before the fork), then the plain loop: the same masks; prepare_bank returns what prop_seghead then finds"""
    import torch
    from examples import propagate_clip as pc
    dev = torch.device("cuda", 0)
    args = pc.parse_args(["--frames", "7", "--fused-mask-step", "--height", "240", "--width", "428"])
    cfg, model = pc.build_model(dev, None, None, None)
    with torch.no_grad():
        emb = pc.synthetic_clip(model, dev, args.frames, args.height, args.width, args.objects, packed=True)
        clip = pc.Clip(cfg, model, emb, args.height, args.width, args.objects, fused_mask_step=True)
        two = clip.one_round_two_streams()
        torch.cuda.synchronize()
        bank = model.prepare_bank(emb[clip.start:clip.start + 1], clip.scribble, pc.SEQ, clip.gt)
        assert bank is model.prepare_bank(emb[clip.start:clip.start + 1], clip.scribble, pc.SEQ, clip.gt)  # cached
        one = clip.one_round()
        assert torch.equal(two, one)


@pytest.mark.gpu
def test_bench_e2e_line_two_ranks_on_one_gpu():
    """`bench.py --e2e --gpus 2`: the clip-parallel propagation as a bench line (strong scaling over a fixed clip), with the
    `collective` echo of the round's gather"""
    env = dict(os.environ, MANET_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    line = _run_json([sys.executable, os.path.join(ROOT, "bench.py"), "--e2e", "--gpus", "2", "--e2e-frames", "9"], env)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0 and line["unit"] == "frames/s"
    assert line["collective"]["world"] == 2 and line["e2e_parallel"]["masks_bit_equal_to_single_rank"] is True
    #  the bench line's `collective` block says what bounds the flow
    col = line["collective"]
    assert col["chain_us_per_frame"] > 0 and col["sharded_us_per_frame"] > 0 and col["chain_ranks"] == 2
    assert col["amdahl_ceiling"] > 2.0 and col["measured_speedup"] > 0
    line1 = _run_json([sys.executable, os.path.join(ROOT, "bench.py"), "--e2e", "--e2e-frames", "9"], env)
    assert line1["n_gpus"] == 1 and line1["collective"] is None and line1["value"] > 0


def test_e2e_flag_needs_a_gpu_and_spawns():
    env = dict(os.environ, MANET_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--e2e", "--gpus", "2", "--e2e-frames", "5"], env=env,
                       capture_output=True, text=True, timeout=300)
    # (the launcher tears the other rank down as soon as one fails: one or both refusals make it to the log)
    assert r.returncode != 0 and (r.stdout + r.stderr).count("bench.py needs an MI355X") >= 1 and '"metric"' not in r.stdout


def test_preflight_script_argument_plumbing():
    """tools/preflight_multigpu.sh --dry-run prints the commands it would run: RCCL with one rank, then N ranks weak / strong /
    end to end -- over RCCL when the box has the GPUs, over gloo (ranks sharing a device) when it has not"""
    sh = os.path.join(ROOT, "tools", "preflight_multigpu.sh")
    r = subprocess.run(["bash", sh, "--dry-run", "--gpus", "4"], env=dict(os.environ, PREFLIGHT_VISIBLE_GPUS="8"), cwd=ROOT,
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0
    out = r.stdout
    assert "MANET_BENCH_FORCE_DIST=1 python3 bench.py" in out
    assert "== weak_n4: python3 bench.py --gpus 4" in out and "--scaling strong" in out and "bench.py --e2e --gpus 4" in out
    assert "MANET_BENCH_BACKEND=gloo" not in out and "HSA_ENABLE_IPC_MODE_LEGACY=0" in out
    r = subprocess.run(["bash", sh, "--dry-run"], env=dict(os.environ, PREFLIGHT_VISIBLE_GPUS="1"), cwd=ROOT,
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "== weak_n2: MANET_BENCH_BACKEND=gloo python3 bench.py --gpus 2" in r.stdout
    assert subprocess.run(["bash", sh, "--bogus"], cwd=ROOT, capture_output=True).returncode == 2


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_compact_line_of_the_r04_capture_fits_4k():
    """the driver keeps an 8 KB tail; r4's ONE line was 20.6 KB and did not parse.  The compacting function on
    that very capture: < 4 KB, the contract's keys, `roofline.frac`, `cpu_baseline.value`, and summaries of every other block."""
    import json
    bench = _load_bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_line.json")))
    assert len(json.dumps(full)) > 15000
    c = bench.compact_line(full)
    text = json.dumps(c)
    assert len(text) < bench.COMPACT_LIMIT == 4096
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in c, k
    assert c["value"] == pytest.approx(full["value"], rel=1e-4) and c["config"]["workload"]
    assert c["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-4) and c["roofline"]["bound"] == "mfma"
    assert "traffic" in c["roofline"] and "traffic_source" not in c["roofline"]
    assert c["cpu_baseline"]["value"] > 0 and c["cpu_baseline"]["cores"] == 256 and c["cpu_baseline"]["kind"] == "port"
    assert len(c["cpu_baseline"]["sample"]) <= 120
    assert [a["cfg"] for a in c["also"]] == [3, 5] and all(a["frac"] > 0 and a["err_0.1"] > 0 for a in c["also"])
    # r6: a leg LEADS with the tolerance-safe mode (bf16r: the fp32 result bit for bit); the plain-bf16 rate stands beside it
    # (on this r4 capture: `value` = its exact_mode.value, `value_plain_bf16` = what r4 called `value`)
    for a, fa in zip(c["also"], full["also"]):
        assert a["value_mode"] == "bf16r" and a["value"] == pytest.approx(fa["exact_mode"]["value"], rel=1e-4)
        assert a["value_plain_bf16"] == pytest.approx(fa["value"], rel=1e-4) and a["value"] < a["value_plain_bf16"]
        assert "err_0.3" in a
    assert len(c["robustness"]["bf16r_fps"]) == 4 and c["e2e"]["value"] > 0 and c["local_stage"]["window_kernel_ms"] > 0
    # a block that outgrows the limit is dropped rather than breaking the line
    full["e2e"]["workload"] = "x" * 100
    full["also"] = full["also"] * 40
    c2 = bench.compact_line(full)
    assert len(json.dumps(c2)) <= 4096 and "roofline" in c2 and "cpu_baseline" in c2 and "also" not in c2


@pytest.mark.gpu
def test_default_command_last_line_is_compact_json():
    """the driver's command (`python bench.py --steps K --warmup W`, everything else default): the LAST stdout line is valid JSON
    under 4 KB with roofline.frac and cpu_baseline.value; the full blocks are on the `#bench_full ` line and in bench_full.json"""
    import json
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = r.stdout.strip().splitlines()
    last = lines[-1]
    assert len(last) < 4096
    line = json.loads(last)
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 0 and line["dtype"] == "f32"
    assert 0.0 < line["roofline"]["frac"] < 1.0 and line["roofline"]["unit"] == "TFLOP/s"
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1
    assert line["parity"]["global_map_max_abs_err"] < 1e-5
    assert line["e2e"]["value"] > 0 and len(line["also"]) == 2 and len(line["robustness"]["bf16r_fps"]) == 4
    for a in line["also"]:  # the bf16 configs lead with the bit-exact mode; plain bf16 and its errors at both scales beside it
        assert a["value_mode"] == "bf16r" and 0 < a["value"] and 0 < a["value_plain_bf16"]
        assert 0 < a["err_0.1"] <= 1e-3 and a["err_0.3"] > 0
    fulls = [l for l in lines if l.startswith("#bench_full ")]
    assert len(fulls) == 1
    full = json.loads(fulls[0][len("#bench_full "):])
    assert "legs" in full["robustness"] and full["value"] == pytest.approx(line["value"], rel=1e-4)
    assert json.load(open(os.path.join(ROOT, "bench_full.json")))["metric"] == line["metric"]
