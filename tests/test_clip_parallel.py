"""Frame sharding + the single bank/halo all-gather, on CPU with the gloo backend, world_size 2 and 3
(the N>1 path of bench.py; on the GPU box the same code runs over RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cvpr2020_manet_amd import clip_parallel as cp


def test_shard_frames_partition():
    for F in [1, 7, 8, 64, 65]:
        for world in [1, 2, 3, 8]:
            spans = [cp.shard_frames(F, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == F
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
            for f in range(F):
                r = cp.owner_of(f, F, world)
                assert spans[r][0] <= f < spans[r][1]


def test_bank_slots():
    slots, table = cp.bank_slots([0, 3, 9, 10, 15], 16, 4)  # ranks own 4 frames each
    assert slots == 2
    assert table == [(0, 0, 0), (0, 1, 3), (2, 0, 9), (2, 1, 10), (3, 0, 15)]
    # slab size follows bank_slots(): spread-out annotations -> 1 slot, clustered ones -> as many as one rank owns
    spread = [0, 16, 32, 48, 63]
    assert cp.slab_bytes(100, 120, 214, spread, 64, 8) == 4 * (1 * (100 * 120 * 214 + 120 * 214) + 100 * 120 * 214)
    assert cp.slab_bytes(100, 120, 214, [0, 1, 2, 3, 4], 64, 8) == 4 * (5 * (100 * 120 * 214 + 120 * 214) + 100 * 120 * 214)
    assert cp.slab_bytes(100, 120, 214, spread, 64, 8, elem_size=2) == 2 * 100 * 120 * 214 + 4 * 120 * 214 + 2 * 100 * 120 * 214
    # round-robin shipping: ceil(T / world) slots wherever the annotations sit
    slots, table = cp.bank_slots([0, 1, 2, 3, 4], 64, 8, ownership="round_robin")
    assert slots == 1 and table == [(0, 0, 0), (1, 0, 1), (2, 0, 2), (3, 0, 3), (4, 0, 4)]
    slots, table = cp.bank_slots([0, 1, 2, 3, 4], 64, 2, ownership="round_robin")
    assert slots == 3 and [t for t in table if t[0] == 1] == [(1, 0, 1), (1, 1, 3)]
    assert cp.slab_bytes(100, 120, 214, [0, 1, 2, 3, 4], 64, 8, ownership="round_robin") == \
        4 * (1 * (100 * 120 * 214 + 120 * 214) + 100 * 120 * 214)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _clip(F, C, h, w):
    g = torch.Generator().manual_seed(7)
    emb = torch.relu(torch.randn(F, C, h, w, generator=g))
    lab = torch.randint(-1, 3, (F, h, w), generator=g, dtype=torch.int32)  # includes -1 (NaN bit pattern)
    return emb, lab


def _worker(rank, world, port, F, bank_frames, q, bf16=False, ownership="block"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        C, h, w = 5, 6, 7
        emb, lab = _clip(F, C, h, w)
        if bf16:  # the producer's 2-byte storage travels bit for bit through the byte slab
            emb = emb.bfloat16()
        s, e = cp.shard_frames(F, world, rank)
        local = emb[s:e].clone()
        if ownership == "block":
            labels = {f: lab[f] for f in bank_frames if s <= f < e}  # a rank only knows its own frames' labels
            extra = None
        else:  # bank frame number j is shipped (and was extracted) by rank j % world
            ship = [f for j, f in enumerate(sorted(bank_frames)) if j % world == rank]
            labels = {f: lab[f] for f in ship}
            extra = {f: emb[f].clone() for f in ship if not (s <= f < e)}
        bank_emb, bank_lab, halo = cp.exchange_bank_and_halo(local, s, bank_frames, labels, F, ownership=ownership,
                                                             extra_embeddings=extra, timing=(rank == 0))
        info = cp.LAST_EXCHANGE
        assert info["world"] == world and info["backend"] == "gloo"
        assert info["slab_bytes"] == cp.slab_bytes(C, h, w, bank_frames, F, world, elem_size=emb.element_size(),
                                                   ownership=ownership)
        if ownership == "round_robin":
            assert info["bank_slots_per_rank"] == -(-len(bank_frames) // world)
        order = sorted(bank_frames)
        ok = torch.equal(bank_emb, emb[order]) and torch.equal(bank_lab, lab[order])
        if rank == 0 or s == 0:
            ok = ok and (halo is None or rank > 0)
        if rank > 0 and s > 0:
            ok = ok and halo is not None and torch.equal(halo, emb[s - 1])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,F,bank,bf16,ownership", [
    (2, 8, [0, 2, 5, 6, 7], False, "block"), (3, 7, [1, 3], False, "block"), (2, 3, [2], False, "block"),
    (2, 8, [0, 2, 5, 6, 7], True, "block"),
    (3, 2, [0, 1], True, "block"),         # world > frames with 2-byte storage: the empty rank must size its slab alike
    (2, 8, [0, 1, 2, 3, 7], False, "round_robin"),  # clustered annotations: slab stays ceil(T / world) slots
    (3, 9, [4, 5], True, "round_robin"),
    # BASELINE configs[3]'s partition at its real shape (world sizes above 3 had never run anywhere): 64 frames,
    # 8 per rank, a 5-frame bank spread over the clip -- one slot per rank either way; and the bank clustered in one block
    (8, 64, [0, 16, 32, 48, 63], False, "block"), (8, 64, [0, 16, 32, 48, 63], True, "round_robin"),
    (8, 64, [8, 9, 10, 11, 12], False, "round_robin"),
    # ... and a clip shorter than the node is wide: three ranks own no frame (empty blocks, halo from the last rank that has one)
    (8, 5, [0, 2, 4], False, "block"), (8, 5, [0, 1, 2, 3, 4], True, "round_robin")])
def test_exchange_bank_and_halo_gloo(world, F, bank, bf16, ownership):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, F, bank, q, bf16, ownership)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert res == [(r, True) for r in range(world)]


def _worker_gather(rank, world, port, F, bf16, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        C, h, w = 5, 6, 7
        emb, _ = _clip(F, C, h, w)
        if bf16:
            emb = emb.bfloat16()
        s, e = cp.shard_frames(F, world, rank)
        # the clip assembled from the ranks' frame blocks (SURVEY 8e's all-gather after sharded feature extraction)
        clip = cp.all_gather_clip(emb[s:e].clone(), F)
        ok = clip.dtype == emb.dtype and torch.equal(clip, emb)
        # the per-round gather of per-frame rows (the global maps of the clip-parallel propagation) to the chain rank
        L = 11
        rows = torch.arange(F * L, dtype=torch.float32).view(F, L) * 0.5 - 3.0
        got = cp.gather_frame_rows(rows[s:e].clone(), F, dst=0, timing=True)
        info = cp.LAST_GATHER
        ok = ok and info["world"] == world and info["backend"] == "gloo" and info["slab_bytes"] == -(-F // world) * L * 4
        if rank == 0:
            ok = ok and got is not None and torch.equal(got, rows) and info["gather_ms"] is not None
        else:
            ok = ok and got is None
        try:  # a block of the wrong size is an error, not a silent mis-assembly
            cp.gather_frame_rows(rows[:e - s + 1].clone(), F, dst=0)
            ok = False
        except ValueError:
            pass
        # dst=None: every rank gets the rows (two chain ranks: the forward / backward halves of the propagation)
        every = cp.gather_frame_rows(rows[s:e].clone(), F, dst=None)
        ok = ok and every is not None and torch.equal(every, rows) and cp.LAST_GATHER["dst"] is None
        # ... and the backward chain's masks go from rank 1 to rank 0 point to point
        if world > 1:
            m = (torch.arange(3 * 4 * 5).view(3, 4, 5) % 7).to(torch.int16)
            if rank == 1:
                cp.send_tensor(m, dst=0)
            elif rank == 0:
                ok = ok and torch.equal(cp.recv_tensor((3, 4, 5), torch.int16, "cpu", src=1), m)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,F,bf16", [(2, 8, False), (3, 7, True), (3, 2, False), (8, 64, False), (8, 5, True)])
def test_all_gather_clip_and_gather_frame_rows_gloo(world, F, bf16):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_gather, args=(r, world, port, F, bf16, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert res == [(r, True) for r in range(world)]


@pytest.mark.gpu
def test_collectives_over_rccl_with_one_rank():
    """The three collectives of the N-GPU flows on DEVICE tensors over the real backend ("nccl" = RCCL) with a single rank -- all a
    1-GPU box can offer: process-group init with device_id, all_gather_into_tensor of byte slabs (bank + halo, clip), dist.gather /
    all_gather of float rows, the device-event timing paths.  (N > 1 over RCCL needs a multi-GPU node: tools/preflight_multigpu.sh.)"""
    import subprocess
    import sys
    code = r'''
import os, torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="%d", RANK="0", WORLD_SIZE="1")
from cvpr2020_manet_amd import clip_parallel as cp
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
g = torch.Generator(device=dev).manual_seed(3)
F, C, h, w = 6, 100, 24, 30
emb = torch.relu(torch.randn(F, C, h, w, generator=g, device=dev))
lab = torch.randint(-1, 3, (F, h, w), generator=g, device=dev, dtype=torch.int32)
bank = [0, 3, 5]
be, bl, halo = cp.exchange_bank_and_halo(emb, 0, bank, {f: lab[f] for f in bank}, F, ownership="round_robin", timing=True)
assert torch.equal(be, emb[bank]) and torch.equal(bl, lab[bank]) and halo is None
assert cp.LAST_EXCHANGE["backend"] == "nccl" and cp.LAST_EXCHANGE["allgather_clock"] == "device events" and cp.LAST_EXCHANGE["allgather_ms"] >= 0
clip = cp.all_gather_clip(emb.bfloat16(), F)
assert clip.dtype == torch.bfloat16 and torch.equal(clip, emb.bfloat16())
rows = torch.randn(F, 77, generator=g, device=dev)
for dst in (0, None):
    got = cp.gather_frame_rows(rows, F, dst=dst, timing=True)
    assert torch.equal(got, rows) and cp.LAST_GATHER["gather_clock"] == "device events" and cp.LAST_GATHER["backend"] == "nccl"
dist.destroy_process_group()
print("rccl one rank ok")
''' % _free_port()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "rccl one rank ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
