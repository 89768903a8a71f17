"""Sharding a clip's frames over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference is single-GPU.  What parallelises in its propagation loop
(test.py:237-259) is the label-independent matching work: the global match of frame t needs only
(bank, embedding_t, stored map_t) and the local distance volume needs (embedding_t, embedding_t-1).
So a clip is cut into contiguous frame blocks, one per rank, and the only data every rank lacks is
  * the memory bank: the embeddings + labels of the annotated frames, which were extracted by
    whichever ranks own those frames, and
  * a one-frame halo: the embedding of the frame just before the rank's block.
Both travel in ONE all-gather (`exchange_bank_and_halo`): each rank contributes a fixed-size slab
  [ bank frames it owns (padded to the largest number any rank owns) | their labels | its last frame ]
and afterwards every rank holds the full bank and its left neighbour's last frame.  With backend
"nccl" this is a single ncclAllGather on RCCL; over 7 x ~153 GB/s xGMI links a direct all-gather of
a slab costs slab_bytes / 153 GB/s (a 480p frame is 10.3 MB -> ~0.07 ms per frame in the slab).
There is no other collective on the data path; results stay on the rank that computed them.

The same code runs on CPU tensors with the gloo backend (tests/test_clip_parallel.py).
"""
import math

import torch
import torch.distributed as dist


def shard_frames(num_frames, world_size, rank):
    """Contiguous block of frame indices [start, stop) owned by `rank` (sizes differ by <= 1)."""
    base, extra = divmod(num_frames, world_size)
    start = rank * base + min(rank, extra)
    stop = start + base + (1 if rank < extra else 0)
    return start, stop


def owner_of(frame, num_frames, world_size):
    for r in range(world_size):
        s, e = shard_frames(num_frames, world_size, r)
        if s <= frame < e:
            return r
    raise ValueError("frame %d outside clip of %d frames" % (frame, num_frames))


def bank_slots(bank_frames, num_frames, world_size):
    """How the annotated (bank) frames map onto the ranks' slabs.

    Returns (slots_per_rank, [(rank, slot, frame), ...]) with frames of one rank in ascending order.
    """
    per_rank = [[] for _ in range(world_size)]
    for f in sorted(bank_frames):
        per_rank[owner_of(f, num_frames, world_size)].append(f)
    slots = max(1, max(len(p) for p in per_rank))
    table = [(r, s, f) for r in range(world_size) for s, f in enumerate(per_rank[r])]
    return slots, table


def _all_gather_flat(slab, world, group):
    """ONE all-gather of equal-size flat slabs.  backend "nccl" (= RCCL): device buffers travel over
    xGMI directly.  gloo has no device all-gather, so device slabs are staged through host memory
    there (CPU tests, single-GPU dry runs of the N>1 flow)."""
    if dist.get_backend(group) == "gloo" and slab.is_cuda:
        host = slab.cpu()
        out = torch.empty(world * host.numel(), dtype=host.dtype)
        dist.all_gather_into_tensor(out, host, group=group)
        return out.to(slab.device)
    out = torch.empty(world * slab.numel(), dtype=slab.dtype, device=slab.device)
    dist.all_gather_into_tensor(out, slab, group=group)
    return out


def exchange_bank_and_halo(local_embeddings, local_start, bank_frames, bank_labels, num_frames,
                           group=None):
    """One all-gather that gives every rank the full memory bank and its halo frame.

    local_embeddings  [f_local, C, h, w] float32 or bfloat16 -- this rank's frames (C-major, as extract_feature
                      produces them, in the producer's storage type), frame i is clip frame local_start + i
    bank_frames       list of clip frame indices that form the memory bank (same on every rank)
    bank_labels       dict frame -> int32 [h, w] labels, needed only for frames this rank owns
    Returns (bank_emb [T, C, h, w], bank_lab [T, h, w] int32, halo [C, h, w] or None for rank 0),
    bank frames in ascending frame order.  The slab is a byte buffer: embeddings and labels travel bit for bit.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    f_local, C, h, w = local_embeddings.shape
    dev = local_embeddings.device
    dt = local_embeddings[0].dtype if f_local > 0 else torch.float32
    esz = torch.empty((), dtype=dt).element_size()
    slots, table = bank_slots(bank_frames, num_frames, world)
    frame_b = C * h * w * esz
    lab_b = h * w * 4
    slab_b = slots * (frame_b + lab_b) + frame_b
    slab = torch.zeros(slab_b, dtype=torch.uint8, device=dev)
    for (r, s, f) in table:
        if r != rank:
            continue
        emb = local_embeddings[f - local_start].contiguous().reshape(-1).view(torch.uint8)
        slab[s * frame_b:(s + 1) * frame_b] = emb
        lab = bank_labels[f].to(device=dev, dtype=torch.int32).contiguous().reshape(-1).view(torch.uint8)
        off = slots * frame_b + s * lab_b
        slab[off:off + lab_b] = lab
    if f_local > 0:
        slab[slots * (frame_b + lab_b):] = local_embeddings[f_local - 1].contiguous().reshape(-1).view(torch.uint8)
    gathered = _all_gather_flat(slab, world, group).view(world, slab_b)
    order = sorted(table, key=lambda t: t[2])
    bank_emb = torch.stack([gathered[r, s * frame_b:(s + 1) * frame_b].view(dt).view(C, h, w)
                            for (r, s, f) in order])
    bank_lab = torch.stack([gathered[r, slots * frame_b + s * lab_b:
                                     slots * frame_b + (s + 1) * lab_b].view(torch.int32).view(h, w)
                            for (r, s, f) in order])
    halo = None
    if rank > 0:
        # left neighbour with at least one frame (ranks can be empty when world > num_frames)
        for r in range(rank - 1, -1, -1):
            s0, e0 = shard_frames(num_frames, world, r)
            if e0 > s0:
                halo = gathered[r, slots * (frame_b + lab_b):].view(dt).view(C, h, w)
                break
    return bank_emb, bank_lab, halo


def slab_bytes(C, h, w, bank_frames, num_frames, world_size, elem_size=4):
    """Bytes each rank contributes to the all-gather (for the xGMI cost model in DESIGN.md).  The slab holds
    as many bank slots as the rank that owns MOST bank frames needs (bank_slots): ceil(T / world) when the
    annotated frames are spread over the clip, up to T when they cluster in one rank's block -- every rank
    then ships that many (mostly empty) slots."""
    slots, _ = bank_slots(bank_frames, num_frames, world_size)
    return slots * (C * h * w * elem_size + h * w * 4) + C * h * w * elem_size
