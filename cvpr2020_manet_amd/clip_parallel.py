"""Sharding a clip's frames over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference is single-GPU.  What parallelises in its propagation loop
(test.py:237-259) is the label-independent matching work: the global match of frame t needs only
(bank, embedding_t, stored map_t) and the local distance volume needs (embedding_t, embedding_t-1).
So a clip is cut into contiguous frame blocks, one per rank, and the only data every rank lacks is
  * the memory bank: the embeddings + labels of the annotated frames, and
  * a one-frame halo: the embedding of the frame just before the rank's block.
Both travel in ONE all-gather (`exchange_bank_and_halo`): each rank contributes a fixed-size slab
  [ its bank slots | their labels | its last frame ]
and afterwards every rank holds the full bank and its left neighbour's last frame.  With backend
"nccl" this is a single ncclAllGather on RCCL; over 7 x ~153 GB/s xGMI links a direct all-gather of
a slab costs slab_bytes / 153 GB/s (a 480p frame is 10.3 MB -> ~0.07 ms per frame in the slab).
There is no other collective on the data path; results stay on the rank that computed them.

Who ships which bank frame (`ownership`):
  "block"        the rank whose frame block contains the annotated frame (it has the embedding anyway).
                 The slab then holds as many slots as the busiest rank owns -- up to T when the annotations
                 cluster in one block.
  "round_robin"  bank frame number j (ascending frame order) is shipped by rank j % world: the slab is
                 ALWAYS ceil(T / world) slots.  The shipping rank needs that frame's embedding: when it is
                 not inside its block the caller hands it over in `extra_embeddings` (feature extraction
                 of the T annotated frames is dealt round-robin as well -- T extra encoder calls per clip
                 over the whole node, against (T - ceil(T / world)) x 10.3 MB less in every rank's slab).

The same code runs on CPU tensors with the gloo backend (tests/test_clip_parallel.py).
"""
import time

import torch
import torch.distributed as dist

# what the last exchange_bank_and_halo() of this process did (bench.py echoes it in its JSON line so that a run with
# N ranks is self-evidencing): backend, world size seen by the collective, slab bytes, all-gather time
LAST_EXCHANGE = {}
# ... and what the last gather_frame_rows() did (the per-round gather of the clip-parallel propagation)
LAST_GATHER = {}


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


def bank_slots(bank_frames, num_frames, world_size, ownership="block"):
    """How the annotated (bank) frames map onto the ranks' slabs.

    Returns (slots_per_rank, [(rank, slot, frame), ...]) with frames of one rank in ascending order.
    """
    per_rank = [[] for _ in range(world_size)]
    if ownership == "round_robin":
        for j, f in enumerate(sorted(bank_frames)):
            per_rank[j % world_size].append(f)
    elif ownership == "block":
        for f in sorted(bank_frames):
            per_rank[owner_of(f, num_frames, world_size)].append(f)
    else:
        raise ValueError("ownership must be 'block' or 'round_robin'")
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


def _bytes(t):
    return t.contiguous().reshape(-1).view(torch.uint8)


def exchange_bank_and_halo(local_embeddings, local_start, bank_frames, bank_labels, num_frames,
                           group=None, ownership="block", extra_embeddings=None, timing=False):
    """One all-gather that gives every rank the full memory bank and its halo frame.

    local_embeddings  [f_local, C, h, w] float32 or bfloat16 -- this rank's frames (C-major, as extract_feature
                      produces them, in the producer's storage type), frame i is clip frame local_start + i.
                      A rank without frames passes an EMPTY [0, C, h, w] tensor of the clip's storage type: its
                      dtype sizes the slab (every rank's slab must have the same byte count).
    bank_frames       list of clip frame indices that form the memory bank (same on every rank)
    bank_labels       dict frame -> int32 [h, w] labels, needed only for the frames this rank ships
    ownership         "block" | "round_robin" (module docstring)
    extra_embeddings  dict frame -> [C, h, w]: embeddings of bank frames this rank ships but does not hold in its
                      block (round_robin)
    timing            synchronise around the collective and record its wall time in LAST_EXCHANGE
    Returns (bank_emb [T, C, h, w], bank_lab [T, h, w] int32, halo [C, h, w] or None for rank 0),
    bank frames in ascending frame order.  The slab is a byte buffer: embeddings and labels travel bit for bit.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    f_local, C, h, w = local_embeddings.shape
    dev = local_embeddings.device
    dt = local_embeddings.dtype  # (an empty block still carries the clip's storage type)
    esz = torch.empty((), dtype=dt).element_size()
    slots, table = bank_slots(bank_frames, num_frames, world, ownership)
    frame_b = C * h * w * esz
    lab_b = h * w * 4
    slab_b = slots * (frame_b + lab_b) + frame_b
    mine = [(s, f) for (r, s, f) in table if r == rank]  # ascending slot order

    def emb_of(f):
        if extra_embeddings is not None and f in extra_embeddings:
            e = extra_embeddings[f]
            if e.dtype != dt:
                raise ValueError("extra_embeddings[%d] is %s, the clip is stored as %s" % (f, e.dtype, dt))
            return e
        if not (local_start <= f < local_start + f_local):
            raise ValueError("rank %d ships bank frame %d but holds neither it nor an extra embedding" % (rank, f))
        return local_embeddings[f - local_start]

    # the slab as ONE concatenation of byte views (a single copy kernel on the device)
    parts = [_bytes(emb_of(f)) for (_, f) in mine]
    if len(mine) < slots:
        parts.append(torch.zeros((slots - len(mine)) * frame_b, dtype=torch.uint8, device=dev))
    parts += [_bytes(bank_labels[f].to(device=dev, dtype=torch.int32)) for (_, f) in mine]
    if len(mine) < slots:
        parts.append(torch.zeros((slots - len(mine)) * lab_b, dtype=torch.uint8, device=dev))
    parts.append(_bytes(local_embeddings[f_local - 1]) if f_local > 0
                 else torch.zeros(frame_b, dtype=torch.uint8, device=dev))
    slab = torch.cat(parts)
    assert slab.numel() == slab_b

    ag_ms, ag_clock = None, None
    device_events = timing and slab.is_cuda and dist.get_backend(group) != "gloo"
    if device_events:
        # RCCL: the collective runs on the backend's own stream, which this stream waits on -- two events on THIS stream
        # bracket it on the device's clock (host wall time around two device syncs also counted the syncs' latency)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    elif timing:
        if slab.is_cuda:
            torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
    gathered = _all_gather_flat(slab, world, group).view(world, slab_b)
    if device_events:
        e1.record()
        e1.synchronize()
        ag_ms, ag_clock = e0.elapsed_time(e1), "device events"
    elif timing:
        if slab.is_cuda:
            torch.cuda.synchronize(dev)
        ag_ms, ag_clock = (time.perf_counter() - t0) * 1e3, "host wall time (staged through host memory)" if slab.is_cuda else "host wall time"
    LAST_EXCHANGE.clear()
    LAST_EXCHANGE.update({"backend": dist.get_backend(group), "world": world, "slab_bytes": int(slab_b),
                          "gathered_bytes": int(world * slab_b), "bank_slots_per_rank": int(slots),
                          "ownership": ownership, "allgather_ms": ag_ms, "allgather_clock": ag_clock})

    # bank frames in ascending frame order: one index_select over the [world * slots] slot rows of the gathered
    # buffer (rank stride slab_b, slot stride frame_b / lab_b) -- no per-frame Python copies
    order = sorted(table, key=lambda t: t[2])
    idx = torch.tensor([r * slots + s for (r, s, f) in order], dtype=torch.long, device=gathered.device)
    emb_rows = torch.as_strided(gathered, (world, slots, frame_b), (slab_b, frame_b, 1))
    lab_rows = torch.as_strided(gathered, (world, slots, lab_b), (slab_b, lab_b, 1), slots * frame_b)
    T = len(order)
    ri, si = idx // slots, idx % slots
    bank_emb = emb_rows[ri, si].view(dt).view(T, C, h, w)  # (advanced indexing: one gather kernel, contiguous result)
    bank_lab = lab_rows[ri, si].view(torch.int32).view(T, h, w)
    halo = None
    if rank > 0:
        # left neighbour with at least one frame (ranks can be empty when world > num_frames)
        for r in range(rank - 1, -1, -1):
            s0, e0 = shard_frames(num_frames, world, r)
            if e0 > s0:
                halo = gathered[r, slots * (frame_b + lab_b):].view(dt).view(C, h, w)
                break
    return bank_emb, bank_lab, halo


def slab_bytes(C, h, w, bank_frames, num_frames, world_size, elem_size=4, ownership="block"):
    """Bytes each rank contributes to the all-gather (for the xGMI cost model in DESIGN.md).  ownership "block": the
    slab holds as many bank slots as the rank that owns MOST bank frames needs -- ceil(T / world) when the annotated
    frames are spread over the clip, up to T when they cluster in one rank's block; "round_robin": always
    ceil(T / world)."""
    slots, _ = bank_slots(bank_frames, num_frames, world_size, ownership)
    return slots * (C * h * w * elem_size + h * w * 4) + C * h * w * elem_size


def all_gather_clip(local_embeddings, num_frames, group=None):
    """Every rank contributes the embeddings of its contiguous frame block (shard_frames) and receives the whole clip
    [num_frames, C, h, w] -- SURVEY 8e's collective after sharded feature extraction: ONE all-gather of equal-size byte
    slabs (blocks differ by at most one frame; the short ones are padded).  Paid once per clip: the embeddings do not
    change between interaction rounds."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    f_local, C, h, w = local_embeddings.shape
    per = -(-num_frames // world)
    frame_b = C * h * w * local_embeddings.element_size()
    slab = torch.zeros(per * frame_b, dtype=torch.uint8, device=local_embeddings.device)
    if f_local:
        slab[:f_local * frame_b] = _bytes(local_embeddings)
    gathered = _all_gather_flat(slab, world, group).view(world, per * frame_b)
    parts = []
    for r in range(world):
        s0, e0 = shard_frames(num_frames, world, r)
        if e0 > s0:
            parts.append(gathered[r, :(e0 - s0) * frame_b])
    return torch.cat(parts).view(local_embeddings.dtype).view(num_frames, C, h, w)


def gather_frame_rows(local_rows, num_frames, dst=0, group=None, timing=False):
    """ONE gather to rank `dst` (dst=None: an all-gather, every rank gets them) of per-frame float32 rows: `local_rows`
    [f_local, L] belong to this rank's contiguous
    frame block (shard_frames); returns [num_frames, L] on `dst` (frames in clip order), None elsewhere.  The
    clip-parallel propagation ships the normalised + merged global maps of a round this way ([h*w*n_ids] = 205 KB per
    480p frame and 2 ids: 13 MB for a 64-frame clip -- 0.09 ms on one xGMI link)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    f_local, L = local_rows.shape
    s0, e0 = shard_frames(num_frames, world, rank)
    if f_local != e0 - s0:
        raise ValueError("rank %d holds %d rows for its %d frames" % (rank, f_local, e0 - s0))
    per = -(-num_frames // world)
    dev = local_rows.device
    slab = torch.zeros((per, L), dtype=torch.float32, device=dev)
    if f_local:
        slab[:f_local] = local_rows
    staged = dist.get_backend(group) == "gloo" and slab.is_cuda  # gloo has no device collectives: stage through the host
    # RCCL: the collective runs on the backend's own stream, which this stream waits on -- two events on THIS stream bracket it
    # on the device's clock, as in exchange_bank_and_halo (r4 took host wall time around two device syncs: it also counted
    # the syncs' latency)
    device_events = timing and slab.is_cuda and not staged
    if device_events:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    elif timing and slab.is_cuda:
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    send = slab.cpu() if staged else slab
    if dst is None:  # every rank gets the rows (two chain ranks: propagate_clip.py's forward / backward directions)
        bufs = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(bufs, send, group=group)
    else:
        bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, bufs, dst=dst, group=group)
    out = None
    if dst is None or rank == dst:
        parts = []
        for r in range(world):
            a, b = shard_frames(num_frames, world, r)
            if b > a:
                parts.append(bufs[r][:b - a])
        out = torch.cat(parts).to(dev)
    g_ms, g_clock = None, None
    if device_events:
        e1.record()
        e1.synchronize()
        g_ms, g_clock = e0.elapsed_time(e1), "device events"
    elif timing:
        if slab.is_cuda:
            torch.cuda.synchronize(dev)
        g_ms = (time.perf_counter() - t0) * 1e3
        g_clock = "host wall time (staged through host memory)" if slab.is_cuda else "host wall time"
    LAST_GATHER.clear()
    LAST_GATHER.update({"backend": dist.get_backend(group), "world": world, "dst": dst, "slab_bytes": int(per * L * 4),
                        "gathered_bytes": int(world * per * L * 4), "gather_ms": g_ms, "gather_clock": g_clock})
    return out


def send_tensor(t, dst, group=None):
    """point-to-point send of a device tensor (gloo: staged through the host); pairs with recv_tensor"""
    staged = dist.get_backend(group) == "gloo" and t.is_cuda
    dist.send(t.contiguous().cpu() if staged else t.contiguous(), dst=dst, group=group)


def recv_tensor(shape, dtype, device, src, group=None):
    """-> the tensor send_tensor(...) shipped from rank `src`"""
    staged = dist.get_backend(group) == "gloo" and torch.device(device).type == "cuda"
    buf = torch.empty(shape, dtype=dtype, device="cpu" if staged else device)
    dist.recv(buf, src=src, group=group)
    return buf.to(device) if staged else buf
