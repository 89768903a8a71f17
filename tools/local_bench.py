#!/usr/bin/env python3
"""Local-window stage alone on synthetic 480p-grid embeddings (the configurations bench.py's cfg3 / cfg2 use):
HIP-event time per call.  Run it under rocprofv3 (tools/local_pmc.sh: the program after `--` is python3, never this
script itself) for per-kernel durations and PMC counters."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cvpr2020_manet_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    for (d, nid, dt) in ((4, 4, torch.bfloat16), (12, 2, torch.float32)):
        h, w, C = 120, 214, 100
        # [C, h, w] storage viewed as [h, w, C]: what IntVOS hands over (x.permute(1, 2, 0) of the encoder's NCHW output)
        prev = torch.randn(C, h, w, device=dev).to(dt).permute(1, 2, 0)
        cur = torch.randn(C, h, w, device=dev).to(dt).permute(1, 2, 0)
        lab = torch.randint(0, nid, (h, w), device=dev, dtype=torch.int32)
        for _ in range(5):
            ops.local_match(prev, cur, lab, nid, d, True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.local_match(prev, cur, lab, nid, d, True)
        e1.record()
        torch.cuda.synchronize()
        print("d=%d %s n_ids=%d: %.1f us per call (pooling pass + fused kernel back to back from Python: HOST-bound at this size -- read kernel durations from rocprofv3, tools/local_pmc.sh)"
              % (d, str(dt).split(".")[-1], nid, e0.elapsed_time(e1) * 1e3 / reps))
        # the prepared path of a propagation loop (r3): ONE frame prepare (query operand + pooled plane) + the fused kernel
        # on the two prepared frames -- lf_pool_pad_kernel / pack_rows_kernel<32,32> are gone from the frame
        emb_prev, emb_cur = prev.permute(2, 0, 1), cur.permute(2, 0, 1)
        fprev = ops.prepare_frames(emb_prev, compute="f32" if dt == torch.float32 else "bf16", max_distance=d)
        for _ in range(reps):
            fcur = ops.prepare_frames(emb_cur, compute="f32" if dt == torch.float32 else "bf16", max_distance=d)
            ops.local_match_frames(fprev, fcur, lab, nid)
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
