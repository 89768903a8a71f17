"""A/B on the GPU box: ops.PreparedBank.match(adaptive=True) against adaptive=False on video-like embeddings at cfg2 size --
what the adaptive policy's probes (one tiny launch into pinned host memory every 4th filtered frame) cost per match."""
import os, sys, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cvpr2020_manet_amd import ops
H, W, T, n_ids, C = 120, 214, 5, 2, 100
g = torch.Generator(device="cuda").manual_seed(1)
bank_rows = torch.relu(torch.randn(T * H * W, C, generator=g, device="cuda")) * 0.1
lab = torch.randint(0, n_ids, (T * H * W,), generator=g, device="cuda", dtype=torch.int32)
frames = [torch.relu(torch.randn(C, H, W, generator=g, device="cuda")) * 0.1 for _ in range(6)]
bank = ops.PreparedBank(bank_rows, lab, n_ids, compute="bf16r")
prep = [ops.prepare_frames(f, compute="bf16r") for f in frames]
mem = torch.ones(H * W, n_ids, device="cuda")
for rep in range(3):
    for adaptive in (False, True):
        for i in range(20):
            bank.match(prep[i % 6], normalize=True, mem=mem, adaptive=adaptive)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(100):
            bank.match(prep[i % 6], normalize=True, mem=mem, adaptive=adaptive)
        torch.cuda.synchronize()
        print("adaptive=%s: %.1f us per match" % (adaptive, (time.perf_counter() - t0) / 100 * 1e6))
