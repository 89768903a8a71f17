"""Ablations of manet_frame_prepare on the GPU box (needs a -DMANET_ABLATION build): HIP-event time per launch with single
phases of the kernel switched off (manet_tune_set(MANET_TUNE_ABLATION, bits)); DESIGN.md 3.3."""
import os, sys, ctypes, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["MANET_TUNING"] = "1"
from cvpr2020_manet_amd import _lib, ops
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from frame_prep_bench import kernel_us
lib = _lib.load()
for (h, w, d, compute, st) in ((120, 214, 12, "f32", torch.float32), (120, 214, 4, "bf16", torch.bfloat16),
                              (120, 214, 12, "bf16", torch.float32)):
    e = (torch.relu(torch.randn(2, 100, h, w, device="cuda")) * 0.1).to(st)
    for abl in (0, 1, 2, 4, 8, 6, 48):
        assert lib.manet_tune_set(3, abl) == 0
        k = kernel_us(lib, lambda: ops.prepare_frames(e[0], compute=compute, max_distance=d), 2)
        print("%s/%s d=%d abl=%2d: %.1f us" % (compute, st, d, abl, k))
    lib.manet_tune_set(3, 0)
