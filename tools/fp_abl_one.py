"""usage: fp_abl_one.py ABL [f32|bf16]  -- 60 frame-prepare launches with one ablation value (for rocprofv3 --kernel-trace --stats;
needs a -DMANET_ABLATION build)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MANET_TUNING"] = "1"
from cvpr2020_manet_amd import _lib, ops
lib = _lib.load()
abl = int(sys.argv[1]); mode = sys.argv[2] if len(sys.argv) > 2 else "f32"
st = torch.float32 if mode == "f32" else torch.bfloat16
e = (torch.relu(torch.randn(100, 120, 214, device="cuda")) * 0.1).to(st)
assert lib.manet_tune_set(3, abl) == 0
for _ in range(60):
    ops.prepare_frames(e, compute=mode, max_distance=12 if mode == "f32" else 4)
torch.cuda.synchronize()
