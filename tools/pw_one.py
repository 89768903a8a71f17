"""usage: pw_one.py [rw|lds] [n]  -- n launches of the exact-fp32 1x1 layer at [3,256,120,214] (for rocprofv3 passes)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MANET_TUNING"] = "1"
from cvpr2020_manet_amd import _lib, ops
lib = _lib.load()
mode = sys.argv[1] if len(sys.argv) > 1 else "rw"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
x = torch.randn(3, 256, 120, 214, device="cuda"); w2t = torch.randn(256, 256, device="cuda") * 0.1; b2 = torch.randn(256, device="cuda")
lib.manet_tune_set(8, 1 if mode == "lds" else -2 ** 31)
abls = [int(a) for a in sys.argv[3:]] or [0]
for abl in abls:
    if abl:
        assert lib.manet_tune_set(3, abl) == 0, "needs a -DMANET_ABLATION build"
    for _ in range(5):
        ops.conv1x1_mfma(x, w2t, b2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.conv1x1_mfma(x, w2t, b2)
    e1.record()
    torch.cuda.synchronize()
    print("%s abl=%d: %.1f us" % (mode, abl, e0.elapsed_time(e1) / n * 1e3))
