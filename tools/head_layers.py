"""DynamicSegHead layer by layer on the GPU box: HIP-event time of layer 1 (shared-embedding route), layer 2, layer 4 + output
layer and the whole head at [1,100,120,214] shared + [3,3,120,214] per-object inputs, for pointwise = f32 / split."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cvpr2020_manet_amd.config import make_cfg
from cvpr2020_manet_amd.networks import IntVOS as M
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
cfg = make_cfg(["--TEST_MODE", "True"])
class Enc(torch.nn.Module):
    def forward(self, x): return x
for pw in ("f32", "split"):
    m = M.IntVOS(cfg, Enc(), pointwise=pw).cuda().eval()
    head = m.dynamic_seghead
    shared = torch.randn(1, 100, 120, 214, device="cuda")
    per = torch.randn(3, 3, 120, 214, device="cuda")
    with torch.no_grad():
        t1 = timeit(lambda: head.layer1.forward_shared(shared, per, defer_relu=True))
        x = head.layer1.forward_shared(shared, per, defer_relu=True)
        t2 = timeit(lambda: head.layer2(x, relu_in=True, defer_relu=True))
        t4 = timeit(lambda: head.layer4(x, relu_in=True, defer_relu=True, head=(head.conv.weight, head.conv.bias)))
        tall = timeit(lambda: head.forward_shared(shared, per))
    print("%s: layer1 (shared-embedding route) %.1f us, layer2 %.1f us, layer4 + output layer %.1f us, whole head %.1f us" % (pw, t1, t2, t4, tall))
