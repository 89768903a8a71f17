"""Phase timeline of local_fused_kernel (development aid): needs a build with -DMANET_LF_TIMELINE
(`make -C cvpr2020_manet_amd/csrc EXTRA=-DMANET_LF_TIMELINE`); prints, per phase boundary, the min / median / max time over
the workgroups since the first workgroup started.  ABL=<bits> sets the ablation tune key (see the kernel)."""
import ctypes, sys, torch, numpy as np
sys.path.insert(0, '.')
from cvpr2020_manet_amd import ops, _lib
lib = _lib.load()
lib.manet_dbg_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
import os
os.environ['MANET_TUNING']='1'
lib.manet_tune_set(3, int(os.environ.get('ABL','0')))
dev = torch.device('cuda:0')
torch.manual_seed(0)
for (d, nid, dt) in ((4, 4, torch.bfloat16), (12, 2, torch.float32)):
    h, w, C = 120, 214, 100
    prev = torch.randn(C, h, w, device=dev).to(dt).permute(1, 2, 0); cur = torch.randn(C, h, w, device=dev).to(dt).permute(1, 2, 0)
    lab = torch.randint(0, nid, (h, w), device=dev, dtype=torch.int32)
    for _ in range(5):
        ops.local_match(prev, cur, lab, nid, d, True)
    torch.cuda.synchronize()
    n = 8192 * 8  # (1-D grid with XCD-region padding: the workgroups that really ran are the rows with every stamp set)
    buf = np.zeros(n, dtype=np.uint64)
    lib.manet_dbg_read(buf.ctypes.data, n)
    t = buf.reshape(8192, 8).astype(np.int64)
    t = t[(t[:, :7] > 0).all(axis=1)]
    t = t[t[:, 0] >= t[:, 0].max() - 10 ** 7]  # the last launch only (100 MHz ticks: 0.1 s)
    t0 = t[:, 0].min()
    t = (t - t0) * 10.0 / 1000.0  # us (100 MHz)
    names = ['start', 'first stage in LDS', 'stage loop done', 'V stored', 'labels/tables/M2 ready', 'items done', 'out written']
    print('d =', d)
    for k, nm in enumerate(names):
        col = t[:, k]
        print('  %-26s min %6.2f  median %6.2f  max %6.2f us' % (nm, col.min(), np.median(col), col.max()))
    dur = t[:, 1:7] - t[:, 0:6]
    print('  per-phase medians:', np.round(np.median(dur, axis=0), 2))
