#!/usr/bin/env python3
"""Phase timeline of the local match on a STORED volume (local_fused_kernel<D, LF_VOL_IN>) -- development aid, needs the timeline
variant of the library: `make -C cvpr2020_manet_amd/csrc VAR=tl EXTRA=-DMANET_LF_TIMELINE`.  Per phase boundary: min / median / max
over the workgroups of the last launch, microseconds since the first workgroup started.
   python tools/local_timeline_vol.py [d] [labels: blobs|random]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cvpr2020_manet_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "cvpr2020_manet_amd", "csrc", "build_tl", "libmanet_hip.so")
from cvpr2020_manet_amd import ops  # noqa: E402

lib = _lib.load()
lib.manet_dbg_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
d = int(sys.argv[1]) if len(sys.argv) > 1 else 12
kind = sys.argv[2] if len(sys.argv) > 2 else "blobs"
dev = torch.device("cuda:0")
torch.manual_seed(0)
h, w, C, n_ids = 120, 214, 100, 3
embs = torch.relu(torch.randn(8, C, h, w, device=dev)) * 0.1
frames = ops.prepare_frames(embs, compute="f32", max_distance=d)
vols = ops.local_volumes(frames[:-1], frames[1:])
if kind == "random":
    lab = torch.randint(0, n_ids, (h, w), dtype=torch.int32, device=dev)
else:
    lab = torch.zeros((h, w), dtype=torch.int32, device=dev)
    lab[10:46, 30:82] = 1
    lab[50:86, 100:152] = 2
out = torch.ones((h, w, n_ids), dtype=torch.float32, device=dev)
for i in range(7):
    ops.local_match_volume(vols[i], frames[i + 1], lab, n_ids, out=out, out_is_preset=True)
torch.cuda.synchronize()
n = 8192 * 8
buf = np.zeros(n, dtype=np.uint64)
lib.manet_dbg_read(buf.ctypes.data, n)
t = buf.reshape(8192, 8).astype(np.int64)
t = t[(t[:, :7] > 0).all(axis=1)]
t = t[t[:, 0] >= t[:, 0].max() - 10 ** 6]
t0 = t[:, 0].min()
t = (t - t0) * 10.0 / 1000.0
names = ["start", "(no staging)", "labels + image DMA issued", "(no V store)", "image landed, labels / masks / tables / M2 ready",
         "items done", "out written"]
print("d = %d, %s labels, %d workgroups" % (d, kind, t.shape[0]))
for k, nm in enumerate(names):
    col = t[:, k]
    print("  %-50s min %6.2f  median %6.2f  max %6.2f us" % (nm, col.min(), np.median(col), col.max()))
print("  per-phase medians:", np.round(np.median(t[:, 1:7] - t[:, 0:6], axis=0), 2))
print("  label bytes stored (mark 7): min %6.2f  median %6.2f  max %6.2f us" % (t[:, 7].min(), np.median(t[:, 7]), t[:, 7].max()))
