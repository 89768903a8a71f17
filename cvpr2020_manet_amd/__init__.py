"""cvpr2020_manet_amd -- MI355X-native implementation of MANet's per-frame matching path.

What is here (and nothing else):
  csrc/            hand-written HIP kernels for gfx950 + the C ABI (include/manet_hip.h)
  _lib.py          ctypes binding of libmanet_hip.so (fails loudly if the library is missing)
  ops.py           torch-tensor front-end of the C ABI (device pointers + current HIP stream)
  config.py        the reference's `cfg` namespace (flags that steer the path)
  networks/IntVOS.py   drop-in counterpart of the reference's networks/IntVOS.py
  clip_parallel.py frame sharding of a clip over the GPUs of a node + RCCL all-gather of the bank
"""
__version__ = "0.1"
