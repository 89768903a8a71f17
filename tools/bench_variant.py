#!/usr/bin/env python3
"""Experiments: bench.py on a VARIANT build of the library (MANET_LIB_VARIANT=path/to/libmanet_hip.so, e.g. one built with
-DMANET_ABLATION or other -D knobs).  Not a product path: the product loads cvpr2020_manet_amd/libmanet_hip.so only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr2020_manet_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(os.environ["MANET_LIB_VARIANT"])
import bench  # noqa: E402

bench.main()
