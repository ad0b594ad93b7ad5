#!/usr/bin/env python3
"""Known-byte-count launch for calibrating FETCH_SIZE / WRITE_SIZE: one device copy of 256 MiB (read 256 MiB, write 256 MiB)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpcsd_amd import _hip

ctx = _hip.Context()
print("copy GB/s", ctx.hbm_copy_peak(1 << 28))
