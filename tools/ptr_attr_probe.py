"""How long does hipPointerGetAttributes take (ctx.hpp: host_is_pinned runs it for every host-side copy)?"""
import ctypes, time, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.set_device(0)
from gpcsd_amd import _hip
ctx = _hip.default_context()
hip = ctypes.CDLL("libamdhip64.so")
buf = ctypes.create_string_buffer(256)
pinned = _hip.pinned_pool.empty((1 << 20,))
pageable = np.zeros(1 << 20)
for name, arr in (("pinned", pinned), ("pageable", pageable)):
    p = ctypes.c_void_p(arr.ctypes.data)
    for _ in range(10):
        hip.hipPointerGetAttributes(buf, p)
    t0 = time.perf_counter()
    n = 2000
    for _ in range(n):
        rc = hip.hipPointerGetAttributes(buf, p)
    dt = (time.perf_counter() - t0) / n
    print("hipPointerGetAttributes(%s): %.2f us per call (rc %d)" % (name, 1e6 * dt, rc))
    hip.hipGetLastError()
