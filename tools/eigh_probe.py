"""Repeated device eigh of a few sizes (dense random + GP kernel) for rocprofv3 --kernel-trace --stats of the eigensolver kernels."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpcsd_amd import _hip
ctx = _hip.default_context()
rs = np.random.RandomState(0)
sizes = [int(a) for a in sys.argv[1:]] or [192, 250, 384, 500]
for n in sizes:
    t = np.linspace(0, 1, n)[:, None]
    K = np.exp(-0.5 * ((t - t.T) / 0.1) ** 2) + 1e-3 * np.eye(n)
    X = rs.standard_normal((n, n)); A = X + X.T
    for M in (A, K):
        for rep in range(10):
            w, Z = ctx.eigh(M)
        wr = np.linalg.eigvalsh(M)
        sc = np.abs(wr).max()
        print("n=%d eig %.2e orth %.2e resid %.2e" % (n, np.abs(w - wr).max() / sc, np.abs(Z.T @ Z - np.eye(n)).max(), np.abs(M @ Z - Z * w).max() / sc), flush=True)
