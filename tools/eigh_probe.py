"""Repeated device eigh of 384/500 matrices (dense random + GP kernel) for rocprofv3 --kernel-trace --stats of the eigensolver kernels."""
import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from gpcsd_amd import _hip
ctx = _hip.default_context()
rs = np.random.RandomState(0)
for n in (384, 500):
    t = np.linspace(0, 1, n)[:, None]
    K = np.exp(-0.5 * ((t - t.T) / 0.1) ** 2) + 1e-3*np.eye(n)
    X = rs.standard_normal((n, n)); A = X + X.T
    for M in (A, K):
        for rep in range(10):
            w, Z = ctx.eigh(M)
        wr = np.linalg.eigvalsh(M)
        print("done", n, np.abs(w - wr).max() / np.abs(wr).max(), np.abs(Z.T @ Z - np.eye(n)).max(), np.abs(M @ Z - Z * w).max() / np.abs(wr).max(), flush=True)
