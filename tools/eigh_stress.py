"""One-off stress of the device eigensolver over many sizes / spectra against numpy.linalg.eigh (run on the GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpcsd_amd import _hip
ctx = _hip.default_context()
rs = np.random.RandomState(0)
worst = (0, 0, 0)
def check(A, label):
    global worst
    n = A.shape[0]
    w, Z = ctx.eigh(A)
    wr = np.linalg.eigvalsh(A)
    sc = max(np.abs(wr).max(), 1e-300)
    e1 = np.abs(w - wr).max() / sc
    e2 = np.abs(Z.T @ Z - np.eye(n)).max()
    e3 = np.abs(A @ Z - Z * w[None, :]).max() / sc
    ok = e1 < 1e-13 * n and e2 < 1e-13 * n and e3 < 1e-12 * n
    worst = (max(worst[0], e1 / n), max(worst[1], e2 / n), max(worst[2], e3 / n))
    if not ok:
        print("FAIL", label, n, e1, e2, e3, flush=True)
    return ok
nfail = 0
sizes = list(range(2, 70)) + list(range(70, 260, 7)) + [191, 192, 193, 250, 255, 256, 257, 300, 383, 384, 385, 448, 500, 512, 600]
for n in sizes:
    X = rs.standard_normal((n, n)); A = X + X.T
    nfail += not check(A, "gauss")
    t = np.linspace(0, 1, n)[:, None]
    K = np.exp(-0.5 * ((t - t.T) / 0.1) ** 2) + 0.3 * np.exp(-np.abs(t - t.T) / 0.05)
    nfail += not check(K, "kernel")
    d = np.repeat(rs.standard_normal(max(n // 4, 1)), 4)[:n]; d = np.pad(d, (0, n - d.size))
    Q, _ = np.linalg.qr(rs.standard_normal((n, n)))
    nfail += not check((Q * d) @ Q.T, "degenerate")
    nfail += not check(1e8 * K + 1e-7 * np.eye(n), "scaled")
print("sizes", len(sizes), "failures", nfail, "worst (eig, orth, resid)/n", worst)
