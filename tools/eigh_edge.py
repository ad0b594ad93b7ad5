"""Edge inputs of the device eigensolver: zero / rank-one / identity / huge-range / non-finite matrices must neither fault nor
hang; finite ones must still be accurate, non-finite ones must raise (run on the GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpcsd_amd import _hip
ctx = _hip.default_context()
rs = np.random.RandomState(3)
for n in (24, 70, 192, 250, 384, 500):
    X = rs.standard_normal((n, n)); G = X + X.T
    cases = {"zeros": np.zeros((n, n)), "ones": np.ones((n, n)), "identity": np.eye(n),
             "tiny": 1e-300 * G, "huge": 1e290 * G, "graded": np.diag(np.logspace(-300, 300, n)),
             "rank2": np.outer(X[0], X[0]) + np.outer(X[1], X[1]), "block": np.kron(np.eye(n // 2 + 1), np.array([[2.0, 1.0], [1.0, 2.0]]))[:n, :n]}
    for name, A in cases.items():
        w, Z = ctx.eigh(A)
        wr = np.linalg.eigvalsh(A)
        sc = max(np.abs(wr).max(), 1e-300)
        e1 = np.abs(w - wr).max() / sc
        e2 = np.abs(Z.T @ Z - np.eye(n)).max()
        with np.errstate(all="ignore"):
            e3 = np.abs(A / sc @ Z - Z * (w / sc)).max()
        ok = e1 < 1e-12 * n and e2 < 1e-12 * n and e3 < 1e-11 * n
        print("%4d %-9s eig %.1e orth %.1e resid %.1e %s" % (n, name, e1, e2, e3, "" if ok else "  <-- CHECK"), flush=True)
    for name, bad in (("nan", np.nan), ("inf", np.inf)):
        A = G.copy(); A[n // 3, n // 2] = A[n // 2, n // 3] = bad
        try:
            ctx.eigh(A)
            print("%4d %-9s returned without error  <-- CHECK" % (n, name))
        except np.linalg.LinAlgError as e:
            print("%4d %-9s LinAlgError (ok)" % (n, name))
    w, Z = ctx.eigh(G)                       # the context still works
    assert np.abs(w - np.linalg.eigvalsh(G)).max() < 1e-11 * n
print("done")
