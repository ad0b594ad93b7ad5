"""NumPy model of the band reduction the register tail runs (round 5): Q^T A Q = B with half-bandwidth b, panels of b columns.

Per panel (columns j0 .. j0+b-1): Householder QR of the part of the panel BELOW the band (rows j0+b ..), reflectors V (m x b) with
the compact-WY factor T (Q_p = I - V T V^T), then the two-sided update of the trailing block in the form the kernel uses:
    X = A22 V,  Y = X T,  M = T^T (V^T X) T,  Z = Y - V M / 2,  A22 -= Z V^T + V Z^T.
The log-likelihood and the prediction then need shifted BANDED systems (lam m B + sig2 I) instead of shifted tridiagonal ones.
Validates the algebra before the HIP port (sytrd_bandtail.hpp); run as a script for a self-check."""
import numpy as np


def house(alpha, xn2):
    """H = I - tau u u^T with u = (u1, x_2, ..) un-normalised (the convention of sytrd_regtail.hpp::rt_house)."""
    s2 = alpha * alpha + xn2
    if xn2 <= 0.0 or s2 <= 1e-100:
        return 0.0, 1.0, alpha
    s = np.sqrt(s2)
    beta = -np.copysign(s, alpha)
    u1 = alpha - beta
    tau = (1.0 / s) / abs(u1)
    return tau, u1, beta


def band_reduce(A, b=4):
    A = np.array(A, dtype=np.float64, copy=True)
    n = A.shape[0]
    Vs = np.zeros((n, n))            # row k = reflector k (support: rows >= k + b)
    taus = np.zeros(n)
    j0 = 0
    while j0 + b < n - 1:
        m0 = j0 + b
        m = n - m0
        P = A[m0:, j0:j0 + b].copy()
        V = np.zeros((m, b))
        tau = np.zeros(b)
        for j in range(b):
            if m - j < 2:
                break
            x = P[j:, j]
            t, u1, beta = house(x[0], float(x[1:] @ x[1:]))
            v = x.copy()
            v[0] = u1
            if t == 0.0:
                v[:] = 0.0
                v[0] = 1.0
            V[j:, j] = v
            tau[j] = t
            for jp in range(j + 1, b):
                P[j:, jp] -= t * (v @ P[j:, jp]) * v
            P[j, j] = beta if t != 0.0 else x[0]
            if t != 0.0:
                P[j + 1:, j] = 0.0
        T = np.zeros((b, b))
        for i in range(b):
            T[i, i] = tau[i]
            if i > 0:
                T[:i, i] = -tau[i] * (T[:i, :i] @ (V[:, :i].T @ V[:, i]))
        A22 = A[m0:, m0:]
        X = A22 @ V
        Y = X @ T
        M = T.T @ (V.T @ X) @ T
        Z = Y - 0.5 * V @ M
        A22 -= Z @ V.T + V @ Z.T
        A[m0:, j0:j0 + b] = P
        A[j0:j0 + b, m0:] = P.T
        for j in range(b):
            Vs[j0 + j, m0:] = V[:, j]
            taus[j0 + j] = tau[j]
        j0 += b
    band = np.zeros((b + 1, n))      # band[j][k] = A[k + j][k]
    for j in range(b + 1):
        band[j, :n - j] = np.diagonal(A, -j)
    return band, Vs, taus, A


def q_of(Vs, taus):
    n = Vs.shape[0]
    Q = np.eye(n)
    for k in range(n):                # Q = H_0 H_1 ...
        if taus[k] != 0.0:
            Q = Q - taus[k] * np.outer(Q @ Vs[k], Vs[k])
    return Q


def band_to_dense(band):
    b1, n = band.shape
    B = np.zeros((n, n))
    for j in range(b1):
        idx = np.arange(n - j)
        B[idx + j, idx] = band[j, :n - j]
        B[idx, idx + j] = band[j, :n - j]
    return B


def banded_ldl_solve(band, lam, sig2, W):
    """(lam B + sig2 I) = L D L^T (unit lower banded L, no pivoting); returns (sum log D, x = solution for the columns of W)."""
    b = band.shape[0] - 1
    n = band.shape[1]
    a = lam * band.copy()
    a[0] += sig2
    L = np.zeros((b + 1, n))          # L[j][k] = L_{k+j, k}
    D = np.zeros(n)
    # column-oriented LDL^T on the band
    for k in range(n):
        D[k] = a[0, k]
        for j in range(1, min(b, n - 1 - k) + 1):
            L[j, k] = a[j, k] / D[k]
        for j in range(1, min(b, n - 1 - k) + 1):          # update the trailing window
            for i in range(j, min(b, n - 1 - k) + 1):
                a[i - j, k + j] -= L[i, k] * D[k] * L[j, k]
    Z = np.array(W, dtype=np.float64, copy=True)
    for k in range(n):                                       # forward
        for j in range(1, min(b, n - 1 - k) + 1):
            Z[k + j] -= L[j, k] * Z[k]
    X = Z / D[:, None] if Z.ndim == 2 else Z / D
    for k in range(n - 1, -1, -1):                           # backward
        for j in range(1, min(b, n - 1 - k) + 1):
            X[k] -= L[j, k] * X[k + j]
    return float(np.sum(np.log(D))), X, Z, D


if __name__ == "__main__":
    rs = np.random.RandomState(0)
    for n in (9, 23, 64, 190, 250):
        t = np.arange(n) * 0.4
        K = 0.5 * np.exp(-0.5 * (t[:, None] - t[None, :]) ** 2 / 20.0 ** 2) + 0.7 * np.exp(-np.abs(t[:, None] - t[None, :]) / 5.0)
        G = rs.standard_normal((n, n))
        for A in (K, G + G.T):
            band, Vs, taus, Ared = band_reduce(A, 4)
            Q = q_of(Vs, taus)
            B = band_to_dense(band)
            err = np.max(np.abs(Q.T @ A @ Q - B)) / np.max(np.abs(A))
            orth = np.max(np.abs(Q.T @ Q - np.eye(n)))
            off = np.max(np.abs(Ared - band_to_dense(band))) / np.max(np.abs(A))
            w = rs.standard_normal((n, 3))
            lam, sig2 = 3.7, 0.05
            Apd = K if A is K else A @ A.T / n
            band2, Vs2, taus2, _ = band_reduce(Apd, 4)
            Q2 = q_of(Vs2, taus2)
            ld, x, _, _ = banded_ldl_solve(band2, lam, sig2, Q2.T @ w)
            Mfull = lam * Apd + sig2 * np.eye(n)
            ld_ref = np.linalg.slogdet(Mfull)[1]
            x_ref = np.linalg.solve(Mfull, w)
            print("n=%3d  |Q^T A Q - B| %.1e  orth %.1e  outside band %.1e  logdet %.1e  solve %.1e" %
                  (n, err, orth, off, abs(ld - ld_ref) / abs(ld_ref), np.max(np.abs(Q2 @ x - x_ref)) / np.max(np.abs(x_ref))))
