"""NumPy prototype of the GPU divide-and-conquer tridiagonal eigensolver (design aid, mirrors eigh_dc.hip stage by
stage so the HIP kernels can be debugged against it).  Not part of the product or the oracle.

T = tridiag(d, e).  Cuppen tearing with all tears applied up front, leaves solved densely, merges bottom-up:
deflation scan (dlaed2-style), secular equation per root with origin shift (wave-per-root on the GPU),
Gu/Eisenstat z-hat reconstruction (Loewner), eigenvector matrix U, Z_new = Q2 @ U, final rank sort.
"""
import numpy as np

EPS = np.finfo(np.float64).eps / 2.0     # unit roundoff, LAPACK dlamch('E')


def partition(n, leaf):
    """Split [0,n) into 2^L contiguous leaves of size <= leaf (sizes differ by <= 1 at each halving)."""
    segs = [(0, n)]
    while max(b - a for a, b in segs) > leaf:
        nxt = []
        for a, b in segs:
            m = (a + b) // 2
            nxt += [(a, m), (m, b)]
        segs = nxt
    return segs


def secular_root(i, K, d, z2, rho, maxit=80):
    """Root i of 1 + rho * sum_j z2_j / (d_j - lam) in (d_i, d_{i+1}) (last: (d_{K-1}, d_{K-1} + rho*sum z2)).
    Returns (origin index, mu) with lam = d[origin] + mu; differences d_j - lam are formed as (d_j - d_origin) - mu."""
    if K == 1:
        return 0, rho * z2[0]
    last = (i == K - 1)
    if last:
        org = K - 1
        delta = d - d[org]
        lo, hi = 0.0, rho * np.sum(z2)
        # f(hi) >= 0 always; f(lo+) = -inf
    else:
        gap = d[i + 1] - d[i]
        delta0 = d - d[i]
        mid = 0.5 * gap
        fmid = 1.0 + rho * np.sum(z2 / (delta0 - mid))
        if fmid >= 0.0:
            org = i
            delta = delta0
            lo, hi = 0.0, mid
        else:
            org = i + 1
            delta = d - d[i + 1]
            lo, hi = -mid, 0.0
    # poles adjacent to the root in shifted coordinates
    if last:
        pl, pr = delta[K - 1], None
    else:
        pl, pr = delta[i], delta[i + 1]
    mu = 0.5 * (lo + hi) if not last else 0.5 * hi
    if last and mu <= 0:
        mu = hi
    for it in range(maxit):
        t = delta - mu
        terms = z2 / t
        left = slice(0, i + 1)
        right = slice(i + 1, K)
        psi = rho * np.sum(terms[left])
        dpsi = rho * np.sum(terms[left] / t[left])
        phi = rho * np.sum(terms[right]) if not last else 0.0
        dphi = rho * np.sum(terms[right] / t[right]) if not last else 0.0
        f = 1.0 + psi + phi
        err = 8.0 * EPS * (1.0 + abs(psi) + abs(phi)) + abs(mu) * EPS * (dpsi + dphi)
        if abs(f) <= err:
            break
        if f < 0:
            lo = max(lo, mu)
        else:
            hi = min(hi, mu)
        if hi - lo <= 2.0 * EPS * max(abs(lo), abs(hi)):
            break
        # rational model: psi ~ a + b/(pl - t), phi ~ c + e/(pr - t), matched in value and slope at mu
        D1 = pl - mu
        if last:
            b = dpsi * D1 * D1
            g = 1.0 + psi - dpsi * D1
            # g + b/(D1 - eta) = 0  ->  eta = D1 + b/g
            eta = D1 + b / g if g > 0 else np.inf
        else:
            D2 = pr - mu
            A = f - dpsi * D1 - dphi * D2
            B = A * (D1 + D2) + dpsi * D1 * D1 + dphi * D2 * D2
            C = D1 * D2 * f
            disc = B * B - 4.0 * A * C
            if disc < 0:
                disc = 0.0
            sq = np.sqrt(disc)
            if A == 0.0:
                eta = C / B if B != 0 else np.inf
            else:
                # two roots: pick the one keeping mu+eta inside (pl, pr)
                q = -0.5 * (-(B) + (-sq if B >= 0 else sq)) if False else None
                r1 = (B - sq) / (2.0 * A)
                r2 = (B + sq) / (2.0 * A)
                # stable forms
                if B >= 0:
                    r2 = (B + sq) / (2.0 * A)
                    r1 = (2.0 * C) / (B + sq) if (B + sq) != 0 else r1
                else:
                    r1 = (B - sq) / (2.0 * A)
                    r2 = (2.0 * C) / (B - sq) if (B - sq) != 0 else r2
                cands = [r for r in (r1, r2) if np.isfinite(r) and D1 < r < D2]
                eta = min(cands, key=abs) if cands else np.inf
        new = mu + eta
        if not np.isfinite(new) or new <= lo or new >= hi:
            # bisection; geometric when the bracket does not straddle zero and spans decades
            if lo > 0 and hi / lo > 16:
                new = np.sqrt(lo * hi)
            elif hi < 0 and lo / hi > 16:
                new = -np.sqrt(lo * hi)
            else:
                new = 0.5 * (lo + hi)
                if new == lo or new == hi:
                    break
        mu = new
    return org, mu


def merge(d1, Q1, d2, Q2, beta):
    """Merge two solved halves coupled by off-diagonal beta (diagonals already reduced by |beta|)."""
    n1, n2 = len(d1), len(d2)
    N = n1 + n2
    rho = abs(beta)
    sgn = 1.0 if beta >= 0 else -1.0
    d = np.concatenate([d1, d2])
    Q = np.zeros((N, N))
    Q[:n1, :n1] = Q1
    Q[n1:, n1:] = Q2
    z = np.concatenate([Q1[-1, :], sgn * Q2[0, :]]) / np.sqrt(2.0)
    rho2 = 2.0 * rho
    # merged ascending order (stable)
    perm = np.argsort(d, kind="stable")
    tol = 8.0 * EPS * max(np.max(np.abs(d)), np.max(np.abs(z)))
    defl = np.zeros(N, dtype=bool)
    nondefl = []
    rots = []
    if rho2 * np.max(np.abs(z)) <= tol:
        defl[:] = True
    else:
        pj = -1
        for idx in perm:
            if rho2 * abs(z[idx]) <= tol:
                defl[idx] = True
                continue
            if pj < 0:
                pj = idx
                continue
            s, c = z[pj], z[idx]
            tau = np.hypot(c, s)
            t = d[idx] - d[pj]
            c /= tau
            s = -s / tau
            if abs(t * c * s) <= tol:
                z[idx] = tau
                z[pj] = 0.0
                rots.append((pj, idx, c, s))
                tt = d[pj] * c * c + d[idx] * s * s
                d[idx] = d[pj] * s * s + d[idx] * c * c
                d[pj] = tt
                defl[pj] = True
                pj = idx
            else:
                nondefl.append(pj)
                pj = idx
        if pj >= 0:
            nondefl.append(pj)
    # column rotations (row-parallel on the GPU)
    for (a, b, c, s) in rots:
        qa, qb = Q[:, a].copy(), Q[:, b].copy()
        Q[:, a] = c * qa + s * qb
        Q[:, b] = -s * qa + c * qb
    K = len(nondefl)
    lam = np.empty(N)
    Zout = np.empty((N, N))
    if K > 0:
        nd = np.array(nondefl)
        dk = d[nd]
        zk = z[nd]
        assert np.all(np.diff(dk) > 0), "non-deflated poles must be strictly increasing"
        org = np.empty(K, dtype=int)
        mu = np.empty(K)
        for i in range(K):
            org[i], mu[i] = secular_root(i, K, dk, zk * zk, rho2)
        # Loewner / Gu-Eisenstat: zhat_i^2 = prod_j (lam_j - d_i) / (rho prod_{j != i} (d_j - d_i))
        # (lam_j - d_i) = (d_org_j - d_i) + mu_j
        diff = (dk[org][None, :] - dk[:, None]) + mu[None, :]          # [i, j] = lam_j - d_i
        dd = dk[None, :] - dk[:, None]                                  # [i, j] = d_j - d_i
        np.fill_diagonal(dd, 1.0)
        ratio = diff / dd
        zhat = np.sqrt(np.abs(np.prod(ratio, axis=1) / rho2)) * np.where(zk >= 0, 1.0, -1.0)
        U = zhat[:, None] / (-diff)                                      # zhat_i / (d_i - lam_j)
        U /= np.linalg.norm(U, axis=0, keepdims=True)
        lamk = dk[org] + mu
        W = Q[:, nd] @ U
    vals = np.concatenate([lamk if K > 0 else np.empty(0), d[defl]])
    cols = np.concatenate([W.T if K > 0 else np.empty((0, N)), Q[:, defl].T])
    order = np.argsort(vals, kind="stable")
    return vals[order], cols[order].T, K


def dc_eigh_tridiag(d, e, leaf=32):
    d = np.array(d, dtype=np.float64)
    e = np.array(e, dtype=np.float64)
    n = len(d)
    segs = partition(n, leaf)
    # all tears up front
    for (a, b) in segs[1:]:
        be = abs(e[a - 1])
        d[a - 1] -= be
        d[a] -= be
    blocks = []
    for (a, b) in segs:
        T = np.diag(d[a:b]) + np.diag(e[a:b - 1], 1) + np.diag(e[a:b - 1], -1)
        w, V = np.linalg.eigh(T)
        blocks.append((a, b, w, V))
    stats = []
    while len(blocks) > 1:
        nxt = []
        for i in range(0, len(blocks), 2):
            a, m, w1, V1 = blocks[i]
            m2, b, w2, V2 = blocks[i + 1]
            assert m == m2
            w, V, K = merge(w1, V1, w2, V2, e[m - 1])
            stats.append((b - a, K))
            nxt.append((a, b, w, V))
        blocks = nxt
    return blocks[0][2], blocks[0][3], stats


if __name__ == "__main__":
    rs = np.random.RandomState(0)
    for n in (5, 33, 64, 100, 257, 500):
        d = rs.standard_normal(n)
        e = rs.standard_normal(n - 1)
        w, V, st = dc_eigh_tridiag(d, e)
        T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        wr = np.linalg.eigvalsh(T)
        print(n, "eval err %.2e" % (np.max(np.abs(w - wr)) / np.max(np.abs(wr))),
              "orth %.2e" % np.max(np.abs(V.T @ V - np.eye(n))),
              "resid %.2e" % (np.max(np.abs(T @ V - V * w)) / np.max(np.abs(wr))), st[-1:])
