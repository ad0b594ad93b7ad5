"""Grid helpers (host) and the Kronecker-eigen primitive (GPU).

Mirrors src/gpcsd/utility_functions.py: normalize :7-8, sort_grid :10-13, expand_grid :15-23, reduce_grid
:25-33, mykron :35-42 (API parity only -- never on the fast path), comp_eig_D :44-64 (two HIP symmetric
eigendecompositions + the D vector)."""
import numpy as np

from . import _hip


def normalize(x):
    """Scale each trial (last axis) by its max |value| over space and time."""
    x = np.asarray(x)
    return x / np.max(np.abs(x), axis=(0, 1))


def sort_grid(x):
    """Order (n, 2) points by first coordinate, ties by second (stable)."""
    x = np.asarray(x)
    by_second = x[np.argsort(x[:, 1])]
    return by_second[np.argsort(by_second[:, 0], kind="mergesort")]


def expand_grid(x1, x2):
    """All (a, b) pairs, x1-major: (len(x1)*len(x2), 2)."""
    a = np.asarray(x1, dtype=np.float64).reshape(-1)
    b = np.asarray(x2, dtype=np.float64).reshape(-1)
    return np.stack([np.repeat(a, b.size), np.tile(b, a.size)], axis=1)


def reduce_grid(x):
    """Inverse of expand_grid: sorted unique values of each column."""
    x = np.asarray(x)
    return np.unique(x[:, 0]), np.unique(x[:, 1])


def mykron(A, B):
    """Materialised Kronecker product (a1*b1, a2*b2); kept for API parity with the reference."""
    A = np.asarray(A)
    B = np.asarray(B)
    return (A[:, None, :, None] * B[None, :, None, :]).reshape(A.shape[0] * B.shape[0], A.shape[1] * B.shape[1])


def comp_eig_D(Ks, Kt, sig2n):
    """Eigenvectors of Ks and Kt and the diagonal D of kron(Ks,Kt) + sig2n*I in the Kronecker eigenbasis.

    D[x*nt + i] = evals_s[x]*evals_t[i] + sig2n   (scalar) or + sig2n[x] (per-electrode list, indexed by the
    ascending eigen-index x exactly as the reference does).  Returns (evec_s, evec_t, Dvec)."""
    return _hip.default_context().eig_D(Ks, Kt, sig2n)


def whitened_quadratic_forms(evec_s, evec_t, Dvec, resid):
    """Per-trial `sum((evec_s.T @ resid_b @ evec_t)**2 / Dvec)` for residuals `resid` of shape (nx, nt) or (nx, nt, nb),
    given the outputs of `comp_eig_D`.  Not a function of the reference package itself: it is the projection step of
    `loglik` (gpcsd1d.py:124-127) as reused by downstream per-trial objectives (auditory_lfp/fit_mean_function.py:311-321),
    batched over trials on the GPU."""
    from . import _hip
    return _hip.default_context().whitened_quad(evec_s, evec_t, Dvec, resid)
