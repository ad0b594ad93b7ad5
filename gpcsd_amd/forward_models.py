"""Forward-model weight functions and trapezoid CSD->LFP simulators, computed on the GPU.

Mirrors src/gpcsd/forward_models.py: `b_fwd_1d` (:9-17), `fwd_model_1d` (:20-39), `b_fwd_2d` (:42-54),
`fwd_model_2d` (:57-81).  The weight functions are HIP elementwise kernels; the simulators build the
(trapezoid weight x b_fwd) operator once and apply it with the fp64 MFMA GEMM instead of the reference's
triple Python loop (same quadrature, so results agree to rounding).
"""
import numpy as np

from . import _hip


def b_fwd_1d(r, R):
    """sqrt((r/R)^2 + 1) - sqrt((r/R)^2), elementwise over `r` (any shape)."""
    r = np.asarray(r, dtype=np.float64)
    return _hip.default_context().b_fwd_1d(r, R).reshape(r.shape)


def b_fwd_2d(delta1, delta2, R, eps, w=None):
    """log(R+eps+sqrt((R+eps)^2+w^2)) - log(eps+sqrt(eps^2+w^2)) with w = sqrt(delta1^2+delta2^2) unless given."""
    ctx = _hip.default_context()
    if w is not None:
        w = np.asarray(w, dtype=np.float64)
        return ctx.b_fwd_2d(None, None, R, eps, w=w).reshape(w.shape)
    d1, d2 = np.broadcast_arrays(np.asarray(delta1, dtype=np.float64), np.asarray(delta2, dtype=np.float64))
    return ctx.b_fwd_2d(d1, d2, R, eps).reshape(d1.shape)


def _trapz_weights(v):
    v = np.asarray(v, dtype=np.float64).reshape(-1)
    w = np.zeros_like(v)
    d = np.diff(v)
    w[:-1] += 0.5 * d
    w[1:] += 0.5 * d
    return w


def fwd_model_1d(arr, x, z, R, varsigma=1):
    """R/(2 varsigma) * trapz_x(b_fwd_1d(z_i - x, R) * arr[:, t]) for every prediction site z_i and time t.

    arr (nx, nt) on the dense grid x (nx, 1); returns (nz, nt)."""
    arr = np.asarray(arr, dtype=np.float64)
    xs = np.asarray(x, dtype=np.float64).reshape(-1)
    zs = np.asarray(z, dtype=np.float64).reshape(-1)
    B = b_fwd_1d(zs[:, None] - xs[None, :], R) * _trapz_weights(xs)[None, :]
    return (R / (2.0 * varsigma)) * _hip.default_context().gemm(B, arr)


def fwd_model_2d(arr, x1, x2, z, R, eps, varsigma=1):
    """Double trapezoid of b_fwd_2d(z_i - (x1, x2)) * arr[:, :, t]; arr (nx1, nx2, nt), z (nz, 2) -> (nz, nt).
    (`varsigma` is accepted and unused, as in the reference.)"""
    arr = np.asarray(arr, dtype=np.float64)
    x1 = np.asarray(x1, dtype=np.float64).reshape(-1)
    x2 = np.asarray(x2, dtype=np.float64).reshape(-1)
    z = np.asarray(z, dtype=np.float64)
    d1 = z[:, 0][:, None, None] - x1[None, :, None]
    d2 = z[:, 1][:, None, None] - x2[None, None, :]
    W = b_fwd_2d(d1, d2, R, eps) * _trapz_weights(x1)[None, :, None] * _trapz_weights(x2)[None, None, :]
    return _hip.default_context().gemm(W.reshape(z.shape[0], -1), arr.reshape(x1.size * x2.size, -1))
