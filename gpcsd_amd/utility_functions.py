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


def fit_trial_shifts(evec_s, evec_t, Dvec, lfp_trials, mu_lfp, t, tau0=None, mutau=0.0, sigtau=10.0, width=None,
                     options=None):
    """Per-trial time shifts of the evoked components: the optimisation loop of auditory_lfp/fit_mean_function.py:299-335
    (one L-BFGS-B per trial over `tau`, objective = 0.5 * whitened quadratic form of `lfp_trial - mean(tau)` + Gaussian
    prior on the shifts), with every trial's optimiser running in lock-step so that ONE batched GPU call
    (`whitened_quadratic_forms`) evaluates the objectives all chains are waiting for -- the reference spreads the trials
    over CPU processes with joblib instead.

    evec_s, evec_t, Dvec: outputs of `comp_eig_D`;  lfp_trials (nx, nt, ntrials);  mu_lfp (nx, nt, nseg + 1): background in
    [..., 0], component means in [..., 1:];  t (nt,) or (nt, 1);  width: chains alive at a time (default min(ntrials, 128),
    memory-capped).  Returns (tau_hat (ntrials, nseg), success, messages)."""
    import numpy as np
    import scipy.interpolate
    import scipy.optimize
    from .lockstep import run_chains
    lfp_trials = np.atleast_3d(np.asarray(lfp_trials, dtype=np.float64))
    nx, nt, ntrials = lfp_trials.shape
    nseg = mu_lfp.shape[2] - 1
    tt = np.asarray(t, dtype=np.float64).reshape(-1)
    mu_f = [scipy.interpolate.interp1d(tt, mu_lfp[:, :, i], axis=1, fill_value="extrapolate") for i in range(1, nseg + 1)]
    tau0 = np.zeros(nseg) if tau0 is None else np.asarray(tau0, dtype=np.float64)

    def batch_fn(items):                     # items: [(trial, tau)] -> {trial: objective}
        resid = np.empty((nx, nt, len(items)))
        for j, (ti, tau) in enumerate(items):
            mu = np.array(mu_lfp[:, :, 0], dtype=np.float64, copy=True)
            for i in range(nseg):
                mu += mu_f[i](tt + tau[i])
            resid[:, :, j] = lfp_trials[:, :, ti] - mu
        quad = whitened_quadratic_forms(evec_s, evec_t, Dvec, resid)
        return {ti: float(0.5 * quad[j] + 0.5 * np.sum(np.square((np.asarray(tau) - mutau) / sigtau)))
                for j, (ti, tau) in enumerate(items)}

    def chain(ti, evaluate):
        # keyed by the trial: the evaluator hands every chain the value of ITS point of the batch
        return scipy.optimize.minimize(lambda tau: evaluate(np.array(tau, copy=True)), tau0, method="l-bfgs-b",
                                       options=options or {})
    if width is None:
        # one OS thread and one (nx, nt) residual per live chain: at most 128 chains, within a 1 GiB budget for the batch's
        # residuals; the remaining trials wait in run_chains' queue and take the slot of a chain that converges
        width = int(max(1, min(ntrials, 128, (1 << 30) // max(8 * nx * nt, 1))))
    out, _ev = run_chains(list(range(ntrials)), chain, lambda items: batch_fn([(k, x) for k, x in items]), width)
    for r in out.values():
        if isinstance(r, Exception):
            raise r
    tau_hat = np.array([out[i].x for i in range(ntrials)])
    return tau_hat, np.array([bool(out[i].success) for i in range(ntrials)]), [out[i].message for i in range(ntrials)]
