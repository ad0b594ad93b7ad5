"""Randomised differential test of the fused calls against the oracle (run on the GPU box):
random 1D / 2D geometries (symmetric, perturbed, odd sizes, above and below the 64-row Jacobi limit), random well-scaled
hyper-parameters, scalar or per-electrode noise, predictions at the electrodes / at other sites, t* = t or shifted.
The oracle is test infrastructure; this tool never ships.   python tools/fuzz_models.py [ncases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import gpcsd_oracle as O
from gpcsd_amd.gpcsd1d import GPCSD1D
from gpcsd_amd.gpcsd2d import GPCSD2D
from gpcsd_amd.covariances import GPCSDTemporalCovSE, GPCSDTemporalCovMatern
SE, MATERN = 0, 1

def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))

def one_case(rs, k):
    dim = int(rs.choice([1, 2]))
    nt = int(rs.choice([7, 33, 64, 65, 90, 131] if not os.environ.get("FUZZ_BIG") else [65, 131, 250, 387, 450, 500, 511]))
    R = int(rs.randint(1, 4))
    t = np.arange(nt, dtype=np.float64)[:, None] * float(rs.uniform(0.3, 1.5))
    if rs.rand() < 0.25:
        t = t + np.cumsum(rs.uniform(0, 0.2, nt))[:, None]            # non-uniform time grid: no temporal symmetry
    if dim == 1:
        nx = int(rs.choice([5, 24, 63, 65, 81, 100] if not os.environ.get("FUZZ_BIG") else [24, 100, 193, 256, 300]))
        x = np.linspace(0.0, 30.0 * nx, nx)[:, None]
        if rs.rand() < 0.3:
            x = x + rs.uniform(-5, 5, (nx, 1)); x = np.sort(x, axis=0)  # perturbed: not mirror-symmetric
        a, b = float(x.min()), float(x.max())
        ngl = int(rs.choice([31, 60]))
        ell_s = (float(rs.uniform(60, 200)),)
        eps = 0.0
    else:
        n1 = int(rs.choice([2, 3, 4])); n2 = int(rs.choice([6, 17, 24, 33]))
        x1 = np.linspace(0.0, 16.0 * (n1 - 1), n1); x2 = np.linspace(0.0, 20.0 * (n2 - 1), n2)
        x = np.stack([np.repeat(x1, n2), np.tile(x2, n1)], axis=1)
        kind = rs.rand()
        if kind < 0.25:
            x = x[rs.permutation(x.shape[0])[: max(5, x.shape[0] - 3)]]   # drop sites: usually asymmetric
        elif kind < 0.5:
            x = x[((np.arange(x.shape[0]) // n2 + np.arange(x.shape[0]) % n2) % 2 == 0)]   # checkerboard: point symmetry at best
        nx = x.shape[0]
        ngl = None
        ell_s = (float(rs.uniform(20, 60)), float(rs.uniform(60, 200)))
        eps = float(rs.uniform(10, 60))
    Rv = float(rs.uniform(50, 150))
    C_ = int(rs.choice([1, 2]))
    temporal = [(int(rs.choice([SE, MATERN])), float(rs.uniform(2, 12)), float(rs.uniform(0.3, 1.5))) for _ in range(C_)]
    lfp = rs.standard_normal((nx, nt, R))
    tcl = []
    for kd, ell, s2 in temporal:
        tc = GPCSDTemporalCovSE(t) if kd == SE else GPCSDTemporalCovMatern(t)
        tc.params["ell"]["value"] = ell; tc.params["sigma2"]["value"] = s2
        tcl.append(tc)
    if dim == 1:
        m = GPCSD1D(lfp, x, t, a=a, b=b, ngl=ngl, temporal_cov_list=tcl)
        m.spatial_cov.params["ell"]["value"] = ell_s[0]
        geom = O.Geometry1D(x, t, a=a, b=b, ngl=ngl)
    else:
        m = GPCSD2D(lfp, x, t, ngl1=8, ngl2=20, temporal_cov_list=tcl, eps=eps)
        m.spatial_cov.params["ell1"]["value"] = ell_s[0]; m.spatial_cov.params["ell2"]["value"] = ell_s[1]
        geom = O.Geometry2D(x, t, ngl1=8, ngl2=20)
    m.R["value"] = Rv
    # forward weights are un-normalised: scale the noise with the signal so the problem stays well conditioned
    Ks0 = O.spatial_kphi(geom, O.make_hparams(Rv, ell_s, temporal, 1.0, eps=eps))
    scale = float(np.mean(np.diag(Ks0))) * float(np.mean([s2 for _, _, s2 in temporal]))
    if rs.rand() < 0.2:
        sig = list(scale * rs.uniform(0.02, 0.2, nx))
    else:
        sig = scale * float(rs.uniform(0.02, 0.2))
    m.sig2n["value"] = sig
    hp = O.make_hparams(Rv, ell_s, temporal, np.asarray(sig) if np.ndim(sig) else sig, eps=eps, jitter=m.JITTER)
    hp0 = dict(hp); hp0["jitter"] = 0.0
    desc = "dim%d nx=%d nt=%d R=%d C=%d siglist=%d" % (dim, nx, nt, R, C_, int(np.ndim(sig) > 0))
    ll = float(m.loglik()); llo = float(O.loglik(geom, hp, lfp))
    e_ll = abs(ll - llo) / abs(llo)
    # predictions
    zmode = rs.rand()
    if zmode < 0.5: z = x
    elif zmode < 0.75: z = x[::2]
    else: z = x[: max(3, nx - 2)] + (0.37 if dim == 1 else np.array([0.37, -0.21]))
    ts = t if rs.rand() < 0.7 else t + 0.31
    m.predict(z, ts, type="both")
    ref = O.predict(geom, hp0, lfp, z, ts, type="both")
    e_c, e_l = relerr(m.csd_pred, ref["csd"]), relerr(m.lfp_pred, ref["lfp"])
    folds = m._context().fold_gemm()
    if np.ndim(sig) > 0:
        # per-electrode noise list: the objective depends on the ORDER of the (near-)degenerate eigenvalues of Ks, i.e. on
        # the rounding of the eigensolver; measure how much equally valid LAPACK drivers disagree on THIS draw
        sp = (O.driver_spread(lambda: O.loglik(geom, hp, lfp)),
              O.driver_spread(lambda: O.predict(geom, hp0, lfp, z, ts, type="csd")["csd"]),
              O.driver_spread(lambda: O.predict(geom, hp0, lfp, z, ts, type="lfp")["lfp"]))
        one_case.last_spread = sp
    else:
        one_case.last_spread = (0.0, 0.0, 0.0)
    if os.environ.get("FUZZ_GRAD"):
        # analytic gradient (natural parameters) against central differences of the library's own loglik
        ll0, g = m._loglik_and_grad_natural()
        slots = m._param_slots()
        fd = np.zeros(len(slots) + 1)
        for i, (getter, setter, _, _, _) in enumerate(slots):
            v = getter(); h = 1e-6 * abs(v)
            setter(v + h); lp = float(m.loglik()); setter(v - h); lm = float(m.loglik()); setter(v)
            fd[i] = (lp - lm) / (2 * h)
        if m._sig2n_is_scalar():
            v = m.sig2n["value"]; h = 1e-6 * abs(v)
            m.sig2n["value"] = v + h; lp = float(m.loglik()); m.sig2n["value"] = v - h; lm = float(m.loglik()); m.sig2n["value"] = v
            fd[-1] = (lp - lm) / (2 * h)
            gg = np.asarray(g)[: len(slots) + 1]
        else:
            fd = fd[:-1]; gg = np.asarray(g)[: len(slots)]
        e_g = float(np.max(np.abs(gg - fd)) / max(np.max(np.abs(fd)), 1e-300))
        return desc, e_ll, max(e_c, 0.0), e_l, folds, e_g
    return desc, e_ll, e_c, e_l, folds

if __name__ == "__main__":
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rs = np.random.RandomState(seed)
    worst = [0.0, 0.0, 0.0]; nbad = 0; t0 = time.time(); nfold = 0
    for k in range(ncases):
        try:
            out = one_case(rs, k)
            desc, e_ll, e_c, e_l, folds = out[:5]
            if len(out) > 5:
                # with a per-electrode noise list the objective depends on the ORDER of (near-)degenerate eigenvalues of a
                # numerically rank-deficient Ks: it is not smooth there and neither the analytic term nor differences mean much
                if "siglist=1" in desc:
                    continue
                worst_g = max(globals().get("worst_g", 0.0), out[5]); globals()["worst_g"] = worst_g
                if out[5] > 2e-4:
                    print("%3d %-44s GRADIENT vs central differences %.1e  <-- CHECK" % (k, desc, out[5]), flush=True)
        except ZeroDivisionError:            # degenerate grid for the default priors (one distinct site spacing): the reference
            continue                         # divides by zero in set_params as well
        nfold += folds > 0
        # a per-electrode noise list is attached to eigen-RANKS (reference quirk): with near-degenerate tiny eigenvalues the
        # assignment depends on rounding, two correct solvers differ at 1e-6..1e-5.  The yardstick is the spread between three
        # LAPACK drivers on the same draw -- an underestimate, since dsyevd / dsyev / dsyevr share dsytrd's tridiagonal
        # matrix and differ only behind it, while this solver tridiagonalises in another order: over 340 draws (seeds 11, 12)
        # it sat within 3x of that spread in all but 4 noise-list cases, the largest ratio 6.3 (3.1e-6 against 4.9e-7) --
        # identically with the single-stream, graph-less path, so rounding, not scheduling.  Gate: 10x.
        sp = one_case.last_spread
        gates = [max(1e-6, 10.0 * v) for v in sp]
        bad = e_ll > gates[0] or e_c > gates[1] or e_l > gates[2]
        if "siglist=1" in desc:
            print("    noise list: LAPACK driver spread ll %.1e csd %.1e lfp %.1e | GPU vs dsyevd ll %.1e csd %.1e lfp %.1e"
                  % (sp[0], sp[1], sp[2], e_ll, e_c, e_l), flush=True)
        nbad += bad
        worst = [max(worst[0], e_ll), max(worst[1], e_c), max(worst[2], e_l)]
        if bad or k % 10 == 0:
            print("%3d %-44s ll %.1e csd %.1e lfp %.1e folded_calls %d %s" % (k, desc, e_ll, e_c, e_l, folds, "<-- FAIL" if bad else ""), flush=True)
    print("cases %d failures %d (gate 1e-6; with a noise list 10x the spread between LAPACK drivers on the same draw) worst ll %.1e csd %.1e lfp %.1e; cases that used the folded path: %d; %.0f s"
          % (ncases, nbad, worst[0], worst[1], worst[2], nfold, time.time() - t0))
    if "worst_g" in globals():
        print("worst gradient deviation from central differences (relative to the largest component): %.1e" % globals()["worst_g"])
