"""Shared machinery of GPCSD1D / GPCSD2D: hyper-parameter (un)packing, device residency, loglik / predict /
sample_prior on the GPU, and the multi-restart MAP fit.

Behavioural contract (reference: src/gpcsd/gpcsd1d.py, src/gpcsd/gpcsd2d.py):
  * hyper-parameters live in mutable dicts (`self.R`, `self.sig2n`, `spatial_cov.params`, `temporal_cov_list[i].params`)
    and are re-read at every call;
  * loglik adds JITTER*I to Ks, predict does not; neither adds the -N/2 log(2 pi) constant;
  * fit optimises -(loglik + log-prior) over log-parameters with L-BFGS-B from prior-sampled restarts and keeps the
    best finite optimum; the reference's autograd gradient is replaced by an analytic gradient computed on the GPU
    (scalar sig2n or a per-electrode list); restarts can run concurrently (`workers`) and be sharded over ranks;
  * predict(z, t, type) fills csd_pred / csd_pred_list / lfp_pred / lfp_pred_list / t_pred / x_pred.
"""
import numpy as np
import scipy.optimize
from tqdm import tqdm

from . import _hip
from .covariances import GPCSDTemporalCov


def _same_array(a, b):
    return a is b or (np.shape(a) == np.shape(b) and np.array_equal(a, b))


class GPCSDModel:
    """Base of GPCSD1D / GPCSD2D.  Subclasses set `dim`, `JITTER`, `_spatial_names` and build the covariances."""

    dim = None
    JITTER = None
    _spatial_names = ()          # names of the spatial length-scale params in spatial_cov.params

    # ------------------------------------------------------------------ device residency
    def _context(self):
        ctx = getattr(self, "_ctx", None)
        if ctx is None:
            ctx = _hip.Context(getattr(self, "_device", None))
            self._ctx = ctx
            self._resident = {}
        return ctx

    def set_device(self, device):
        """Pin this model to a GPU ordinal (default: $LOCAL_RANK or 0).  Must be called before the first evaluation."""
        if getattr(self, "_ctx", None) is not None:
            raise RuntimeError("device already initialised for this model")
        self._device = int(device)

    def shard_trials(self, sharding):
        """Evaluate only this rank's contiguous block of trials and combine partial sums across ranks
        (gpcsd_amd.dist.TrialSharding).  `self.lfp` keeps the full array; predictions are returned for the local block
        unless `sharding.gather_predictions` is set."""
        self._sharding = sharding
        self._resident = {} if getattr(self, "_ctx", None) is not None else getattr(self, "_resident", {})

    def invalidate(self):
        """Force re-upload of lfp / coordinates at the next call (use after editing arrays in place)."""
        self._resident = {}

    def _local_lfp(self):
        lfp = np.atleast_3d(self.lfp)
        sh = getattr(self, "_sharding", None)
        if sh is None:
            return lfp
        return lfp[:, :, sh.local_slice(lfp.shape[2])]

    # "fp32 kernel build + fp64 factor" (BASELINE cfg5): set `model.gram_precision = 32` to evaluate the Gram builders of
    # this model's fused calls in single precision; everything after the Gram matrices stays fp64.  Default 64 = the reference.
    gram_precision = 64

    def _sync_device(self, need_lfp=True):
        ctx = self._context()
        res = self._resident
        sc = self.spatial_cov
        if res.get("gram_precision") != self.gram_precision:
            ctx.set_gram_precision(self.gram_precision)
            res["gram_precision"] = self.gram_precision
        # time grid: the reference evaluates each temporal covariance on its own `t`; they must agree
        t = self.temporal_cov_list[0].t
        for tc in self.temporal_cov_list[1:]:
            if not _same_array(tc.t, t):
                raise ValueError("temporal covariance components hold different time grids")
        if "t" not in res or not _same_array(res["t"], t):
            ctx.set_time(t)
            res["t"] = np.array(t, dtype=np.float64, copy=True)
        if self.dim == 1:
            key = (np.asarray(sc.x, dtype=np.float64).reshape(-1), np.asarray(sc.gl_x), np.asarray(sc.gl_w))
        else:
            key = (np.asarray(sc.x, dtype=np.float64), np.asarray(sc.gl_x1), np.asarray(sc.gl_w1), np.asarray(sc.gl_x2),
                   np.asarray(sc.gl_w2))
        old = res.get("geo")
        if old is None or len(old) != len(key) or not all(_same_array(a, b) for a, b in zip(old, key)):
            if self.dim == 1:
                ctx.set_geometry_1d(*key)
            else:
                ctx.set_geometry_2d(*key)
            res["geo"] = tuple(np.array(k, copy=True) for k in key)
        if need_lfp:
            lfp = self.lfp
            if np.ndim(lfp) != 3:
                raise ValueError("lfp must have shape (n_spatial, n_time, n_trials)")
            sh = getattr(self, "_sharding", None)
            ident = (id(lfp), lfp.__array_interface__["data"][0], lfp.shape, lfp.strides,
                     None if sh is None else (sh.rank, sh.world_size))
            if res.get("lfp") != ident:
                ctx.set_lfp(self._local_lfp())
                res["lfp"] = ident
        return ctx

    # ------------------------------------------------------------------ hyper-parameters
    def _temporal_triplets(self):
        """[(kind, ell, sigma2)] per temporal component.  SE / Matern are evaluated by the library's own Gram builders; any
        other object with a `compute_Kt` method (the reference accepts those: covariances.py:235-238, gpcsd1d.py:118-120)
        is marked KIND_HOST and its Gram matrices are evaluated on the host by that method (see `_hparams`)."""
        out = []
        for tc in self.temporal_cov_list:
            kind = getattr(tc, "kind", None)
            if isinstance(tc, GPCSDTemporalCov) and kind in (_hip.KIND_SE, _hip.KIND_MATERN):
                out.append((kind, tc.params["ell"]["value"], tc.params["sigma2"]["value"]))
            elif callable(getattr(tc, "compute_Kt", None)):
                out.append((_hip.KIND_HOST, 0.0, 0.0))
            else:
                raise TypeError("temporal covariance %r has no compute_Kt method" % type(tc).__name__)
        return out

    def _uses_host_kt(self):
        return any(k == _hip.KIND_HOST for k, _, _ in self._temporal_triplets())

    def _host_kt_is_differentiable(self):
        """Every temporal component of a model with user-defined covariances offers compute_dKt(name)."""
        return all(callable(getattr(tc, "compute_dKt", None)) for tc in self.temporal_cov_list)

    def _hparams(self, jitter, tstar=None, want_dkt=False):
        ell_s = [self.spatial_cov.params[n]["value"] for n in self._spatial_names]
        eps = getattr(self, "eps", 0.0)
        ctx = self._context()
        trip = self._temporal_triplets()
        if any(k == _hip.KIND_HOST for k, _, _ in trip):
            # user-defined temporal covariance: its Gram matrices come from its own compute_Kt (host NumPy), everything after
            # them -- eigensolver, projections, predict chain -- still runs on the GPU
            Kt = sum(np.asarray(tc.compute_Kt(), dtype=np.float64) for tc in self.temporal_cov_list)
            cross = None
            if tstar is not None:
                cross = np.stack([np.asarray(tc.compute_Kt(tstar), dtype=np.float64) for tc in self.temporal_cov_list])
            ctx.set_host_temporal_gram(Kt, cross)
            if want_dkt:                             # d Kt / d (ell_c, sigma2_c) per component, from the objects themselves
                ctx.set_host_temporal_dgram(np.stack([np.asarray(tc.compute_dKt(nm), dtype=np.float64)
                                                      for tc in self.temporal_cov_list for nm in ("ell", "sigma2")]))
            self._resident["host_kt"] = True
        elif self._resident.get("host_kt"):
            ctx.set_host_temporal_gram(None)
            self._resident["host_kt"] = False
        hp, keep = ctx.make_hparams(self.R["value"], eps, ell_s, trip, self.sig2n["value"], jitter)
        return hp, keep

    def _sig2n_is_scalar(self):
        return np.isscalar(self.sig2n["value"]) or np.ndim(self.sig2n["value"]) == 0

    def extract_model_params(self):
        p = {"R": self.R["value"]}
        if self.dim == 2:
            p["eps"] = self.eps
        p["sig2n"] = self.sig2n["value"]
        if self.dim == 1:
            p["spatial_ell"] = self.spatial_cov.params["ell"]["value"]
        else:
            p["spatial_ell1"] = self.spatial_cov.params["ell1"]["value"]
            p["spatial_ell2"] = self.spatial_cov.params["ell2"]["value"]
        p["temporal_ell_list"] = [tc.params["ell"]["value"] for tc in self.temporal_cov_list]
        p["temporal_sigma2_list"] = [tc.params["sigma2"]["value"] for tc in self.temporal_cov_list]
        return p

    def restore_model_params(self, params):
        self.R["value"] = params["R"]
        if self.dim == 2:
            self.eps = params["eps"]
        self.sig2n["value"] = params["sig2n"]
        if self.dim == 1:
            self.spatial_cov.params["ell"]["value"] = params["spatial_ell"]
        else:
            self.spatial_cov.params["ell1"]["value"] = params["spatial_ell1"]
            self.spatial_cov.params["ell2"]["value"] = params["spatial_ell2"]
        if len(self.temporal_cov_list) != len(params["temporal_ell_list"]):
            print("different number of temporal covariance functions! stopping.")
            return
        for tc, ell, s2 in zip(self.temporal_cov_list, params["temporal_ell_list"], params["temporal_sigma2_list"]):
            tc.params["ell"]["value"] = ell
            tc.params["sigma2"]["value"] = s2

    def _describe(self, header_lines):
        s = "GPCSD1D object\n"          # (the reference prints this header for both classes)
        s += "LFP shape: (%d, %d, %d)\n" % tuple(np.shape(self.lfp)[:3])
        for line in header_lines:
            s += line
        s += "R parameter prior: %s\n" % str(self.R["prior"])
        s += "R parameter value %0.4g\n" % self.R["value"]
        return s

    def _describe_temporal(self):
        s = ""
        for i, tc in enumerate(self.temporal_cov_list):
            s += "Temporal covariance %d class name: %s\n" % (i + 1, type(tc).__name__)
            s += "Temporal covariance %d ell prior: %s\n" % (i + 1, str(tc.params["ell"]["prior"]))
            s += "Temporal covariance %d ell value %0.4g\n" % (i + 1, tc.params["ell"]["value"])
            s += "Temporal covariance %d sigma2 prior: %s\n" % (i + 1, str(tc.params["sigma2"]["prior"]))
            s += "Temporal covariance %d sigma2 value %0.4g\n" % (i + 1, tc.params["sigma2"]["value"])
        return s

    # ------------------------------------------------------------------ likelihood
    def loglik(self):
        """Log marginal likelihood (up to the 2*pi constant) of all trials under the current hyper-parameters."""
        ctx = self._sync_device()
        hp, _keep = self._hparams(self.JITTER)
        sumlog, quad = ctx.loglik_parts(hp)
        sh = getattr(self, "_sharding", None)
        if sh is not None:
            quad = float(sh.allreduce_sum(np.array([quad]))[0])
        ntrials = np.shape(self.lfp)[2]
        return np.float64(-0.5 * ntrials * sumlog - 0.5 * quad)

    def _loglik_and_grad_natural(self):
        """(loglik, d loglik / d[R, ell_s.., (ell_t, sigma2_t).., sig2n or sig2n_0..sig2n_{nx-1}]) on the GPU."""
        host = self._uses_host_kt()
        if host and not self._host_kt_is_differentiable():
            raise NotImplementedError("no analytic gradient for user-defined temporal covariances without compute_dKt(name)")
        ctx = self._sync_device()
        hp, _keep = self._hparams(self.JITTER, want_dkt=host)
        nsig = 1 if self._sig2n_is_scalar() else len(self.sig2n["value"])
        ng = 1 + self.dim + 2 * len(self.temporal_cov_list) + nsig
        sumlog, quad, g = ctx.loglik_grad(hp, ng)
        r_local = self._local_lfp().shape[2]
        ll = -0.5 * r_local * sumlog - 0.5 * quad
        sh = getattr(self, "_sharding", None)
        if sh is not None:                       # both pieces are additive over shards
            red = sh.allreduce_sum(np.concatenate([[ll], g]))
            ll, g = float(red[0]), red[1:]
        return ll, g

    def _eval_batch_local(self, hps):
        """(loglik of the LOCAL trials [B], natural gradient [B, ng], status [B]) for a list of hyper-parameter structs: one
        shared chain of launches (gpcsd_loglik_grad_batch)."""
        ctx = self._sync_device()
        nsig = 1 if self._sig2n_is_scalar() else len(self.sig2n["value"])
        ng = 1 + self.dim + 2 * len(self.temporal_cov_list) + nsig
        sumlog, quad, g, st = ctx.loglik_grad_batch(hps, ng)
        r_local = self._local_lfp().shape[2]
        return -0.5 * r_local * sumlog - 0.5 * quad, g, st

    def _eval_batch_reduced(self, hps):
        """(loglik [B], natural gradient [B, ng], failed [B]) of a batch over ALL trials: the local evaluation, and under trial
        sharding ONE all-reduce for the whole batch.  Slot b must mean the same hyper-parameter set on every rank (the
        lock-step drivers hand their batches over sorted by restart index for exactly this reason)."""
        ll, g, st = self._eval_batch_local(hps)
        ll, g, st = np.asarray(ll, dtype=np.float64), np.asarray(g, dtype=np.float64), np.asarray(st)
        B, ng = g.shape
        sh = getattr(self, "_sharding", None)
        if sh is not None:                       # both pieces are additive over shards; a set that failed anywhere fails everywhere
            red = sh.allreduce_sum(np.concatenate([ll, g.ravel(), (st != 0).astype(np.float64)]))
            ll, g, st = red[:B], red[B:B + B * ng].reshape(B, ng), red[B + B * ng:]
        return ll, g, st != 0

    @staticmethod
    def _batch_failure(b):
        return np.linalg.LinAlgError("numerical failure in the eigensolver (hyper-parameter set %d of the batch)" % b)

    def _loglik_and_grad_natural_batch(self, hps):
        """[(loglik, gradient) or LinAlgError] for a batch of hyper-parameter structs."""
        ll, g, failed = self._eval_batch_reduced(hps)
        return [self._batch_failure(b) if failed[b] else (float(ll[b]), np.array(g[b])) for b in range(len(ll))]

    # ------------------------------------------------------------------ fit
    def _param_slots(self):
        """[(getter, setter, prior, (min, max), scale)] in the reference's log-parameter order."""
        slots = []

        def dict_slot(d, scale):
            return (lambda: d["value"], lambda v: d.__setitem__("value", v), d["prior"], (d["min"], d["max"]), scale)
        slots.append(dict_slot(self.R, 100.0))
        for n in self._spatial_names:
            slots.append(dict_slot(self.spatial_cov.params[n], 100.0))
        for tc in self.temporal_cov_list:
            slots.append(dict_slot(tc.params["ell"], 1.0))
            slots.append(dict_slot(tc.params["sigma2"], 1.0))
        return slots

    def _bounds(self):
        b = []
        with np.errstate(divide="ignore"):
            for _, _, _, (lo, hi), scale in self._param_slots():
                b.append((np.log(lo / scale), np.log(hi / scale)))
            if self._sig2n_is_scalar():
                b.append((np.log(self.sig2n["min"]), np.log(self.sig2n["max"])))
            else:
                for lo, hi in zip(self.sig2n["min"], self.sig2n["max"]):
                    b.append((np.log(lo), np.log(hi)))
        return b

    def _set_from_tparams(self, tparams, fix_R):
        slots = self._param_slots()
        for i, (_, setter, _, _, scale) in enumerate(slots):
            if i == 0 and fix_R:
                continue
            setter(np.exp(tparams[i]) * scale)
        p = len(slots)
        if self._sig2n_is_scalar():
            self.sig2n["value"] = np.exp(tparams[p])
        else:
            self.sig2n["value"] = np.exp(tparams[p:])

    def _log_prior(self):
        lp = 0.0
        for getter, _, prior, _, _ in self._param_slots():
            lp = lp + prior.lpdf(getter())
        if self._sig2n_is_scalar():
            lp = lp + self.sig2n["prior"].lpdf(self.sig2n["value"])
        else:
            for pr, v in zip(self.sig2n["prior"], self.sig2n["value"]):
                lp = lp + pr.lpdf(v)
        return lp

    def _safe_loglik(self):
        return self.loglik()

    def _objective(self, tparams, fix_R):
        """-(loglik + log prior) at log-parameters `tparams` (writes them into the param dicts, as the reference does)."""
        with np.errstate(all="ignore"):          # the reference runs under np.seterr(all='ignore') (gpcsd1d.py:7)
            self._set_from_tparams(tparams, fix_R)
            lp = self._log_prior()
            return -1.0 * (self._safe_loglik() + lp)

    def _objective_grad(self, tparams, fix_R, fd_step=1e-6):
        """Gradient of `_objective` w.r.t. the log-parameters (non-finite values are passed through to the optimiser
        silently, as under the reference's module-level `np.seterr(all='ignore')`, gpcsd1d.py:7)."""
        return self._objective_and_grad(tparams, fix_R, fd_step)[1]

    def _chain_rule(self, tparams, g_nat, fix_R):
        """d(-(loglik + log prior)) / d log-parameters from the natural-parameter gradient of loglik."""
        slots = self._param_slots()
        g = np.zeros_like(tparams)
        for i, (getter, _, prior, _, _) in enumerate(slots):
            v = getter()
            g[i] = -(g_nat[i] + prior.dlpdf(v)) * v          # d/dlog(v) = v d/dv
        p = len(slots)
        if self._sig2n_is_scalar():
            v = self.sig2n["value"]
            g[p] = -(g_nat[p] + self.sig2n["prior"].dlpdf(v)) * v
        else:                                                # per-electrode noise list (gpcsd1d.py:71-73)
            for k, (pr, v) in enumerate(zip(self.sig2n["prior"], self.sig2n["value"])):
                g[p + k] = -(g_nat[p + k] + pr.dlpdf(v)) * v
        if fix_R:
            g[0] = 0.0
        return g

    def _objective_and_grad(self, tparams, fix_R, fd_step=1e-6):
        """(objective, gradient) from ONE device evaluation: the front half (Gram builds + both eigendecompositions) is most
        of an evaluation, so the optimiser is driven with jac=True instead of separate fun / jac callbacks that would each
        run it.  A numerical failure raises LinAlgError, which ends the restart exactly as the reference's jac callback
        does (gpcsd1d.py:219, gpcsd2d.py:258)."""
        with np.errstate(all="ignore"):          # the reference runs under np.seterr(all='ignore') (gpcsd1d.py:7)
            tparams = np.asarray(tparams, dtype=np.float64)
            if getattr(self, "_use_analytic_grad", True) and (not self._uses_host_kt() or self._host_kt_is_differentiable()):
                if self._batch_can_evaluate() and self._vector_glue_applies():
                    # the batch of one: the same array expressions (and the same device code, gpcsd_loglik_grad is the B = 1 case
                    # of gpcsd_loglik_grad_batch), so a restart gets the same bits alone and in a lock-step batch
                    r = self._objective_and_grad_batch([(0, tparams)], fix_R)[0]
                    if isinstance(r, Exception):
                        raise r
                    return r
                self._set_from_tparams(tparams, fix_R)
                lp = self._log_prior()
                ll, g_nat = self._loglik_and_grad_natural()
                return -1.0 * (ll + lp), self._chain_rule(tparams, g_nat, fix_R)
            # finite differences of the objective (user-defined temporal covariances without compute_dKt; diagnostics)
            f = self._objective(tparams, fix_R)
            g = np.zeros_like(tparams)
            for i in range(tparams.size):
                if i == 0 and fix_R:
                    continue
                e = np.zeros_like(tparams)
                e[i] = fd_step
                g[i] = (self._objective(tparams + e, fix_R) - self._objective(tparams - e, fix_R)) / (2 * fd_step)
            self._set_from_tparams(tparams, fix_R)
            return f, g

    def _current_tparams(self):
        """The log-parameter vector of the current hyper-parameter state (inverse of `_set_from_tparams`)."""
        with np.errstate(divide="ignore"):
            tp = [np.log(getter() / scale) for getter, _, _, _, scale in self._param_slots()]
            tp.extend(np.log(np.atleast_1d(np.asarray(self.sig2n["value"], dtype=np.float64))))
        return np.array(tp, dtype=np.float64)

    def _sample_start(self, fix_R):
        slots = self._param_slots()
        tp = []
        for i, (getter, _, prior, _, scale) in enumerate(slots):
            if i == 0 and fix_R:
                tp.append(np.log(getter()) - np.log(scale))
            else:
                tp.append(np.log(prior.sample()) - np.log(scale))
        if self._sig2n_is_scalar():
            tp.append(np.log(self.sig2n["prior"].sample()))
        else:
            for pr in self.sig2n["prior"]:
                tp.append(np.log(pr.sample()))
        return np.array(tp)

    # ------------------------------------------------------------------ restarts: concurrency and sharding
    def shard_restarts(self, sharding):
        """Split the restarts of `fit` across ranks (restart k runs on rank k % world_size, every rank holds all trials);
        results are combined with two tiny all-reduces and every rank ends with the same best parameters
        (BASELINE cfg5).  `sharding`: gpcsd_amd.dist.TrialSharding (only rank / world_size / allreduce)."""
        self._restart_sharding = sharding

    def _clone_for_worker(self):
        """Independent hyper-parameter state on top of the SAME data arrays, with its own device context (= own HIP stream):
        restarts are latency-bound chains of small kernels, so several of them interleave well on one GPU."""
        import copy
        m = copy.copy(self)
        m.R = copy.deepcopy(self.R)
        m.sig2n = copy.deepcopy(self.sig2n)
        m.spatial_cov = copy.copy(self.spatial_cov)
        m.spatial_cov.params = copy.deepcopy(self.spatial_cov.params)
        m.temporal_cov_list = []
        for tc in self.temporal_cov_list:
            t2 = copy.copy(tc)
            t2.params = copy.deepcopy(tc.params)
            m.temporal_cov_list.append(t2)
        m._ctx = None
        m._resident = {}
        return m

    def _batch_can_evaluate(self):
        # (a subclass that brings its own objective is evaluated through it, one point at a time)
        cls = type(self)
        own = (cls._loglik_and_grad_natural_batch is not GPCSDModel._loglik_and_grad_natural_batch or
               cls._eval_batch_local is not GPCSDModel._eval_batch_local or
               (cls._objective_and_grad is GPCSDModel._objective_and_grad and
                cls._loglik_and_grad_natural is GPCSDModel._loglik_and_grad_natural))
        # (scalar noise or a per-electrode list: gpcsd_loglik_grad_batch takes either since round 5)
        return (own and getattr(self, "_use_analytic_grad", True) and not self._uses_host_kt())

    def _vector_glue_applies(self):
        """The hyper-parameters of a batch can be unpacked, their priors evaluated and the chain rule applied as array
        operations: the library's own temporal kernels, priors that offer lpdf_many / dlpdf_many (scalar noise or a
        per-electrode list of priors)."""
        if self._uses_host_kt():
            return False
        priors = [sl[2] for sl in self._param_slots()] + self._noise_priors()
        return all(callable(getattr(pr, "lpdf_many", None)) and callable(getattr(pr, "dlpdf_many", None)) for pr in priors)

    def _noise_priors(self):
        """the noise prior(s) as a list: one entry for scalar noise, nx for a per-electrode list (gpcsd1d.py:58-60)"""
        return [self.sig2n["prior"]] if self._sig2n_is_scalar() else list(self.sig2n["prior"])

    def _objective_and_grad_batch(self, items, fix_R):
        """{key: (objective, gradient) or exception} for [(key, tparams)]: the lock-step evaluation behind fit(batch=k).
        Runs single-threaded (every optimiser chain is waiting for its result), so walking the shared param dicts is safe.
        With the library's own priors and kernels the host side of a batch is array arithmetic (B x p): per tick of a 32-restart
        fit the Python loop below it cost about as much as a quarter of the device evaluation."""
        if not self._vector_glue_applies():
            return self._objective_and_grad_batch_loop(items, fix_R)
        with np.errstate(all="ignore"):
            slots = self._param_slots()
            p = len(slots)
            noise_priors = self._noise_priors()
            nsig = len(noise_priors)
            tps = np.stack([np.asarray(tp, dtype=np.float64) for _, tp in items])                 # (B, p + nsig)
            scales = np.array([sl[4] for sl in slots] + [1.0] * nsig)
            nat = np.exp(tps) * scales                                                            # natural values, slot order
            if fix_R:
                nat[:, 0] = slots[0][0]()
            priors = [sl[2] for sl in slots] + noise_priors
            lp = np.zeros(len(items))
            for i, pr in enumerate(priors):                                                       # same order as _log_prior
                lp = lp + pr.lpdf_many(nat[:, i])
            ns = len(self._spatial_names)
            C = len(self.temporal_cov_list)
            kinds = [k for k, _, _ in self._temporal_triplets()]
            tcols = nat[:, 1 + ns:1 + ns + 2 * C]
            hps = _hip.HParamsBatch(nat[:, 0], getattr(self, "eps", 0.0), nat[:, 1:1 + ns], kinds, tcols[:, 0::2], tcols[:, 1::2],
                                    nat[:, p] if self._sig2n_is_scalar() else nat[:, p:], self.JITTER)
            if getattr(self, "_resident", {}).get("host_kt"):
                self._context().set_host_temporal_gram(None)
                self._resident["host_kt"] = False
            ll, g_nat, failed = self._eval_batch_reduced(hps)
            dlp = np.stack([pr.dlpdf_many(nat[:, i]) for i, pr in enumerate(priors)], axis=1)
            G = -(g_nat + dlp) * nat                                                              # d/dlog(v) = v d/dv
            if fix_R:
                G[:, 0] = 0.0
            F = -1.0 * (ll + lp)
            out = {}
            for b, (key, _) in enumerate(items):
                out[key] = self._batch_failure(b) if failed[b] else (F[b], G[b])
            self._set_from_tparams(tps[-1], fix_R)       # the param dicts hold the last point evaluated, as after a scalar call
            return out

    def _objective_and_grad_batch_loop(self, items, fix_R):
        """The same, one hyper-parameter set at a time through the param dicts (user-defined priors without array methods)."""
        with np.errstate(all="ignore"):
            hps, keep, lps, tps = [], [], [], []
            for _, tp in items:
                tp = np.asarray(tp, dtype=np.float64)
                self._set_from_tparams(tp, fix_R)
                lps.append(self._log_prior())
                hp, k = self._hparams(self.JITTER)
                hps.append(hp)
                keep.append(k)
                tps.append(tp)
            res = self._loglik_and_grad_natural_batch(hps)
            out = {}
            for (key, _), tp, lp, r in zip(items, tps, lps, res):
                if isinstance(r, Exception):
                    out[key] = r
                    continue
                ll, g_nat = r
                self._set_from_tparams(tp, fix_R)            # the chain rule reads the values back from the dicts
                out[key] = (-1.0 * (ll + lp), self._chain_rule(tp, g_nat, fix_R))
            return out

    def _run_restart(self, tparams0, method, fix_R, options, bounds, evaluate=None):
        """One L-BFGS-B chain.  evaluate: callback x -> (objective, gradient) of a lock-step batch; default: this model."""
        fun = evaluate if evaluate is not None else (lambda tp: self._objective_and_grad(tp, fix_R))
        try:
            res = scipy.optimize.minimize(fun, tparams0, method=method, options=options, bounds=bounds, jac=True)
            return res.fun, res.x, res.message
        except (ValueError, np.linalg.LinAlgError) as e:
            print(e)
            if self.dim == 2:
                print("\nrestarting optimization...")
            return None

    # device memory a lock-step batch may take for its per-set arrays (five of nx * ntrials * nt doubles per set)
    FIT_BATCH_BYTES = 32 << 30
    # "auto" | "setulb" | "threads": who steps the restarts' optimisers between batched evaluations (see _fit)
    fit_driver = "auto"

    def _auto_batch(self, n):
        """Default lock-step width of fit(): all restarts of this rank, capped at 32 and by FIT_BATCH_BYTES."""
        lfp = self._local_lfp()
        per_set = 5 * 8 * lfp.shape[0] * lfp.shape[1] * lfp.shape[2]
        return int(max(1, min(n, 32, self.FIT_BATCH_BYTES // max(per_set, 1))))

    def _fit(self, n_restarts, method, fix_R, verbose, options, starts=None, workers=1, batch=None):
        bounds = self._bounds()
        # starting points are drawn up front, in the order the sequential loop of the reference consumes the RNG
        # (the optimiser itself draws nothing), so results do not depend on `workers` or on the number of ranks
        if starts is None:
            starts = [self._sample_start(fix_R) for _ in range(n_restarts)]
        starts = [np.asarray(s0, dtype=np.float64) for s0 in starts]
        rs = getattr(self, "_restart_sharding", None)
        # Ranks do not share a random stream: constructors and `_sample_start` draw from each process's own NumPy RNG.
        # Under trial sharding every rank must walk the SAME optimiser trajectory (the all-reduces inside the objective pair
        # up call by call), under restart sharding restart k must mean the same start everywhere: rank 0's current
        # hyper-parameters (fix_R reads R from them) and rank 0's starts are broadcast before anything is evaluated.
        sync = getattr(self, "_sharding", None) or rs
        if sync is not None and sync.world_size > 1:
            self._set_from_tparams(sync.broadcast(self._current_tparams(), src=0), False)
            if len(starts):
                flat = sync.broadcast(np.stack(starts).ravel(), src=0)
                starts = [row.copy() for row in flat.reshape(len(starts), -1)]
        self.fit_starts_ = [s0.copy() for s0 in starts]
        mine = [k for k in range(n_restarts) if rs is None or k % rs.world_size == rs.rank]
        results = {}
        workers = max(1, min(int(workers), len(mine)))
        # Replicas in one chain of launches are nearly free on the GPU and each restart gets the bits of a run of its own, so
        # by default all of this rank's restarts advance in lock-step
        if batch is None:
            batch = self._auto_batch(len(mine)) if (len(mine) > 1 and self._batch_can_evaluate()) else 1
        batch = max(1, min(int(batch), max(len(mine), 1)))
        if batch > 1 and self._batch_can_evaluate():
            # lock-step restarts: `batch` SciPy chains alive at a time, their evaluations served by ONE batched device call.
            # workers > 1: that many such groups side by side, each on its own context (= own streams and buffers).  A batched
            # evaluation is a latency-bound eigen phase (a few CUs) followed by a throughput-bound GEMM phase (all CUs); two
            # groups drift out of phase and one's eigen chains run under the other's GEMMs.  Not under trial sharding: the
            # groups' all-reduces would interleave differently on different ranks.
            from .lockstep import run_chains
            from . import lbfgsb_lockstep
            ngroups = 1 if getattr(self, "_sharding", None) is not None else max(1, min(workers, len(mine) // 2))
            parts = [mine[gi::ngroups] for gi in range(ngroups)]
            # Who steps the optimisers between the batched evaluations.  "setulb" (the default where this SciPy has it): ONE
            # driver steps every chain's L-BFGS-B state through SciPy's reverse-communication entry point -- the same compiled
            # optimiser scipy.optimize.minimize runs, the same trajectories bit for bit, without a thread, two condition-variable
            # hand-offs and a GIL switch per chain and evaluation.  "threads": unmodified minimize() calls on threads that
            # rendezvous in their objective callbacks (round 2; any `method`).
            driver = getattr(self, "fit_driver", "auto")
            use_setulb = (method == "L-BFGS-B" and driver in ("auto", "setulb") and lbfgsb_lockstep.available())
            if driver == "setulb" and not use_setulb:
                raise RuntimeError("fit_driver='setulb' needs method='L-BFGS-B' and scipy.optimize._lbfgsb.setulb")
            self.fit_driver_used_ = "setulb" if use_setulb else "threads"
            if driver == "auto" and not use_setulb:
                # never a silent fallback: the threaded rendezvous gives the same optima at about two thirds of the rate
                import warnings
                why = ("method=%r is not L-BFGS-B" % (method,) if method != "L-BFGS-B" else
                       "this SciPy (%s) does not reproduce minimize(method='L-BFGS-B') through scipy.optimize._lbfgsb.setulb "
                       "(lbfgsb_lockstep.available() is False)" % scipy.__version__)
                warnings.warn("gpcsd_amd.fit: lock-step restarts are stepped by the thread rendezvous around unmodified "
                              "scipy.optimize.minimize calls because %s; same optima, but 0.45-0.69 of the evaluation rate of "
                              "the single L-BFGS-B driver (measured: 5.2 k against 7.5 k evaluations/s on the 24 x 500 x 200 "
                              "fit).  Set model.fit_driver = 'threads' to choose this path without the warning." % why,
                              RuntimeWarning, stacklevel=3)

            class _Stats:
                batches = points = 0

            def run_group(model, ks):
                if not use_setulb:
                    return run_chains([starts[k] for k in ks],
                                      lambda s0, evaluate: model._run_restart(s0, method, fix_R, options, bounds, evaluate),
                                      lambda items: model._objective_and_grad_batch(items, fix_R), min(batch, len(ks)))
                res, st = lbfgsb_lockstep.minimize_many(lambda items: model._objective_and_grad_batch(items, fix_R),
                                                        [starts[k] for k in ks], bounds, options, min(batch, len(ks)))
                out = {}
                for i in range(len(ks)):
                    r = res[i]
                    if isinstance(r, Exception):          # what _run_restart does with a failed restart (gpcsd1d.py:219, gpcsd2d.py:258-260)
                        print(r)
                        if model.dim == 2:
                            print("\nrestarting optimization...")
                        r = None
                    out[i] = r
                ev = _Stats()
                ev.batches, ev.points = st["batches"], st["points"]
                return out, ev
            if ngroups == 1:
                outs = [run_group(self, parts[0])]
            else:
                from concurrent.futures import ThreadPoolExecutor
                models = [self] + [self._clone_for_worker() for _ in range(ngroups - 1)]
                with ThreadPoolExecutor(max_workers=ngroups) as ex:
                    outs = list(ex.map(lambda a: run_group(*a), zip(models, parts)))
            nb = npts = 0
            for ks, (out, ev) in zip(parts, outs):
                for i, k in enumerate(ks):
                    r = out[i]
                    if isinstance(r, Exception):
                        raise r
                    results[k] = r
                nb += ev.batches
                npts += ev.points
            self.fit_batches_ = (nb, npts)
        elif workers == 1:
            for k in tqdm(mine, desc="Restarts"):
                results[k] = self._run_restart(starts[k], method, fix_R, options, bounds)
        else:
            from concurrent.futures import ThreadPoolExecutor
            import queue
            pool = queue.Queue()
            for _ in range(workers):
                pool.put(self._clone_for_worker())

            def job(k):
                m = pool.get()
                try:
                    return k, m._run_restart(starts[k], method, fix_R, options, bounds)
                finally:
                    pool.put(m)
            with ThreadPoolExecutor(max_workers=workers) as ex:
                for k, r in ex.map(job, mine):
                    results[k] = r
        p = len(starts[0]) if starts else 0
        status = np.zeros(n_restarts)            # 0 failed, 1 finite optimum, 2 non-finite objective
        nll_all = np.zeros(n_restarts)
        par_all = np.zeros((n_restarts, p))
        msgs = {}
        for k, r in results.items():
            if r is None:
                continue
            fun, x, msg = r
            status[k] = 1 if np.isfinite(fun) else 2
            nll_all[k] = fun if np.isfinite(fun) else 0.0
            par_all[k] = x
            msgs[k] = msg
        if rs is not None:
            status = rs.allreduce_sum(status)
            nll_all = rs.allreduce_sum(nll_all)
            par_all = rs.allreduce_sum(par_all.ravel()).reshape(n_restarts, p)
        done = [k for k in range(n_restarts) if status[k] > 0]
        nll_values = np.array([nll_all[k] if status[k] == 1 else np.inf for k in done])
        if len(nll_values) < 1:
            print("problem with optimization!")
            return None
        finite = np.isfinite(nll_values)
        params = [par_all[k] for k, ok in zip(done, finite) if ok]
        best_ind = np.argmin(nll_values[finite])
        if verbose:
            print("\nNeg log lik values across different initializations:")
            print(nll_values)
            print("Best index termination message")
            print(msgs.get([k for k, ok in zip(done, finite) if ok][best_ind], "(other rank)"))
        self._set_from_tparams(params[best_ind], fix_R)
        self.fit_nll_values_ = nll_values
        self.fit_params_ = params
        return None

    # ------------------------------------------------------------------ prediction
    _PRED_BUFFERS = (("csd", _hip.PRED_CSD, "pred_out_csd"), ("lfp", _hip.PRED_LFP, "pred_out_lfp"))

    def _device_predictions(self, ctx, code, nz, nt, R_local):
        """Zero-copy views (`_hip.DeviceArray`) of the posterior means a resident prediction left in HBM."""
        C = len(self.temporal_cov_list)
        res = {}
        for name, bit, buf in self._PRED_BUFFERS:
            if code & bit:
                res[name] = ctx.device_array(buf, (nz, nt, R_local))
                res[name + "_list"] = ctx.device_array(buf + "_list", (C, nz, nt, R_local))
        return res

    def _store_predictions(self, res, z, t):
        if res.get("csd") is not None:
            self.csd_pred_list = [res["csd_list"][i] for i in range(res["csd_list"].shape[0])]
            self.csd_pred = res["csd"]
        if res.get("lfp") is not None:
            self.lfp_pred_list = [res["lfp_list"][i] for i in range(res["lfp_list"].shape[0])]
            self.lfp_pred = res["lfp"]
        self.t_pred = t
        self.x_pred = z

    def predict(self, z, t, type="csd", resident=False):
        """Posterior mean of CSD and/or LFP at sites z and times t for every (local) trial (gpcsd1d.py:248-293 / gpcsd2d.py:289-334).

        resident=True (no reference counterpart) leaves the results in HBM: `csd_pred` / `lfp_pred` are then zero-copy device views
        (`__cuda_array_interface__`, float64, (nz, nt, ntrials); the `*_list` attributes (C, nz, nt, ntrials)) instead of NumPy
        arrays -- `torch.as_tensor(m.csd_pred, device="cuda")` -- valid until the model's next prediction.  Returning host arrays
        is bound by the PCIe link (231 MB per call at 384 x 500 x 50), the resident form by the GPU."""
        if type not in ("csd", "lfp", "both"):
            raise ValueError("type must be 'csd', 'lfp' or 'both'")
        ctx = self._sync_device()
        z = np.asarray(z, dtype=np.float64)
        z2 = z.reshape(-1, 1) if self.dim == 1 else z
        t = np.asarray(t)
        hp, _keep = self._hparams(0.0, tstar=t)            # no jitter in predict
        code = {"csd": _hip.PRED_CSD, "lfp": _hip.PRED_LFP, "both": _hip.PRED_BOTH}[type]
        R_local = self._local_lfp().shape[2]
        sh = getattr(self, "_sharding", None)
        gather = sh is not None and getattr(sh, "gather_predictions", False)
        if resident or (gather and sh.on_device()):
            ctx.predict_resident(hp, z2, t, code, want_lists=True)
            res = self._device_predictions(ctx, code, z2.shape[0], t.shape[0], R_local)
            if gather:
                # trial blocks gathered ON THE DEVICE (RCCL all-gather over xGMI), one copy out on the gathering rank(s) only
                res = {k: sh.gather_trials_device(v, dst=getattr(sh, "gather_dst", None)) for k, v in res.items()}
        else:
            res = ctx.predict(hp, z2, t, code, (z2.shape[0], t.shape[0], R_local))
            if gather:
                res = {k: sh.gather_trials(v) for k, v in res.items()}
        self._store_predictions(res, z, t)

    def loglik_predict_many(self, param_sets, z, t, type="csd", resident=False, share_spatial=False):
        """loglik() and predict(z, t, type) under each of a LIST of hyper-parameter sets -- dicts as extract_model_params() returns
        them: the optima of every restart of a fit, a grid, posterior draws -- in order.  Returns the log-likelihoods, one per set;
        the prediction attributes hold the LAST set's posterior means as after predict() (resident=True: left in HBM, see
        predict), and the model's hyper-parameters are left at the last set.

        No reference counterpart as ONE call: the reference's callers loop restore_model_params -> loglik -> predict
        (neuropixels/fit_gpcsd2d.py:93-107).  Because the sets are known up front, every set's evaluation is queued as one paired
        call (its four eigenproblems share launches two by two) that ANNOUNCES the next set (gpcsd_prefetch_pair): the next set's
        decomposition chains run under this set's products instead of behind the host's collection of its value.
        share_spatial=True additionally decomposes ONE spatial matrix per set where both calls have the same spatial
        hyper-parameters (Ks + jitter I and Ks share eigenvectors; results then differ from the fenced calls in the last bits)."""
        if type not in ("csd", "lfp", "both"):
            raise ValueError("type must be 'csd', 'lfp' or 'both'")
        param_sets = list(param_sets)
        if not param_sets:
            return np.zeros(0)
        ctx = self._sync_device()
        z = np.asarray(z, dtype=np.float64)
        z2 = np.ascontiguousarray(z.reshape(-1, 1) if self.dim == 1 else z)
        t = np.asarray(t)
        code = {"csd": _hip.PRED_CSD, "lfp": _hip.PRED_LFP, "both": _hip.PRED_BOTH}[type]
        sets = []
        for p in param_sets:                               # every set's two hyper-parameter structs, before anything is queued
            self.restore_model_params(p)
            sets.append((self._hparams(self.JITTER), self._hparams(0.0, tstar=t)))
        R_local = self._local_lfp().shape[2]
        ntrials = np.shape(self.lfp)[2]
        sh = getattr(self, "_sharding", None)
        was = ctx.pair_share_s_on()
        ctx.pair_share_s(bool(share_spatial))
        out = np.empty(len(sets))
        try:
            for k, ((hp, _k1), (hp0, _k0)) in enumerate(sets):
                ctx.loglik_predict_async(hp, hp0, z2, t, code, want_lists=True)
                if k + 1 < len(sets):
                    (hn, _a), (hn0, _b) = sets[k + 1]
                    ctx.prefetch_pair(hn, hn0, z2, t)
                sumlog, quad = ctx.loglik_parts_wait()
                if sh is not None:
                    quad = float(sh.allreduce_sum(np.array([quad]))[0])
                out[k] = -0.5 * ntrials * sumlog - 0.5 * quad
        finally:
            ctx.pair_share_s(was)
        res = self._device_predictions(ctx, code, z2.shape[0], t.shape[0], R_local)
        gather = sh is not None and getattr(sh, "gather_predictions", False)
        if gather and sh.on_device():
            res = {k: sh.gather_trials_device(v, dst=getattr(sh, "gather_dst", None)) for k, v in res.items()}
        elif not resident or gather:
            res = {k: ctx.fetch(v.name, v.shape) for k, v in res.items()}
            if gather:
                res = {k: sh.gather_trials(v) for k, v in res.items()}
        self._store_predictions(res, z, t)
        return out

    def _sample_prior_from_normals(self, normals, which):
        """Ls Z_r Lt^T on the GPU for host-supplied standard normals (nx, nt, ntrials)."""
        ctx = self._sync_device(need_lfp=False)
        hp, _keep = self._hparams(self.JITTER)
        return ctx.sample_prior(hp, which, normals)
