"""Multi-process (world_size 2, gloo, CPU) test of the trial-sharding layer used by the N>1 bench path.
The local evaluator is the CPU oracle here (tests may use it as the checker); on GPUs it is the HIP context."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from gpcsd_amd.dist import TrialSharding, sharded_loglik
    from helpers import load_model_case, with_jitter
    from oracle import gpcsd_oracle as O
    c, g, geom, hp, lfp = load_model_case("1d_odd_17x37x5")
    hp = with_jitter(hp, 1e-8)
    sh = TrialSharding(gather_predictions=True)
    sl = sh.local_slice(lfp.shape[2])
    local = lfp[:, :, sl]

    def parts():
        Ks = O.spatial_kphi(geom, hp) + hp["jitter"] * np.eye(17)
        Kt = O.temporal_sum(hp["temporal"], geom.t)
        Qs, Qt, D = O.eig_D(Ks, Kt, hp["sig2n"])
        Y = np.moveaxis(local, 2, 0)
        alpha = np.matmul(np.matmul(Qs.T, Y), Qt)
        return float(np.sum(np.log(D))), float(np.sum(alpha ** 2 / D.reshape(1, 17, 37)))

    ll = sharded_loglik(parts, lfp.shape[2], sh)
    hp0 = with_jitter(hp, 0.0)
    pred_local = O.predict(geom, hp0, local, c["x"], c["t"], type="csd")["csd"]
    pred_full = sh.gather_trials(pred_local)
    vec = sh.broadcast(np.arange(5.0) * (1 + rank), src=0)
    pend = sh.allreduce_sum_async(np.array([1.0 + rank, 2.0]))
    assert np.allclose(pend(), [3.0, 4.0])
    if rank == 0:
        ref = O.loglik(geom, hp, lfp)
        pref = O.predict(geom, hp0, lfp, c["x"], c["t"], type="csd")["csd"]
        q.put((ll, ref, float(np.max(np.abs(pred_full - pref))), vec.tolist(), (sl.start, sl.stop)))
    else:
        q.put((ll, None, None, vec.tolist(), (sl.start, sl.stop)))
    td.destroy_process_group()


def test_block_partition():
    from gpcsd_amd.dist import TrialSharding
    for n, w in [(400, 8), (5, 2), (3, 4), (50, 1)]:
        blocks = [TrialSharding.block(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in blocks]
        assert max(sizes) - min(sizes) <= 1
    assert TrialSharding.block(400, 3, 8) == (150, 200)


@pytest.mark.timeout(300)
def test_sharded_loglik_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    lls = [r[0] for r in res]
    ref = [r[1] for r in res if r[1] is not None][0]
    assert abs(lls[0] - lls[1]) < 1e-9 * abs(ref)
    assert abs(lls[0] - ref) / abs(ref) < 1e-12
    perr = [r[2] for r in res if r[2] is not None][0]
    assert perr < 1e-12
    assert all(r[3] == [0.0, 1.0, 2.0, 3.0, 4.0] for r in res)
    assert sorted(r[4] for r in res) == [(0, 3), (3, 5)]


# ---------------------------------------------------------------------------------------------------------------------
# fit(): restarts run concurrently (workers) and sharded over ranks -- host logic only, the objective is a stub
# ---------------------------------------------------------------------------------------------------------------------
def _stub_model():
    """GPCSDModel whose objective is an analytic multi-well function of two log-parameters (no device needed)."""
    from gpcsd_amd.model_base import GPCSDModel

    class _P:
        params = {}

    class Stub(GPCSDModel):
        dim = 1

        def __init__(self):
            self.R = {"value": 1.0}
            self.sig2n = {"value": 0.1}
            self.spatial_cov = _P()
            self.temporal_cov_list = []
            self.best = None

        def _bounds(self):
            return [(-3.0, 3.0), (-3.0, 3.0)]

        def _objective(self, tp, fix_R):
            x, y = tp
            return float((x * x - 1.0) ** 2 + 0.3 * x + (y - 0.5) ** 2)

        def _objective_and_grad(self, tp, fix_R, fd_step=1e-6):
            x, y = tp
            return self._objective(tp, fix_R), np.array([4.0 * x * (x * x - 1.0) + 0.3, 2.0 * (y - 0.5)])

        def _current_tparams(self):
            return np.zeros(2)

        def _set_from_tparams(self, tp, fix_R):
            self.best = np.array(tp, dtype=np.float64)

    return Stub()


def _restart_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from gpcsd_amd.dist import TrialSharding
    m = _stub_model()
    m.shard_restarts(TrialSharding())
    starts = [np.array([s, -s]) for s in np.linspace(-2.0, 2.0, 7)]
    m._fit(7, "L-BFGS-B", False, False, {"maxiter": 200}, starts=starts, workers=2)
    q.put((rank, m.best.tolist(), np.asarray(m.fit_nll_values_).tolist()))
    td.destroy_process_group()


def test_fit_restarts_workers_match_sequential():
    starts = [np.array([s, -s]) for s in np.linspace(-2.0, 2.0, 7)]
    a, b = _stub_model(), _stub_model()
    a._fit(7, "L-BFGS-B", False, False, {"maxiter": 200}, starts=starts, workers=1)
    b._fit(7, "L-BFGS-B", False, False, {"maxiter": 200}, starts=starts, workers=3)
    assert np.allclose(a.fit_nll_values_, b.fit_nll_values_, rtol=0, atol=0)
    assert np.array_equal(a.best, b.best)
    assert abs(a.best[0] + 1.03) < 0.02 and abs(a.best[1] - 0.5) < 1e-5       # the deeper of the two wells


@pytest.mark.timeout(300)
def test_fit_restart_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_restart_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    seq = _stub_model()
    seq._fit(7, "L-BFGS-B", False, False, {"maxiter": 200}, starts=[np.array([s, -s]) for s in np.linspace(-2.0, 2.0, 7)])
    for rank, best, nll in res:
        assert np.allclose(best, seq.best, rtol=0, atol=1e-12)
        assert np.allclose(nll, seq.fit_nll_values_, rtol=0, atol=1e-12)


# ---------------------------------------------------------------------------------------------------------------------
# fit(starts=None) under trial sharding with UNSEEDED ranks: constructors and restarts draw from each process's own RNG,
# so rank 0's hyper-parameters and starts must be broadcast or the ranks optimise different models and their all-reduces
# stop pairing up.  CPU evaluator: the oracle stands in for the HIP context (checker code in a test).
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_backed_model(seed):
    from gpcsd_amd.gpcsd1d import GPCSD1D
    from gpcsd_amd.covariances import GPCSDTemporalCovSE
    from oracle import gpcsd_oracle as O
    x = np.linspace(0, 1100, 12)[:, None]
    t = np.linspace(0, 29, 30)[:, None]
    lfp = np.random.RandomState(5).standard_normal((12, 30, 4))
    geom = O.Geometry1D(x, t, a=0.0, b=1100.0, ngl=30)

    class OracleBacked(GPCSD1D):
        """GPCSD1D whose device evaluation is replaced by the oracle on the local block of trials."""
        evals = 0

        def _natural_hp(self, vec):
            return O.make_hparams(vec[0], (vec[1],), [(O.SE, vec[2], vec[3])], vec[4], jitter=1e-8)

        def _loglik_and_grad_natural(self):
            OracleBacked.evals += 1
            tc = self.temporal_cov_list[0]
            v0 = np.array([self.R["value"], self.spatial_cov.params["ell"]["value"], tc.params["ell"]["value"],
                           tc.params["sigma2"]["value"], self.sig2n["value"]], dtype=np.float64)
            local = self._local_lfp()
            ll = O.loglik(geom, self._natural_hp(v0), local)
            g = np.zeros(5)
            for i in range(5):
                h = 1e-6 * abs(v0[i])
                vp, vm = v0.copy(), v0.copy()
                vp[i] += h
                vm[i] -= h
                g[i] = (O.loglik(geom, self._natural_hp(vp), local) - O.loglik(geom, self._natural_hp(vm), local)) / (2 * h)
            sh = getattr(self, "_sharding", None)
            if sh is not None:
                red = sh.allreduce_sum(np.concatenate([[ll], g]))
                ll, g = float(red[0]), red[1:]
            return ll, g

    np.random.seed(seed)                     # what the constructors and _sample_start draw from
    m = OracleBacked(lfp, x, t, a=0.0, b=1100.0, ngl=30, temporal_cov_list=[GPCSDTemporalCovSE(t)])
    return m, OracleBacked


def _unseeded_fit_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from gpcsd_amd.dist import TrialSharding
    m, cls = _oracle_backed_model(seed=100 + 17 * rank)          # every rank has its own random stream
    m.shard_trials(TrialSharding())
    m.fit(n_restarts=2, options={"maxiter": 4, "disp": False, "gtol": 1e-5, "ftol": 1e-9})
    q.put((rank, np.asarray(m.fit_nll_values_).tolist(), [p.tolist() for p in m.fit_params_], cls.evals,
           float(m.R["value"]), [s0.tolist() for s0 in m.fit_starts_]))
    td.destroy_process_group()


@pytest.mark.timeout(600)
def test_fit_with_unseeded_ranks_is_synchronised_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_unseeded_fit_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=500) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, nll0, par0, ev0, R0, st0), (_, nll1, par1, ev1, R1, st1) = res
    assert ev0 == ev1                                    # same number of objective evaluations: collectives paired up
    assert np.array_equal(nll0, nll1) and np.array_equal(par0, par1) and R0 == R1 and st0 == st1
    # and it is rank 0's starts that were optimised: the single-process run with rank 0's seed draws the same ones
    # (the optima themselves are not compared: four iterations on a finite-difference gradient summed over two shards
    # take different line-search decisions than the same gradient evaluated in one piece)
    m, _ = _oracle_backed_model(seed=100)
    m.fit(n_restarts=2, options={"maxiter": 1, "disp": False})
    assert np.allclose(np.array(m.fit_starts_), np.array(st0), rtol=1e-12, atol=0)


# ---------------------------------------------------------------------------------------------------------------------
# fit(batch=k) under trial sharding goes through the REAL lock-step batch path: the whole batch's partial sums travel in one
# element-wise all-reduce, so slot b must be the same restart on every rank (the batch used to come out in
# thread-arrival order, which differs between ranks and between runs).
# ---------------------------------------------------------------------------------------------------------------------
def _batched_oracle_model(seed):
    m, cls = _oracle_backed_model(seed)
    from oracle import gpcsd_oracle as O
    geom = O.Geometry1D(m.x, m.t, a=0.0, b=1100.0, ngl=30)

    class BatchedOracle(cls):
        """The per-rank device evaluation of a batch is the oracle on the local block of trials; hyper-parameter structs
        are natural-parameter vectors (no device context on the CPU)."""
        batch_orders = []

        def _hparams(self, jitter, tstar=None):
            tc = self.temporal_cov_list[0]
            return np.array([self.R["value"], self.spatial_cov.params["ell"]["value"], tc.params["ell"]["value"],
                             tc.params["sigma2"]["value"], self.sig2n["value"]], dtype=np.float64), None

        def _local_ll_grad(self, v0):
            local = self._local_lfp()
            ll = O.loglik(geom, self._natural_hp(v0), local)
            g = np.zeros(5)
            for i in range(5):
                h = 1e-6 * abs(v0[i])
                vp, vm = v0.copy(), v0.copy()
                vp[i] += h
                vm[i] -= h
                g[i] = (O.loglik(geom, self._natural_hp(vp), local) - O.loglik(geom, self._natural_hp(vm), local)) / (2 * h)
            return ll, g

        def _eval_batch_local(self, hps):
            if hasattr(hps, "rec"):                  # the array form of a batch (gpcsd_amd._hip.HParamsBatch): one record per set
                rec = hps.rec
                hps = [np.array([rec["R"][b], rec["ell_s"][b, 0], rec["ell_t"][b, 0], rec["sigma2_t"][b, 0], hps.sig[b]])
                       for b in range(len(hps))]
            res = [self._local_ll_grad(v) for v in hps]
            return np.array([r[0] for r in res]), np.stack([r[1] for r in res]), np.zeros(len(hps))

        def _loglik_and_grad_natural(self):          # batch=1 path: the same numbers, one point at a time
            ll, g = self._local_ll_grad(self._hparams(0.0)[0])
            sh = getattr(self, "_sharding", None)
            if sh is not None:
                red = sh.allreduce_sum(np.concatenate([[ll], g]))
                ll, g = float(red[0]), red[1:]
            return ll, g

        def _objective_and_grad_batch(self, items, fix_R):
            BatchedOracle.batch_orders.append([k for k, _ in items])
            return super()._objective_and_grad_batch(items, fix_R)

    m.__class__ = BatchedOracle
    return m, BatchedOracle


def _batched_fit_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from gpcsd_amd.dist import TrialSharding
    out = []
    for batch in (None, 1):
        m, cls = _batched_oracle_model(seed=7)
        assert m._batch_can_evaluate()
        m.shard_trials(TrialSharding())
        np.random.seed(11)
        m.fit(n_restarts=5, batch=batch, options={"maxiter": 5, "disp": False, "gtol": 1e-5, "ftol": 1e-9})
        out.append((np.asarray(m.fit_nll_values_).tolist(), [p.tolist() for p in m.fit_params_],
                    getattr(m, "fit_batches_", None), list(cls.batch_orders)))
    q.put((rank, out))
    td.destroy_process_group()


@pytest.mark.timeout(300)
def test_fit_lockstep_batch_under_trial_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batched_fit_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted(q.get(timeout=150) for _ in procs)
    finally:
        for p in procs:                          # a rank that died leaves its peer blocked in a collective: never wait for it
            p.join(timeout=20)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    (_, (b0, s0)), (_, (b1, s1)) = res
    # the lock-step run really batched, in sorted slot order, identically on both ranks
    assert b0[2] is not None and b0[2][1] > b0[2][0]
    assert all(o == sorted(o) for o in b0[3]) and b0[3] == b1[3]
    # both ranks end with identical optima ...
    assert b0[0] == b1[0] and b0[1] == b1[1]
    # ... and they are the optima of the one-after-the-other loop, bit for bit (the batch adds the same numbers slot by slot)
    assert b0[0] == s0[0] and b0[1] == s0[1] and s0[0] == s1[0]


# ---------------------------------------------------------------------------------------------------------------------
# the same sharding layer over the HIP path: two ranks share the one GPU of the test box (gloo for the tiny collectives)
# ---------------------------------------------------------------------------------------------------------------------
def _gpu_shard_worker(rank, world, port, q, case="1d_odd_17x37x5"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GPCSD_DEVICE"] = "0"
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    from gpcsd_amd.dist import TrialSharding
    import test_hip_parity as T
    m, c, g, geom, hp, lfp = T._build_model(case)
    m.shard_trials(TrialSharding(gather_predictions=True))
    ll = float(m.loglik())
    ll2, grad = m._loglik_and_grad_natural()
    m.predict(c["x"], c["t"], type="csd")
    perr = float(np.max(np.abs(m.csd_pred - g["csd_pred"])) / np.max(np.abs(g["csd_pred"])))
    q.put((rank, ll, float(ll2), np.asarray(grad).tolist(), tuple(m.csd_pred.shape), perr, float(g["loglik"])))
    td.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("case", ["1d_odd_17x37x5", "2d_npx_96x120x3"])      # unfolded path / folded-basis path, uneven shards
def test_hip_path_trial_sharding_world2_one_gpu(case):
    """model.shard_trials over the HIP path: each rank evaluates its block of trials on the GPU, the partial sums are
    all-reduced, predictions gathered; every rank reproduces the reference's loglik / posterior mean of the full data."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_shard_worker, args=(r, 2, port, q, case)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for rank, ll, ll2, grad, shape, perr, ll_ref in res:
        assert abs(ll - ll_ref) / abs(ll_ref) < 1e-6
        assert abs(ll2 - ll_ref) / abs(ll_ref) < 1e-6
        assert shape[2] == (5 if case.startswith("1d_odd") else 3) and perr < 1e-6
    assert np.allclose(res[0][3], res[1][3], rtol=1e-12, atol=0)      # the reduced gradient is identical on both ranks


# ---------------------------------------------------------------------------------------------------------------------
# lock-step evaluation of optimiser chains (host logic of fit(batch=k); no device needed)
# ---------------------------------------------------------------------------------------------------------------------
def test_lockstep_chains_match_independent_runs():
    import scipy.optimize
    from gpcsd_amd.lockstep import run_chains

    def fg(x):
        return float((x[0] ** 2 - 1.0) ** 2 + 0.3 * x[0] + (x[1] - 0.5) ** 2), np.array([4 * x[0] * (x[0] ** 2 - 1) + 0.3,
                                                                                           2 * (x[1] - 0.5)])
    sizes = []

    def batch_fn(items):
        sizes.append(len(items))
        out = {}
        for k, x in items:
            out[k] = ValueError("poisoned start") if abs(x[0] - 7.0) < 1e-12 else fg(x)
        return out

    def chain(x0, evaluate):
        try:
            r = scipy.optimize.minimize(evaluate, x0, jac=True, method="L-BFGS-B", options={"maxiter": 100})
            return r.fun, r.x, r.nit
        except ValueError:
            return None
    starts = [np.array([s0, -s0]) for s0 in np.linspace(-2.0, 2.0, 9)] + [np.array([7.0, 0.0])]
    out, ev = run_chains(starts, chain, batch_fn, width=4)
    assert out[9] is None                                        # the failing chain ends alone, the others go on
    for i, s0 in enumerate(starts[:9]):
        ref = scipy.optimize.minimize(fg, s0, jac=True, method="L-BFGS-B", options={"maxiter": 100})
        assert out[i][0] == ref.fun and np.array_equal(out[i][1], ref.x)
    assert max(sizes) == 4 and ev.points == sum(sizes) and ev.batches == len(sizes)
    assert sum(1 for n in sizes if n == 4) > len(sizes) // 2      # most evaluations really were batched four wide


# ---------------------------------------------------------------------------------------------------------------------
# device-side gather of predictions ("direct ... all-gather across the 7 links")
# ---------------------------------------------------------------------------------------------------------------------
def _device_gather_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    from gpcsd_amd.dist import TrialSharding
    sh = TrialSharding(gather_dst=None)
    full = np.random.RandomState(5).standard_normal((2, 3, 7, 5))         # (C, nz, nt, all trials): 5 trials over 2 ranks = 3 + 2
    a, b = TrialSharding.block(5, rank, world)
    local = torch.from_numpy(np.ascontiguousarray(full[..., a:b]))
    every = sh.gather_trials_device(local)                                # dst=None: every rank copies the result out
    only0 = sh.gather_trials_device(local, dst=0)
    q.put((rank, bool(np.array_equal(every, full)), only0 is None or bool(np.array_equal(only0, full)), only0 is None, sh.d2h_bytes))
    td.destroy_process_group()


def test_device_gather_of_ragged_trial_blocks_world2():
    """TrialSharding.gather_trials_device -- one all_gather_into_tensor on the tensors' own device, the reorder to
    (..., all trials) there, one copy out on the gathering rank(s) -- with uneven trial blocks, here on CPU tensors over gloo
    (the RCCL run with device tensors is tests/test_resident_predictions.py on the GPU box)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_device_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    full_bytes = 2 * 3 * 7 * 5 * 8
    assert res[0][1] and res[1][1] and res[0][2] and res[1][2]
    assert not res[0][3] and res[1][3]                                   # dst=0: rank 1 gets nothing back
    assert res[0][4] == 2 * full_bytes and res[1][4] == full_bytes       # bytes copied out: the gathered result, on the gathering ranks only
