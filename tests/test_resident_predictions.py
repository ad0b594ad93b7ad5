"""Predictions that stay on the GPU, and the announced loop as a class-API caller (round 6; -m gpu).

  * GPCSD{1,2}D.predict(..., resident=True): zero-copy device views of the posterior means (gpcsd1d.py:286-293 / gpcsd2d.py:327-334),
    bit for bit the host arrays of predict();
  * loglik_predict_many: loglik() + predict() over a LIST of hyper-parameter sets, every paired call announcing the next one
    (gpcsd_prefetch_pair) -- the mode bench.py times as `value` -- at the headline configuration (384 x 500 x 50), with one
    spatial decomposition per pair (gpcsd_pair_share_s), against the ORACLE: loglik 1e-9, csd and both component lists 1e-6;
  * trial-sharded predictions gathered on the device over RCCL (`nccl`, one rank on the one-GPU box): no host staging -- the only
    device-to-host traffic is the gathered result;
  * three models opened one after another in one process: an interval above 3 ms only inside a model's first 0.45 s -- the window bench.py settles through.
"""
import os
import socket
import sys
import time

import numpy as np
import pytest
import torch.multiprocessing as mp

from oracle import gpcsd_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _step_model(R, name="cfg3", seed=11):
    import bench
    w = bench.workload(name)
    m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
    lfp = bench.synth_data(w, m, R, seed=seed)
    m.update_lfp(lfp, w["t"])
    return w, m, lfp


@pytest.mark.parametrize("name,R,kind", [("cfg2", 12, "both"), ("cfg3", 8, "csd")])
def test_resident_predict_gives_device_views_with_the_bits_of_the_host_arrays(name, R, kind):
    import torch
    w, m, lfp = _step_model(R, name)
    z = w.get("z", w["x"])
    m.predict(z, w["t"], type=kind)
    host = {"csd": np.array(m.csd_pred), "csd_list": [np.array(a) for a in m.csd_pred_list]}
    if kind == "both":
        host["lfp"] = np.array(m.lfp_pred)
    m.predict(z, w["t"], type=kind, resident=True)
    assert hasattr(m.csd_pred, "__cuda_array_interface__") and m.csd_pred.shape == host["csd"].shape
    dev = torch.as_tensor(m.csd_pred, device="cuda")
    assert dev.is_cuda and dev.dtype == torch.float64 and dev.data_ptr() == m.csd_pred.ptr          # zero copy
    assert np.array_equal(dev.cpu().numpy(), host["csd"])
    for c, a in enumerate(host["csd_list"]):
        assert np.array_equal(torch.as_tensor(m.csd_pred_list[c], device="cuda").cpu().numpy(), a)
    assert np.array_equal(m.csd_pred.numpy(), host["csd"])                                         # the pool-backed host copy
    if kind == "both":
        assert np.array_equal(torch.as_tensor(m.lfp_pred, device="cuda").cpu().numpy(), host["lfp"])
    assert np.array_equal(np.asarray(m.x_pred), np.asarray(z)) and np.array_equal(np.asarray(m.t_pred), np.asarray(w["t"]))


@pytest.mark.parametrize("name,R,kind", [("cfg3", 7, "both"), ("cfg2", 5, "csd")])
def test_the_64_orbit_unfold_writes_the_bits_of_the_32_orbit_one(name, R, kind, monkeypatch):
    """GPCSD_UNFOLD_BF=2 (gemm_f64.hip: gemm_pred_unfold_kernel<CC, 2>, opt-in: fewer S~ fetches, slower launch) changes which
    workgroup owns an output, not the order any element is accumulated in: predictions equal bit for bit (ragged trial count,
    250 and 40 time orbits: the second does not fill a 64-orbit tile)."""
    w, m, lfp = _step_model(R, name, seed=5)
    z = w.get("z", w["x"])
    m.predict(z, w["t"], type=kind)
    ref = [np.array(m.csd_pred)] + [np.array(a) for a in m.csd_pred_list] + ([np.array(m.lfp_pred)] if kind == "both" else [])
    monkeypatch.setenv("GPCSD_UNFOLD_BF", "2")
    m.predict(z, w["t"], type=kind)
    got = [np.array(m.csd_pred)] + [np.array(a) for a in m.csd_pred_list] + ([np.array(m.lfp_pred)] if kind == "both" else [])
    assert all(np.array_equal(a, b) for a, b in zip(ref, got))
    assert np.isfinite(ref[0]).all() and np.abs(ref[0]).max() > 0


def test_loglik_predict_many_announced_with_one_spatial_decomposition_at_cfg3_x_50_vs_oracle():
    """The mode `value` is timed in (announcements + one spatial decomposition per pair) through its class-API caller, at 50 trials,
    against the oracle on every set's log-likelihood (1e-9) and the last set's csd and BOTH per-component lists (1e-6); then the
    same list with the library's defaults is bit for bit the fenced loglik() / predict() of the class API."""
    import bench
    R = 50
    w, m, lfp = _step_model(R)
    O_, geom, hp, hp0 = bench.oracle_setup(w, m)
    ctx = m._sync_device()
    ctx.decomposition_cache(False)
    z = w["x"]
    base = m.extract_model_params()
    sets = []
    for k, (e0, e1) in enumerate([(20.0, 5.0), (23.0, 4.5), (18.0, 5.5)]):
        p = dict(base, temporal_ell_list=[e0, e1], sig2n=0.05 + 0.01 * k)
        sets.append(p)
    q0, t0 = ctx.prefetch_stats()
    s0 = ctx.pair_share_s()
    lls = m.loglik_predict_many(sets, z, w["t"], type="csd", share_spatial=True)
    q1, t1 = ctx.prefetch_stats()
    assert q1 - q0 == 2 and t1 - t0 == 2                       # every set but the last announced its successor, and was taken
    assert ctx.pair_share_s() - s0 == 3 and not ctx.pair_share_s_on()      # one spatial decomposition per pair; switch restored
    got = np.array(m.csd_pred)
    got_list = [np.array(a) for a in m.csd_pred_list]
    worst = 0.0
    for k, p in enumerate(sets):
        temporal = [(kd, ell, s2) for (kd, _, s2), ell in zip(hp["temporal"], p["temporal_ell_list"])]
        hk = O.make_hparams(hp["R"], hp["ell_s"], temporal, p["sig2n"], eps=hp["eps"], jitter=hp["jitter"])
        ref = O.loglik(geom, hk, lfp)
        worst = max(worst, abs(lls[k] - ref) / abs(ref))
        assert abs(lls[k] - ref) <= 1e-9 * abs(ref), (k, lls[k], ref)
    hk0 = dict(hk, jitter=0.0)
    ref = O.predict(geom, hk0, lfp, z, w["t"], type="csd")
    e_csd = np.max(np.abs(got - ref["csd"])) / np.max(np.abs(ref["csd"]))
    e_list = [np.max(np.abs(got_list[c] - ref["csd_list"][c])) / np.max(np.abs(ref["csd_list"][c])) for c in range(2)]
    print("announced + shared spatial side, 50 trials: loglik %.1e, csd %.1e, lists %s" % (worst, e_csd, e_list))
    assert e_csd <= 1e-6 and max(e_list) <= 1e-6
    assert m.extract_model_params()["temporal_ell_list"] == sets[-1]["temporal_ell_list"]
    # library defaults (two spatial decompositions): the announced list is bit for bit the class API's fenced calls
    lls_def = m.loglik_predict_many(sets, z, w["t"], type="csd")
    pred_def = np.array(m.csd_pred)
    for k, p in enumerate(sets):
        m.restore_model_params(p)
        assert float(m.loglik()) == lls_def[k]
    m.predict(z, w["t"], type="csd")
    assert np.array_equal(np.asarray(m.csd_pred), pred_def)
    # ... and resident=True leaves the last set's means on the device
    m.loglik_predict_many(sets[:2], z, w["t"], type="csd", resident=True)
    assert hasattr(m.csd_pred, "__cuda_array_interface__") and m.csd_pred.shape == pred_def.shape


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rccl_gather_worker(port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch
    import torch.distributed as td
    torch.cuda.set_device(0)
    td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from gpcsd_amd.dist import TrialSharding
    import test_hip_parity as T
    m, c, g, geom, hp, lfp = T._build_model("2d_npx_96x120x3")
    m.predict(c["x"], c["t"], type="both")
    plain = {k: np.array(getattr(m, k)) for k in ("csd_pred", "lfp_pred")}
    plain_list = [np.array(a) for a in m.csd_pred_list]
    sh = TrialSharding(gather_predictions=True, gather_dst=0)
    assert sh.on_device()
    m.shard_trials(sh)
    ll = float(m.loglik())
    m.predict(c["x"], c["t"], type="both")
    ok = all(np.array_equal(np.asarray(getattr(m, k)), plain[k]) for k in plain)
    ok = ok and all(np.array_equal(np.asarray(a), b) for a, b in zip(m.csd_pred_list, plain_list))
    nz, nt, R = plain["csd_pred"].shape
    expect = (2 + 2 * len(plain_list)) * nz * nt * R * 8          # csd, lfp and their per-component lists: the result itself, once
    q.put((ok, int(sh.d2h_bytes), int(expect), ll, float(g["loglik"])))
    td.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_predictions_are_gathered_on_the_device_over_rccl_without_host_staging():
    """model.shard_trials(TrialSharding(gather_predictions=True)) over `nccl` (= RCCL): predict() leaves the means in HBM, the trial
    blocks are all-gathered as device tensors and the gathering rank copies the result out once (one rank here: the RCCL code path
    with device tensors on the box's single GPU; N > 1 over xGMI is the driver's 8-GPU node).  Bytes copied device -> host = the
    result, nothing else; values bit for bit the unsharded predict()."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_gather_worker, args=(_free_port(), q))
    p.start()
    ok, d2h, expect, ll, ll_ref = q.get(timeout=480)
    p.join(timeout=60)
    assert ok and d2h == expect, (ok, d2h, expect)
    assert abs(ll - ll_ref) <= 1e-6 * abs(ll_ref)


def test_three_models_in_one_process_stall_only_inside_the_settle_window():
    """DESIGN 6 (the 10-30 ms stall), as characterised in round 6: about every second model of a process sees ONE interval of
    9 / 19 / 29 ms in which none of its queues make progress, 35-110 ms after the model's first evaluation -- never later, and never
    caused by allocations, frees or new contexts while a model is in steady state (tools/stall_probe.py, tools/stall_inject.py).
    Three models opened one after another (cfg2's shape), every paired step's completion time-stamped on the host: an interval
    above 3 ms may only end inside the first 0.45 s of its model's life (measured: 35-110 ms), and the 300 steps after that contain
    none -- the window bench.py's step loops keep out of their timed region (bench.SETTLE_S)."""
    import gc
    import bench
    from gpcsd_amd import _hip
    assert bench.SETTLE_S >= 0.45
    seen = []
    for loop in range(3):
        w, m, lfp = _step_model(200, "cfg2", seed=100 + loop)
        ctx = m._sync_device()
        ctx.decomposition_cache(False)
        assert ctx.bounce_stats() >= lfp.nbytes          # (pageable arrays never reach the runtime: ctx.hpp copy_in / copy_out)
        hp, k1 = m._hparams(m.JITTER)
        hp0, k0 = m._hparams(0.0)

        def step():
            ctx.loglik_predict_async(hp, hp0, w["x"], w["t"], _hip.PRED_CSD, want_lists=True)
            return ctx.loglik_parts_wait()
        t0 = time.perf_counter()
        stamps = [t0]
        while stamps[-1] - t0 < 0.45:
            step()
            stamps.append(time.perf_counter())
        n_early = len(stamps)
        for _ in range(300):
            step()
            stamps.append(time.perf_counter())
        ctx.synchronize()
        st = np.array(stamps)
        iv = 1e3 * np.diff(st)
        big = [(round(float(st[i + 1] - t0), 3), round(float(iv[i]), 1)) for i in np.nonzero(iv > 3.0)[0] if i >= 3]   # (the first calls allocate and capture)
        late = [(round(float(st[i + 1] - t0), 3), round(float(iv[i]), 1)) for i in np.nonzero(iv > 3.0)[0] if i >= n_early - 1]
        seen.append({"stalls (s after first evaluation, ms)": big, "median_ms": round(float(np.median(iv)), 3)})
        # (measured: every stall ended 35-110 ms after the first evaluation; the gate is the window bench.py keeps clear of)
        assert all(t_end <= 0.45 for t_end, _ in big), seen
        assert not late, seen
        del m, ctx
        gc.collect()
    print("per model:", seen)
