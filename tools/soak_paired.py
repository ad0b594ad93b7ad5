"""Soak of the queued call forms: N steps with hyper-parameters that change every step, the paired / queued / two-deep forms
against the same calls fenced one by one -- every log-likelihood and the final predictions bit for bit.  Mixes the forms at
random, so generations, chain streams and the result ring see every hand-over; three paired calls in four announce a next step
(gpcsd_prefetch_pair), one in five of those a step that does not come.   python tools/soak_paired.py [cfg3|cfg2] [N] [tri] [R]
(R resident trials, default 6; from 16 on the prediction takes its tridiagonal form too)
("tri": with the shifted-tridiagonal log-likelihood forced on, gpcsd_ll_tridiag mode 1 -- its fenced values are also held to
1e-12 of the eigenvector form's; without it the eigenvector form is forced)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpcsd_amd import _hip

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
TRI = len(sys.argv) > 3 and sys.argv[3] == "tri"
w = bench.workload(name)
m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
R = int(sys.argv[4]) if len(sys.argv) > 4 else 6
lfp = bench.synth_data(w, m, R, seed=5)
m.update_lfp(lfp, w["t"])
ctx = m._sync_device()
z = np.ascontiguousarray(w["x"])
shape = (z.shape[0], w["nt"], R)
rs = np.random.RandomState(0)
base_ell = m.temporal_cov_list[0].params["ell"]["value"]
base_R = m.R["value"]
thetas = [(base_ell * rs.uniform(0.8, 1.25), base_R * rs.uniform(0.9, 1.1), bool(rs.rand() < 0.3)) for _ in range(N)]
kinds = [(int(rs.choice([_hip.PRED_CSD, _hip.PRED_BOTH, _hip.PRED_LFP])), bool(rs.rand() < 0.7)) for _ in range(N)]   # (type, lists)
kinds[-1] = (_hip.PRED_BOTH, True)               # the last step leaves every output buffer for the final comparison


def hps(k):
    ell, Rv, other = thetas[k]
    m.temporal_cov_list[0].params["ell"]["value"] = ell
    m.R["value"] = Rv
    h1 = m._hparams(m.JITTER)
    if other:                                   # predict at another spatial set than loglik: no shared Gram assembly
        m.R["value"] = Rv * 1.01
    h0 = m._hparams(0.0)
    return h1, h0


ctx.decomposition_cache(False)
ctx.ll_tridiag(0)
if TRI:
    eig_ll = [ctx.loglik_parts(hps(k)[0][0]) for k in range(N)]
    ctx.ll_tridiag(1)
ref_ll, ref_pred = [], None
for k in range(N):
    h1, h0 = hps(k)
    ref_ll.append(ctx.loglik_parts(h1[0]))
    ctx.predict_resident(h0[0], z, w["t"], kinds[k][0], want_lists=kinds[k][1])
    ctx.synchronize()
C = len(m.temporal_cov_list)
outs = [("pred_out_csd", shape), ("pred_out_lfp", shape), ("pred_out_csd_list", (C,) + shape), ("pred_out_lfp_list", (C,) + shape)]
ref_pred = [ctx.fetch(nm, sh).copy() for nm, sh in outs]
if TRI:
    dev = max(abs(sum(a[:2]) - sum(b[:2])) / abs(sum(b[:2])) for a, b in zip(ref_ll, eig_ll))
    print("tridiagonal form vs eigenvector form: %d of %d log-likelihoods took it, max relative deviation %.2e" % (
        ctx.ll_tridiag(), N, dev), flush=True)
    if ctx.ll_tridiag() < N or not dev < 1e-12:
        sys.exit(1)

for cache in (False, True):
    ctx.decomposition_cache(cache)
    got, outstanding, keep, announced = [], 0, [], 0
    q0, t0 = ctx.prefetch_stats()
    for k in range(N):
        h1, h0 = hps(k)
        keep.append((h1, h0))
        form = rs.randint(4)
        if form == 0:                           # fenced
            while outstanding:
                got.append(ctx.loglik_parts_wait()); outstanding -= 1
            got.append(ctx.loglik_parts(h1[0]))
            ctx.predict_resident(h0[0], z, w["t"], kinds[k][0], want_lists=kinds[k][1])
            if rs.rand() < 0.5:
                ctx.synchronize()
        elif form == 1:                         # two queued calls
            ctx.loglik_parts_async(h1[0]); outstanding += 1
            ctx.predict_resident(h0[0], z, w["t"], kinds[k][0], want_lists=kinds[k][1])
        else:                                   # paired call
            ctx.loglik_predict_async(h1[0], h0[0], z, w["t"], kinds[k][0], want_lists=kinds[k][1]); outstanding += 1
            u = rs.rand()
            if u < 0.75 and k + 1 < N:          # ... that announces the next step (gpcsd_prefetch_pair): truthfully, or -- one in five --
                ka = k + 1 if u < 0.6 else int(rs.randint(N))        # some other step's hyper-parameters (dropped by whatever comes next)
                a1, a0 = hps(ka)
                keep.append((a1, a0))
                ctx.prefetch_pair(a1[0], a0[0], z, w["t"])
                announced += 1
        while outstanding > (2 if form == 3 else 0) or outstanding >= 4:      # form 3 leaves up to two outstanding
            got.append(ctx.loglik_parts_wait()); outstanding -= 1
    while outstanding:
        got.append(ctx.loglik_parts_wait()); outstanding -= 1
    pred = [ctx.fetch(nm, sh) for nm, sh in outs]
    same = all(np.array_equal(a, b) for a, b in zip(pred, ref_pred))
    bad = [k for k in range(N) if got[k] != ref_ll[k]]
    q1, t1 = ctx.prefetch_stats()
    print("%s cache=%s: %d steps, log-likelihood mismatches %d, predictions (csd, lfp, both lists) %s; %d announcements, %d taken over" % (
        name, cache, N, len(bad), "identical" if same else "DIFFER", q1 - q0, t1 - t0), flush=True)
    if bad or not same:
        print("first mismatches:", bad[:5])
        sys.exit(1)
print("soak ok")
