"""The 5-25 ms stall of DESIGN 6 ("ONE gap per affected step loop, no kernel of the process running"), looked at from the host:
N successive models in ONE process, each running the bench's paired step loop; every step's completion is time-stamped on the
host and the largest interval of every loop reported.  One variable per run (environment switches come from the shell):

    python tools/stall_probe.py cfg2 8                 # baseline: each loop's model replaces the previous one (refcount frees it)
    python tools/stall_probe.py cfg2 8 --keep          # every model / context stays alive: nothing is freed between loops
    python tools/stall_probe.py cfg2 8 --close         # previous context closed explicitly, gc.collect(), device fence, 0.3 s pause
    python tools/stall_probe.py cfg2 8 --no-announce   # the library-default loop (no gpcsd_prefetch_pair, two spatial decompositions)
    python tools/stall_probe.py cfg2 8 --pin-lfp       # the trials uploaded from a page-locked block (no pageable H2D copy at all)
    python tools/stall_probe.py cfg2 8 --thp-off       # prctl(PR_SET_THP_DISABLE) before anything is mapped: no huge-page collapse
    python tools/stall_probe.py cfg2 8 --mempolicy     # set_mempolicy(MPOL_PREFERRED, node 0): the kernel's automatic NUMA
                                                       # balancing does not scan (unmap / migrate) mappings with an explicit policy
"""
import gc, json, os, sys, time
import ctypes
_libc = ctypes.CDLL(None, use_errno=True)
if "--thp-off" in sys.argv:
    rc = _libc.prctl(41, 1, 0, 0, 0)                                           # PR_SET_THP_DISABLE
    print("prctl(PR_SET_THP_DISABLE) ->", rc, ctypes.get_errno(), file=sys.stderr)
if "--mempolicy" in sys.argv:
    mask = ctypes.c_ulong(1)
    rc = _libc.syscall(238, 1, ctypes.byref(mask), 65)                         # x86-64 set_mempolicy(MPOL_PREFERRED, {0}, maxnode)
    print("set_mempolicy(MPOL_PREFERRED, node 0) ->", rc, ctypes.get_errno(), file=sys.stderr)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import torch
torch.cuda.set_device(0)
from gpcsd_amd import _hip

name, nloops = sys.argv[1], int(sys.argv[2])
flags = set(sys.argv[3:])
steps = int(os.environ.get("PROBE_STEPS", "200"))
w = bench.workload(name)
keepalive, out = [], []
prev = None
for loop in range(nloops):
    if "--close" in flags and prev is not None:
        prev[1].close()
        prev = None
        gc.collect()
        torch.cuda.synchronize()
        time.sleep(0.3)
    t_setup = time.perf_counter()
    m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
    m.set_device(0)
    lfp = bench.synth_data(w, m, w["trials_per_gpu"], seed=1000 + loop)
    if "--pin-lfp" in flags:
        pinned = _hip.pinned_pool.empty(lfp.shape)
        pinned[...] = lfp
        lfp = pinned
    m.update_lfp(lfp, w["t"])
    ctx = m._sync_device()
    ctx.decomposition_cache(False)
    announce = "--no-announce" not in flags
    ctx.pair_share_s(announce)
    z = w.get("z", w["x"])
    hp, k1 = m._hparams(m.JITTER)
    hp0, k0 = m._hparams(0.0)

    def step():
        ctx.loglik_predict_async(hp, hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
        if announce:
            ctx.prefetch_pair(hp, hp0, z, w["t"])
        return ctx.loglik_parts_wait()
    t_first_call = time.perf_counter()                    # the model's device allocations happen in its first evaluations
    warm_stamps = [t_first_call]
    for _ in range(60):
        step()
        warm_stamps.append(time.perf_counter())
    if "--settle" in flags:                               # keep stepping (untimed) until 0.45 s after the first evaluation
        while time.perf_counter() - t_first_call < 0.45:
            step()
            warm_stamps.append(time.perf_counter())
    ctx.synchronize()
    setup_s = time.perf_counter() - t_setup
    stamps = [time.perf_counter()]
    for _ in range(steps):
        step()
        stamps.append(time.perf_counter())
    ctx.synchronize()
    st = np.array(stamps)
    iv = 1e3 * np.diff(st)
    wiv = 1e3 * np.diff(np.array(warm_stamps))
    worst = int(np.argmax(iv))
    out.append({"loop": loop, "median_ms": round(float(np.median(iv)), 4), "mean_ms": round(float(iv.mean()), 4),
                "max_ms": round(float(iv.max()), 3), "at_step": worst, "over_2ms": int(np.sum(iv > 2.0)), "setup_s": round(setup_s, 2),
                "stall_s_after_first_evaluation": [round(float(st[i + 1] - t_first_call), 3) for i in np.nonzero(iv > 3.0)[0]],
                "untimed_stalls_s_after_first_evaluation": [(round(float(warm_stamps[i + 1] - t_first_call), 3), round(float(wiv[i]), 1))
                                                            for i in np.nonzero(wiv > 3.0)[0] if i >= 3]})
    if "--keep" in flags:
        keepalive.append((m, ctx, lfp))
    prev = (m, ctx)
    del m, ctx, lfp
env = {k: os.environ[k] for k in ("HSA_ENABLE_SDMA", "HSA_NO_SCRATCH_RECLAIM", "GPCSD_NO_GRAPH", "AMD_DIRECT_DISPATCH", "HIP_FORCE_DEV_KERNARG",
                                  "GPU_MAX_HW_QUEUES", "GPCSD_STREAM_POOL", "MALLOC_MMAP_THRESHOLD_", "HSA_ENABLE_INTERRUPT") if k in os.environ}
print(json.dumps({"workload": name, "flags": sorted(flags), "env": env, "stalled_loops": sum(1 for o in out if o["max_ms"] > 3.0),
                  "loops": out}), flush=True)
