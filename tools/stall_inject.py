"""What makes the 10-30 ms stall of DESIGN 6 happen?  ONE model in steady state (its allocations long done), then ONE injected event
at a known time, and every step's completion time-stamped on the host: does a stall follow, and how long after the event?

    python tools/stall_inject.py <event> [workload]
events:  none | malloc (256 MB hipMalloc, kept) | mallocfree (256 MB hipMalloc + hipFree) | manysmall (200 x 1 MB hipMalloc + hipFree)
         hostalloc (64 MB page-locked block, kept) | hostallocfree | numpy (a 64 MB pageable array allocated, touched and freed)
         newctx (a second library context created and closed: streams, events, pinned blocks) | graphs (hipGraph re-instantiation:
         the model's buffers grown by one byte -> every captured chain retired and captured again)
Also runs a heartbeat on a SECOND python process (a 1-element kernel + synchronize in a loop, its own KFD process) when
PROBE_HEARTBEAT=1: does that one stall at the same wall-clock time (a device-wide stall) or not (this process's queues only)?
"""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "--heartbeat":          # the second process
    import torch
    torch.cuda.set_device(0)
    x = torch.zeros(64, device="cuda")
    t_end = time.time() + float(sys.argv[2])
    stamps = []
    while time.time() < t_end:
        x.add_(1.0)
        torch.cuda.synchronize()
        stamps.append(time.time())
    iv = np.diff(np.array(stamps))
    big = [(round(stamps[i + 1], 4), round(1e3 * iv[i], 2)) for i in np.nonzero(iv > 2e-3)[0]]
    print("HEARTBEAT " + json.dumps({"beats": len(stamps), "median_us": round(1e6 * float(np.median(iv)), 1), "over_2ms": big[:20]}), flush=True)
    sys.exit(0)

import bench
import torch
torch.cuda.set_device(0)
from gpcsd_amd import _hip

event = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
hb = None
if os.environ.get("PROBE_HEARTBEAT") == "1":
    hb = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--heartbeat", "9"], stdout=subprocess.PIPE, text=True)
w = bench.workload(name)
m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
m.set_device(0)
lfp = bench.synth_data(w, m, w["trials_per_gpu"], seed=1)
m.update_lfp(lfp, w["t"])
ctx = m._sync_device()
ctx.decomposition_cache(False)
ctx.pair_share_s(True)
z = w.get("z", w["x"])
hp, k1 = m._hparams(m.JITTER)
hp0, k0 = m._hparams(0.0)


def step():
    ctx.loglik_predict_async(hp, hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
    ctx.prefetch_pair(hp, hp0, z, w["t"])
    return ctx.loglik_parts_wait()


t_first = time.time()
for _ in range(60):
    step()
# well past anything the model's own allocations may have scheduled
while time.time() - t_first < 1.5:
    step()
ctx.synchronize()
keep = []
stamps = [time.time()]
t_event = None
for k in range(int(os.environ.get("PROBE_STEPS", "1800"))):
    if k == 300:
        t_event = time.time()
        if event == "malloc":
            keep.append(torch.empty(256 << 20, dtype=torch.uint8, device="cuda"))
        elif event == "mallocfree":
            a = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
            del a
            torch.cuda.empty_cache()
        elif event == "manysmall":
            aa = [torch.empty(1 << 20, dtype=torch.uint8, device="cuda") for _ in range(200)]
            del aa
            torch.cuda.empty_cache()
        elif event == "hostalloc":
            keep.append(_hip.pinned_pool.empty((8 << 20,)))
        elif event == "hostallocfree":
            a = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
            del a
        elif event == "numpy":
            a = np.ones(8 << 20)
            a += 1.0
            del a
        elif event == "newctx":
            c2 = _hip.Context(0)
            c2.close()
        elif event == "graphs":
            m2 = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
            m2.set_device(0)
            m2.update_lfp(lfp, w["t"])
            c2 = m2._sync_device()
            c2.decomposition_cache(False)
            for _ in range(4):
                c2.loglik_predict_async(hp, hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
                c2.loglik_parts_wait()
            c2.synchronize()
            keep.append((m2, c2))
        t_event_done = time.time()
        stamps.append(time.time())        # (the event's own duration is not a stall)
        continue
    step()
    stamps.append(time.time())
ctx.synchronize()
st = np.array(stamps)
iv = 1e3 * np.diff(st)
stalls = [{"at_ms_after_event": round(1e3 * (st[i + 1] - t_event), 1), "ms": round(float(iv[i]), 2)} for i in np.nonzero(iv > 2.0)[0] if i != 300]
out = {"event": event, "workload": name, "median_ms": round(float(np.median(iv)), 4), "event_took_ms": round(1e3 * (t_event_done - t_event), 2),
       "stalls_over_2ms": stalls, "event_wallclock": round(t_event, 4), "stall_wallclock": [round(float(st[i + 1]), 4) for i in np.nonzero(iv > 2.0)[0] if i != 300]}
print(json.dumps(out), flush=True)
if hb is not None:
    print(hb.communicate(timeout=30)[0].strip(), flush=True)
