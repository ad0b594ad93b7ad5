"""Does a model's step loop depend on what ran in the process before it?  (The cfg2 sub-result of the default bench line read
0.62 or 0.90 ms per step as the second workload of the cfg3 process, 0.627 as a command of its own.)
usage: python tools/two_models_probe.py first[,second,...]      e.g.  cfg2   |   cfg3,cfg2   |   cfg3,gc,cfg2   |  cfg2,cfg2"""
import argparse, gc, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
torch.cuda.set_device(0)
from gpcsd_amd import _hip

LOG = []
_init0, _close0 = _hip.Context.__init__, _hip.Context.close
def _init(self, *a, **k):
    _init0(self, *a, **k)
    LOG.append(("open", hex(self._h.value), [hex(h) for h in self.stream_handles()], self.stream_pool_stats()))
def _close(self):
    if getattr(self, "_h", None):
        LOG.append(("close", hex(self._h.value)))
    _close0(self)
_hip.Context.__init__, _hip.Context.close = _init, _close

CALLS = []
def _timed(name):
    f0 = getattr(_hip.Context, name)
    def f(self, *a, **k):
        t0 = time.perf_counter()
        r = f0(self, *a, **k)
        CALLS.append((time.perf_counter() - t0, name, len(CALLS)))
        return r
    setattr(_hip.Context, name, f)
for _n in ("loglik_predict_async", "loglik_parts_wait", "prefetch_pair", "make_hparams", "synchronize"):
    _timed(_n)

def live_contexts():
    return sum(1 for o in gc.get_objects() if isinstance(o, _hip.Context))

def run(name):
    a = argparse.Namespace(gpus=1, steps=100, warmup=5, workload=name, trials_per_gpu=None, setup_steps=60, no_cpu_baseline=True,
                           only_value=True, no_sub_results=True, cpu_budget_s=45.0, fit_batch=None, fit_maxiter=15, fit_groups=1,
                           n1_value=None, sub_result=None)
    r = bench.run_step_bench(a, bench.workload(name), 0, 1, 0, "nccl", cpu_legs=[])
    return r["ms_per_step"]

out = []
for item in sys.argv[1].split(","):
    if item == "gc":
        gc.collect()
        out.append(("gc", live_contexts()))
    elif item == "gcoff":
        gc.collect()
        gc.disable()
        out.append(("gcoff", gc.get_count()))
    elif item == "sleep":
        time.sleep(2.0)
        out.append(("sleep", 2.0))
    else:
        g0 = [st["collections"] for st in gc.get_stats()]
        del CALLS[:]
        ms = run(item)
        slow = sorted(CALLS, reverse=True)[:3]
        tot = {}
        for d, n, i in CALLS:
            tot[n] = tot.get(n, 0.0) + d
        g1 = [st["collections"] for st in gc.get_stats()]
        out.append((item, round(ms, 4), "gc %s" % [b - a for a, b in zip(g0, g1)],
                    "slowest calls (ms, name, index of %d): %s" % (len(CALLS), [(round(1e3 * d, 2), n, i) for d, n, i in slow]),
                    "total ms per call kind: %s" % {n: round(1e3 * v, 1) for n, v in tot.items()}))
def kfd_evicted():
    import glob
    d = {}
    for f in glob.glob("/sys/class/kfd/kfd/proc/%d/stats_*/evicted_ms" % os.getpid()):
        try:
            d[f.split("/")[-2]] = int(open(f).read().strip())
        except (OSError, ValueError) as e:
            d[f] = repr(e)
    return d
def vm():
    want = ("thp_collapse_alloc", "compact_stall", "pgmigrate_success", "thp_fault_alloc", "compact_migrate_scanned")
    return {k: int(v) for k, v in (ln.split() for ln in open("/proc/vmstat")) if k in want}
print("# kfd evicted_ms of this process:", kfd_evicted(), " vmstat:", vm(), file=sys.stderr)
if os.environ.get("PROBE_LOG") == "1":
    for e in LOG:
        print("#", e)
print(json.dumps({"hwq": os.environ.get("GPU_MAX_HW_QUEUES"), "seq": out}), flush=True)
