#!/usr/bin/env python3
"""bench.py -- GPCSD log-marginal-likelihood + posterior-predict throughput on MI355X.

Default workload (BASELINE.json configs[2]/[3]): GPCSD2D, 384-channel Neuropixels checkerboard x 500 time
points, 50 synthetic trials PER GPU (cfg3 at N=1; cfg4 = 400 trials at N=8: weak scaling), float64, ngl 20x60,
SE + Matern-1/2 temporal kernels.  One step = one loglik() evaluation over the resident trials + one
predict(z = electrodes, t, type="csd") of every resident trial, inputs resident in HBM, outputs left in HBM
(no PCIe in the timed region; the PCIe-inclusive rate of the class-API predict() is reported separately).

    python bench.py --gpus 1 --steps 100 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --workload cfg2          # GPCSD1D 24 x 500 x 200 trials (BASELINE configs[1]), same step
    python bench.py --workload cfg3fit       # GPCSD2D.fit() at the headline geometry: objective + analytic gradient evaluations
    python bench.py --workload cfg5          # GPCSD1D fit: restarts evaluated in lock-step batches (BASELINE configs[4])
    python bench.py --only-value             # setup + warm-up + timed loop and nothing else: the command the rocprofv3 kernel
                                             # stats / PMC passes under profiles/ are taken over (tools/profile_r06.sh)
The default cfg3 line at N=1 also carries compact cfg2 / cfg3fit / cfg5 / npx69 / aud24 / potrf sub-results (each a child process of
its own) so that the driver's record holds them; --no-sub-results skips them.  The measuring code lives in benchlib/ (workloads,
step, fit, cpu, profiles, line); this file is the contract: CLI, rank launch, sub-results, the printed line.

Rank 0 prints ONE JSON line.
  roofline      step level, as SURVEY 8(d) specifies: flops per step / ms_per_step against the fp64 MFMA peak, once in flops
                actually launched (folded basis, symmetry-folded eigensolver) and once in flops of the reference's algorithm;
                beside it the kernel with the largest share of GPU time (the single-workgroup tridiagonalisation tail) with
                its own launch time, rate and CU occupancy, and the largest GEMM launch.  Launch times are HIP events on the
                library's own streams around the kernels of the SAME paired, queued step the timed loop runs (asynchronous
                scopes, gpcsd_prof_enable mode 2: chains launched eagerly so that the scopes inside them record; mode 3: chains
                replayed as hipGraphs as in the timed loop, chain-level scopes).
  cpu_baseline  the NumPy oracle (a port: the Python reference cannot travel to the GPU box) on the SAME number of trials
                as the GPU step, BLAS threads swept, >= 20 loglik / >= 3 predict repetitions at the best setting, plus the
                single-thread figure and the reference's strided per-trial layout priced next to the contiguous one.
  setup_steps   untimed steady-state setup evaluations in front of the W warm-up steps (graph capture, clocks).
"""
import argparse
import json
import os
import sys
import time

# A context keeps four streams busy; torch's side stream and RCCL's own bring more.  The HIP runtime multiplexes
# streams onto 4 hardware queues by default, and streams that share one serialise: the 8-byte all-reduce of a multi-rank step
# then waits behind a whole predict tail (3.7 instead of 1.94 ms per step).  Must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


# The parts (names re-exported: tests and tools use bench.workload, bench.build_model, bench.synth_data, ...)
from benchlib.workloads import (FP64_MFMA_SPEC_TFLOPS, HBM_PEAK_GBS, N_CUS, SETTLE_S, algorithmic_flops, build_model,  # noqa: E402,F401
                                neuropixels_xy, oracle_setup, synth_data, workload)
from benchlib.cpu import cpu_baseline  # noqa: E402,F401
from benchlib.profiles import PROFILE_ROUND, pmc_step_traffic, rocprof_kernel  # noqa: E402,F401
from benchlib.line import DETAIL_FILE, LINE_LIMIT, _NESTED_KEYS, _TOP_KEYS, compact_record, sub_headlines  # noqa: E402,F401
from benchlib.step import run_step_bench  # noqa: E402,F401
from benchlib.fit import _quiesce_host, potrf_bench, run_fit_bench  # noqa: E402,F401


# ------------------------------------------------------------------------------------------------------- launcher
def launch_ranks(n, argv):
    """Start `python -m torch.distributed.run --nnodes=1 --nproc-per-node n bench.py <argv>` as a child process (one rank per
    GPU, rendezvous on 127.0.0.1 at a free port), pass its output through and return its exit code.  The caller is a process
    that has made no GPU call; nothing is exec'ed and nothing is retried."""
    import socket
    import subprocess
    backend = os.environ.get("GPCSD_BENCH_BACKEND", "nccl")
    if backend == "nccl" and "GPCSD_DEVICE" not in os.environ:
        import torch                                   # device_count() does not initialise the runtime on this image
        have = torch.cuda.device_count()
        if have < n:
            print("bench.py: --gpus %d but this node shows %d GPU(s); one rank per GPU over RCCL needs %d (rehearsal on fewer "
                  "cards: GPCSD_BENCH_BACKEND=gloo GPCSD_DEVICE=0)" % (n, have, n), file=sys.stderr)
            return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    other = [ln for ln in r.stdout.splitlines() if not ln.startswith("{")]
    if other:
        print("\n".join(other), file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    elif r.returncode == 0:
        print("bench.py: the ranks exited 0 without a result line", file=sys.stderr)
        return 1
    return r.returncode


# ------------------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--trials-per-gpu", type=int, default=None)
    ap.add_argument("--setup-steps", type=int, default=150)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only-value", action="store_true",
                    help="setup + warm-up + timed loop, then print value / ms_per_step and exit: the command profiled under profiles/")
    ap.add_argument("--no-sub-results", action="store_true", help="skip the compact cfg2 / cfg5 sub-results of the default line")
    ap.add_argument("--sub-result", default=None, choices=SUB_RESULT_KEYS,
                    help="measure one sub-result of the default line and print its dict (what the default run starts as a child)")
    ap.add_argument("--cpu-budget-s", type=float, default=45.0)
    ap.add_argument("--fit-batch", type=int, default=None, help="cfg5: restarts evaluated per lock-step batch")
    ap.add_argument("--fit-maxiter", type=int, default=15)
    ap.add_argument("--fit-groups", type=int, default=1,
                    help="cfg5: lock-step groups running side by side on one GPU, each on its own context and host thread "
                         "(measured: one large batch beats several groups -- 7.5 k evals/s at 1 x 32 against 6.3 k at 2 x 16)")
    ap.add_argument("--n1-value", type=float, default=None,
                    help="N > 1: the N = 1 `value` of this workload (trials/s) to quote scaling_efficiency against; without it the "
                         "efficiency is quoted against the ranks' own rates without collectives, measured in the same processes")
    args = ap.parse_args()

    # `python bench.py --gpus N` (no launcher): start the N ranks ourselves.  This process has not touched the GPU (torch is not
    # even imported yet) and never will: the ranks are a fresh `torch.distributed.run` child, whose rank 0 prints the line.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks: they must agree (run "
                         "`python bench.py --gpus N` and let it start the ranks, or launch N ranks with --gpus N)" % (args.gpus, world))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs (one-GPU box): GPCSD_BENCH_BACKEND=gloo + GPCSD_DEVICE=0 run N ranks on one card
    backend = os.environ.get("GPCSD_BENCH_BACKEND", "nccl")
    if "GPCSD_DEVICE" in os.environ:
        local_rank = int(os.environ["GPCSD_DEVICE"])
    import torch
    torch.cuda.set_device(local_rank)
    # keep this rank's host threads (those running now; RCCL's and the library's inherit) on the CPUs of its GPU's NUMA node: a
    # host process that migrates between the sockets runs the same queued step at 1.12 .. 1.25 ms from run to run, a bound one
    # at 1.13 (GPCSD_BENCH_NUMA_BIND=0 leaves the affinity alone).  After torch has initialised HIP: torch's
    # bundled runtime does not come up once another copy of the runtime (the library's) has been initialised first.
    host_numa = None
    if os.environ.get("GPCSD_BENCH_NUMA_BIND", "1") != "0":
        from gpcsd_amd import _hip as _hip_mod
        host_numa = _hip_mod.bind_host_to_device_numa(local_rank)
    # GPCSD_BENCH_FORCE_DIST=1: initialise the process group and shard even with one rank -- the only way to drive the RCCL
    # code path (device tensors, broadcast, async all-reduce, barrier) on a one-GPU box
    force_dist = os.environ.get("GPCSD_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as td
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            td.init_process_group(backend, rank=rank, world_size=world)

    if args.sub_result is not None:                   # one sub-result of the default line, as a command of its own
        legs = []
        r = sub_result_one(args, args.sub_result, local_rank, backend, legs)
        for leg in legs:
            leg()
        sys.stdout.flush()
        print(SUB_RESULT_MARK + json.dumps(r, default=_JSON_DEFAULT), flush=True)
        return
    if args.workload == "potrf":                      # the dense Cholesky path on its own (the command profiled as r04_*_potrf)
        if rank == 0:
            emit(potrf_bench())
        return
    w = workload(args.workload)
    # Every GPU measurement of this process runs BEFORE any CPU leg (oracle parity checks, the CPU baseline's BLAS-thread sweep):
    # the run_* functions append their CPU legs to `cpu_legs` and main() runs them last.  (Round 4's driver record read cfg2 at
    # 1.12 ms per step for 0.86: its sub-result ran after the CPU baseline, whose OpenBLAS workers were still spinning on the CPUs
    # the launch thread is bound to.)
    cpu_legs = []
    if args.workload in FIT_WORKLOADS:
        out = run_fit_bench(args, w, rank, world, local_rank, backend, cpu_legs=cpu_legs)
    else:
        out = run_step_bench(args, w, rank, world, local_rank, backend, cpu_legs=cpu_legs)
    # the driver runs `bench.py --gpus 1` only: carry compact cfg2 / cfg5 / potrf / npx69 / aud24 results with that run (N=1, a few
    # seconds each); their headline scalars go into `config` of the line, the full dicts into the detail file
    sub = None
    if (rank == 0 and out is not None and world == 1 and args.workload == "cfg3" and args.trials_per_gpu is None
            and not args.only_value and not args.no_sub_results):
        sub = sub_results(args, local_rank, backend, cpu_legs)
    for leg in cpu_legs:
        leg()
    if sub is not None:
        out["sub_results"] = sub
        out["config"].update(sub_headlines(sub))
    if rank == 0 and out is not None:
        out["host_affinity"] = ({"bound_to_numa_node": host_numa["node"], "cpus": host_numa["cpus"], "device_pci": host_numa["pci"]}
                                if host_numa else {"bound_to_numa_node": None, "cpus": len(os.sched_getaffinity(0))})
        emit(out)
    if world > 1 or force_dist:                       # the line is out; leave the communicator the orderly way (no exit-time warning)
        import torch.distributed as td
        try:
            if td.is_initialized():
                td.destroy_process_group()
        except Exception as e:
            print("bench.py: destroy_process_group: %r" % (e,), file=sys.stderr)


def emit(full):
    """Write the full result dict to DETAIL_FILE beside the script (and under gpurun_out/ when that exists, so that a gpurun call
    brings it home), then print the compact line -- the last thing on stdout."""
    blob = json.dumps(full, indent=1, default=_JSON_DEFAULT)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, DETAIL_FILE), "w") as fh:
                    fh.write(blob)
            except OSError as e:
                print("bench.py: could not write %s: %s" % (os.path.join(d, DETAIL_FILE), e), file=sys.stderr)
    sys.stdout.flush()
    print(compact_record(full), flush=True)


SUB_RESULT_KEYS = ("cfg2", "cfg3fit", "cfg5", "npx69", "aud24", "potrf")
FIT_WORKLOADS = ("cfg5", "npx69fit", "aud24", "cfg3fit")


def sub_result_one(args, key, local_rank, backend, cpu_legs):
    """One sub-result of the default line, measured by the same functions `--workload <name>` runs.  GPU work only: every parity
    check against the oracle is appended to `cpu_legs` (the caller runs them after the last GPU measurement)."""
    import copy

    def step(name, steps, warmup, setup):
        a = copy.copy(args)
        a.workload, a.steps, a.warmup, a.setup_steps, a.no_cpu_baseline, a.trials_per_gpu = name, steps, warmup, setup, True, None
        r = run_step_bench(a, workload(name), 0, 1, local_rank, backend, compact=True, cpu_legs=cpu_legs)
        r["fenced_calls_ms"] = [r["fenced_calls"]["loglik_ms"], r["fenced_calls"]["predict_resident_ms"]]
        r["loglik_evals_per_sec"] = r["fenced_calls"]["loglik_evals_per_sec_per_gpu"]
        r["predict_trials_per_sec"] = r["fenced_calls"]["predict_trials_per_sec_per_gpu"]
        r["roofline_frac_step_executed"] = r["roofline"]["frac"]
        for k in ("roofline", "pipelining", "fenced_calls", "two_steps_in_flight", "with_decomposition_cache", "class_api_predict_note"):
            r.pop(k, None)                                         # (the sub-result keeps its scalars; r is what the CPU leg fills in)
        return r

    def fit(name):
        a = copy.copy(args)
        a.workload, a.steps, a.warmup, a.setup_steps, a.fit_batch, a.fit_groups = name, 40, 3, 30, None, 1
        r = run_fit_bench(a, workload(name), 0, 1, local_rank, backend, compact=True, cpu_legs=cpu_legs)
        roof = r.pop("roofline", None) or {}
        r["roofline_frac_step_executed"], r["hbm_traffic_bytes_per_step"] = roof.get("frac"), roof.get("traffic")
        if isinstance(r.get("fit"), dict):
            r["fit"].pop("nll_values", None)
        return r

    def npx():
        # the reference's own 2D workload shape (69 channels without mirror symmetry) beside a point-symmetric control
        r69, r72 = step("npx69", 200, 5, 150), step("npx72sym", 200, 5, 150)
        rf = fit("npx69fit")
        r69["fit"] = rf
        r69["symmetric_control_72ch"] = r72
        r69["step_over_symmetric_control"] = r69["ms_per_step"] / r72["ms_per_step"]
        r69["headline"] = {"trials_per_sec": r69["value"], "ms_per_step": r69["ms_per_step"],
                           "loglik_evals_per_sec": r69["loglik_evals_per_sec"], "fit_evals_per_sec": rf["value"],
                           "step_over_symmetric_control": r69["step_over_symmetric_control"]}
        return r69

    if key == "cfg2":
        return step("cfg2", 100, 5, 60)
    if key == "cfg5":
        return fit("cfg5")
    if key == "cfg3fit":                               # GPCSD2D.fit() at the headline geometry: objective + analytic gradient
        return fit("cfg3fit")
    if key == "npx69":
        return npx()
    if key == "aud24":                                 # the reference's 1D script shape: per-electrode noise list (fit_gpcsd_baseline.py:79-105)
        return fit("aud24")
    if key == "potrf":
        return potrf_bench()
    raise SystemExit("bench.py: unknown sub-result %r (one of %s)" % (key, ", ".join(SUB_RESULT_KEYS)))


_JSON_DEFAULT = lambda o: float(o) if isinstance(o, np.floating) else str(o)
SUB_RESULT_MARK = "BENCH_SUB_RESULT "


def sub_result_child(key, timeout_s=240.0):
    """Run one sub-result as a command of its own -- `python bench.py --sub-result <key>`, a fresh child process -- and return its
    dict: every workload is then measured as `--workload <key>` measures it, in the first step loop of its process.  Measured in
    the same process behind the cfg3 legs, cfg2 read 0.90-0.91 ms per step in two rehearsals of the driver command out of three
    (0.62 in the third) against 0.624-0.629 ms in six runs as a command of its own: on some boxes a loop that is not the first one
    of its process contains one stall of 5-25 ms in which no kernel of the process runs (tools/two_models_probe.py, DESIGN 6: not
    the streams, not Python's collector, not the allocator, not NUMA balancing; first loops are hit about once in thirty).
    The parent is idle on the GPU while a child runs (one child at a time), and no exec happens in a GPU-initialised process:
    the child is started with subprocess and this process goes on to print the line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--sub-result", key]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s, cwd=ROOT)
    for ln in reversed(p.stdout.splitlines()):
        if ln.startswith(SUB_RESULT_MARK):
            return json.loads(ln[len(SUB_RESULT_MARK):])
    raise RuntimeError("sub-result %s: exit code %d, no result line; stderr tail: %s" % (key, p.returncode, p.stderr[-400:]))


def sub_results(args, local_rank, backend, cpu_legs):
    """Compact cfg2 (BASELINE configs[1]), cfg5 (configs[4], this GPU's share at N=1: all 32 restarts), dense Cholesky, npx69 (the
    reference's 2D script shape) and aud24 (its 1D script shape) results.  Each is measured in a child process of its own
    (sub_result_child); GPCSD_BENCH_SUB_INPROC=1 measures them inside this process instead, CPU legs deferred to `cpu_legs`."""
    out = {}
    t0 = time.perf_counter()
    inproc = os.environ.get("GPCSD_BENCH_SUB_INPROC") == "1"
    out["measured_in"] = "this process" if inproc else "one child process per sub-result (python bench.py --sub-result <key>)"
    # Order: the latency-bound step loops first, the machine-filling legs last.  The dense Cholesky (18 ms launches at 0.4 of the
    # MFMA peak) leaves the card's clocks low for the next tenth of a second: the driver's round-5 rehearsal read npx69 at 1.13 ms
    # per step right behind it against 0.586 ms as a command of its own (and 0.57 for its control, which ran one leg later).
    for key in ("cfg2", "cfg3fit", "cfg5", "npx69", "aud24", "potrf"):
        try:                                                       # a sub-result must never take the headline down
            out[key] = sub_result_one(args, key, local_rank, backend, cpu_legs) if inproc else sub_result_child(key)
        except Exception as e:
            out[key] = {"error": repr(e)}
    out["seconds_spent_gpu_legs"] = time.perf_counter() - t0
    return out


if __name__ == "__main__":
    main()
