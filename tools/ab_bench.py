"""A/B timing of two builds of libgpcsd_hip.so on the SAME box, interleaved (run-to-run and box-to-box variation of the
bench step is ~1.5 %, more than most single kernel changes).  python tools/ab_bench.py base.so new.so [rounds] [steps]
A side may carry environment settings after '@':  new.so@GPCSD_NO_FOLD_GEMM=1  (the same library under two switches).
AB_WORKLOAD=cfg2|npx69|... times another step workload than cfg3."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [sys.argv[1], sys.argv[2]]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
steps = sys.argv[4] if len(sys.argv) > 4 else "40"
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        path, _, extra = l.partition("@")
        env = dict(os.environ, GPCSD_LIB_PATH=os.path.abspath(path))
        for kv in filter(None, extra.split(",")):
            env[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--only-value", "--steps", steps, "--warmup", "5",
                              "--workload", os.environ.get("AB_WORKLOAD", "cfg3")],
                             env=env, capture_output=True, text=True)
        if out.returncode != 0:
            print(out.stderr[-2000:])
            sys.exit(1)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        res[l].append(d["ms_per_step"])
        print(r, os.path.basename(l), "%.4f ms/step" % d["ms_per_step"], flush=True)
for l in libs:
    v = sorted(res[l])
    print(os.path.basename(l), "min %.4f median %.4f" % (v[0], v[len(v) // 2]))
