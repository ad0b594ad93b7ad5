"""Build libgpcsd_hip.so (gfx950) in-tree with hipcc.  `python -m gpcsd_amd.build [--force]`.

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU box with the
gpurun snapshot.  Objects are cached under gpcsd_amd/csrc/_build and rebuilt when a source or header is newer.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(CSRC, os.environ.get("GPCSD_BUILD_OBJDIR", "_build"))
# developer knobs for A/B experiments (tools/ab_bench.py, tools/sytrd_time.py): another output name and extra compiler flags
LIB = os.environ.get("GPCSD_BUILD_LIB") or os.path.join(HERE, "libgpcsd_hip.so")
SOURCES = ["capi.hip", "gemm_f64.hip", "gram.hip", "eigh.hip", "eigh_dc.hip", "stedc.hip", "wy.hip", "grad.hip", "chol.hip"]
ARCH = "gfx950"
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs; the default AGPR form made hipcc copy all of them AGPR<->VGPR
# around every K tile of the GEMM main loop (64 v_accvgpr moves + s_nop per 16 MFMAs)
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-fno-fast-math", "-Wall", "-Wno-unused-function",
         "-mllvm", "-amdgpu-mfma-vgpr-form=1"]
# the elementwise Gram builders keep the reference's operation order exactly (no FMA contraction)
EXTRA = {"gram.hip": ["-ffp-contract=off"]}


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm toolchain to build libgpcsd_hip.so)")


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inl"))]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "gpcsd_hip.h"))
    hdrs.append(os.path.abspath(__file__))
    return max(os.path.getmtime(h) for h in hdrs)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_time = _deps()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + EXTRA.get(os.path.basename(s), []) + os.environ.get("GPCSD_CXXFLAGS", "").split() + ["-c", s, "-o", o]
        if verbose:
            print("[gpcsd_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (s, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    if jobs or force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        if verbose:
            print("[gpcsd_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
