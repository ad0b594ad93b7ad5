"""CPU-only: the C-ABI library builds, loads, and exports every symbol include/gpcsd_hip.h declares;
the product path fails loudly without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "gpcsd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gpcsd_[A-Za-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from gpcsd_amd import build, _hip
    build.build(verbose=False)
    lib = _hip.load_library()
    names = _declared_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "libgpcsd_hip.so does not export %s" % n
        assert n in _hip.SIGNATURES, "ctypes prototype missing for %s" % n
    assert set(_hip.SIGNATURES) == set(names)
    assert lib.gpcsd_version() >= 100


def test_hparams_struct_layout():
    from gpcsd_amd import _hip
    import ctypes
    # double R, eps, ell_s[2]; int n_temporal, kind[8]; (pad) double ell_t[8], sigma2_t[8]; int n_sig2n; ptr; double
    assert ctypes.sizeof(_hip.HParams) == 8 * 4 + 4 * 9 + 4 + 8 * 16 + 8 + 8 + 8
    assert _hip.HParams.ell_t.offset == 72


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gpcsd_amd import _hip, forward_models
    with pytest.raises(_hip.HipUnavailable):
        forward_models.b_fwd_1d(np.ones(4), 1.0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "gpcsd_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), "%s mentions the oracle" % f


def test_host_helpers_match_golden():
    """Host-side (non-GPU) pieces of the mirrored surface against the reference's outputs."""
    from helpers import golden
    from gpcsd_amd import utility_functions as U, priors as P, predict_csd
    g = golden("ops")
    assert np.array_equal(U.mykron(g["kron_A"], g["kron_B"]), g["kron_out"])
    assert np.array_equal(U.expand_grid(np.array([1.0, 2.0, 3.5]), np.array([-1.0, 0.5])), g["expand_grid_out"])
    assert np.array_equal(U.sort_grid(g["grid_perm"]), g["sort_grid_out"])
    r1, r2 = U.reduce_grid(g["grid_perm"])
    assert np.array_equal(r1, g["reduce_grid_1"]) and np.array_equal(r2, g["reduce_grid_2"])
    assert np.array_equal(U.normalize(g["normalize_in"]), g["normalize_out"])
    ig = P.GPCSDInvGammaPrior()
    ig.set_params(1.2, 80.0)
    assert np.allclose([ig.alpha, ig.beta], g["ig_alpha_beta"], rtol=1e-15)
    hn = P.GPCSDHalfNormalPrior(0.1)
    for v, a, b in zip(g["prior_x"], g["ig_lpdf"], g["hn_lpdf"]):
        assert (ig.lpdf(v) == a) or np.isclose(ig.lpdf(v), a, rtol=1e-15)
        assert (hn.lpdf(v) == b) or np.isclose(hn.lpdf(v), b, rtol=1e-15)
    # derivative of lpdf (ours) against central differences
    for pr in (ig, hn):
        for v in (0.3, 5.0):
            fd = (pr.lpdf(v + 1e-6) - pr.lpdf(v - 1e-6)) / 2e-6
            assert np.isclose(pr.dlpdf(v), fd, rtol=1e-6)
    assert str(ig).startswith("InvGamma(") and str(hn) == "HalfNormal(0.10)"
    # the traditional CSD estimators are device operators now: without a GPU they raise (no CPU fallback)
    assert callable(predict_csd.predictcsd_trad_1d) and callable(predict_csd.predictcsd_trad_2d)


def test_model_construction_and_param_surface_cpu():
    """Constructors, param dicts, bounds and (de)serialisation work without touching the GPU."""
    import cases as C
    from gpcsd_amd.gpcsd1d import GPCSD1D
    from gpcsd_amd.gpcsd2d import GPCSD2D
    np.random.seed(3)
    x = np.linspace(0, 2300, 24)[:, None]
    t = np.arange(50.0)[:, None]
    m = GPCSD1D(np.zeros((24, 50)), x, t)
    assert m.lfp.shape == (24, 50, 1) and m.a == 0 and m.b == 2300 and m.ngl == 100
    assert set(m.spatial_cov.params["ell"].keys()) == {"value", "prior", "min", "max"}
    assert m.R["min"] == 50.0 and np.isclose(m.R["max"], 0.8 * 2300)
    assert m.sig2n["max"] == 0.5 and len(m._bounds()) == 7
    p = m.extract_model_params()
    assert set(p) == {"R", "sig2n", "spatial_ell", "temporal_ell_list", "temporal_sigma2_list"}
    p["R"] = 123.0
    m.restore_model_params(p)
    assert m.R["value"] == 123.0
    assert "GPCSD1D object" in str(m)
    m2 = GPCSD2D(np.zeros((48, 50, 2)), C.grid_xy(4, 12, 0, 48, 0, 440), t)
    assert m2.eps == 5 * 16.0 and m2.ngl1 == 20 and m2.ngl2 == 60
    assert set(m2.extract_model_params()) == {"R", "eps", "sig2n", "spatial_ell1", "spatial_ell2", "temporal_ell_list",
                                              "temporal_sigma2_list"}
    assert len(m2._bounds()) == 8 and m2._bounds()[6][0] == -np.inf      # Matern sigma2 lower bound 0 -> log(0)
    from gpcsd_amd.priors import GPCSDHalfNormalPrior
    m3 = GPCSD1D(np.zeros((24, 50, 2)), x, t, sig2n_prior=[GPCSDHalfNormalPrior(0.1) for _ in range(24)])
    assert m3.sig2n["value"].shape == (24,) and len(m3._bounds()) == 6 + 24
    assert m2.spatial_cov.gl_x_grid.shape == (1200, 2) and m2.spatial_cov.delta_w.shape == (48, 1200)


def test_shard_block_matches_the_python_partition():
    """gpcsd_shard_block (the C-side partition for binders without Python) == TrialSharding.block; no GPU needed."""
    import ctypes
    from gpcsd_amd import _hip
    from gpcsd_amd.dist import TrialSharding
    lib = _hip.load_library()
    for n, w in [(400, 8), (5, 2), (3, 4), (50, 1), (0, 3)]:
        for r in range(w):
            a, c = ctypes.c_int(), ctypes.c_int()
            assert lib.gpcsd_shard_block(n, r, w, ctypes.byref(a), ctypes.byref(c)) == 0
            lo, hi = TrialSharding.block(n, r, w)
            assert (a.value, a.value + c.value) == (lo, hi)
    a, c = ctypes.c_int(), ctypes.c_int()
    assert lib.gpcsd_shard_block(10, 3, 3, ctypes.byref(a), ctypes.byref(c)) == -3
    out = ctypes.c_double()
    assert lib.gpcsd_combine_loglik(4, 2.0, 3.0, ctypes.byref(out)) == 0 and out.value == -0.5 * 4 * 2.0 - 0.5 * 3.0


def test_cpulist_parser_and_numa_binding_is_a_noop_without_a_gpu():
    """bind_host_to_device_numa: the kernel's cpulist format, and no change of affinity when the device cannot be asked."""
    import os
    from gpcsd_amd import _hip
    assert _hip._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert _hip._parse_cpulist("") == set() and _hip._parse_cpulist("5") == {5}
    before = os.sched_getaffinity(0)
    import torch
    if not torch.cuda.is_available():
        assert _hip.bind_host_to_device_numa(0) is None
        assert os.sched_getaffinity(0) == before


def test_bench_gpus_flag_is_honoured_without_a_gpu():
    """`bench.py --gpus N` must mean N ranks: under a launcher whose WORLD_SIZE differs it refuses; without a launcher it starts
    the ranks itself -- here, with no GPU, that stops at the device count with exit code 2 (nothing is launched)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GPCSD_DEVICE", "GPCSD_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4"], cwd=root, env=dict(env, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 4" in r.stderr and "WORLD_SIZE=2" in r.stderr
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 2 and "one rank per GPU" in r.stderr and not r.stdout.strip()
