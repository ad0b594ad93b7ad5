"""bench.py's printed line: ONE compact JSON line of scalars, at most 4 KB, with everything
else in a side file -- round 4's 20 KB line outgrew the driver's 8 KB stdout tail and was recorded as `parsed: null`."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

REQUIRED = {
    None: ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
           "data", "parity_rel_err_loglik_vs_oracle", "parity_rel_err_predict_vs_oracle", "detail"),
    "config": ("workload", "n_elec", "n_t", "trials_per_gpu", "total_trials", "parallelism", "cfg2_trials_per_sec", "cfg2_ms_per_step",
               "cfg5_evals_per_sec", "cfg5_fit_evals_per_sec", "potrf_ms", "potrf_frac", "npx69_ms_per_step",
               "class_api_predict_trials_per_sec"),
    "roofline": ("bound", "unit", "peak", "achieved", "frac", "executed_gflop_per_step", "dominant_kernel_name", "dominant_kernel_frac",
                 "dominant_kernel_avg_ms", "dominant_kernel_share", "largest_gemm_frac", "all_gemm_frac", "traffic",
                 "algorithmic_bytes_per_step", "traffic_over_algorithmic", "measured_mfma_f64_peak_tflops"),
    "cpu_baseline": ("value", "unit", "kind", "cores", "blas_threads", "loglik_evals_per_sec", "predict_trials_per_sec",
                     "reference_layout_loglik_evals_per_sec", "single_thread_trials_per_sec"),
}


def _canned():
    """Round 4's full result dict (20 KB as a line): the record the driver could not parse."""
    with open(os.path.join(ROOT, "profiles", "r04_bench_cfg3.json")) as fh:
        return json.load(fh)


def test_compact_line_of_a_full_result_fits_and_keeps_the_contract_keys():
    full = _canned()
    assert len(json.dumps(full)) > 8192                        # the canned dict really is the oversized one
    full["cpu_baseline"]["sample"] = "oracle loglik x20 + predict(csd) x3, the bench's own 50 trials, medians, 16 BLAS threads"
    line = bench.compact_record(full)
    assert "\n" not in line and len(line) <= bench.LINE_LIMIT == 4096
    rec = json.loads(line)
    for obj, keys in REQUIRED.items():
        d = rec if obj is None else rec[obj]
        missing = [k for k in keys if k not in d]
        assert not missing, (obj, missing)
    assert rec["cpu_baseline"]["sample"] and rec["cpu_baseline"]["kind"] == "port" and rec["roofline"]["bound"] == "mfma"
    # scalars only, one level deep: nothing nested below config / roofline / cpu_baseline / distributed, no long prose
    for k, v in rec.items():
        if isinstance(v, dict):
            assert k in ("config", "roofline", "cpu_baseline", "distributed")
            assert all(not isinstance(x, (dict, list)) for x in v.values()), k
        else:
            assert not isinstance(v, list)
    assert all(len(v) <= 200 for d in [rec] + [x for x in rec.values() if isinstance(x, dict)] for v in d.values() if isinstance(v, str))
    assert "sub_results" not in rec and "pipelining" not in rec
    assert rec["value"] == full["value"] and rec["ms_per_step"] == full["ms_per_step"]


def test_compact_line_of_a_multi_rank_result_keeps_the_distributed_scalars():
    full = _canned()
    full["n_gpus"] = 8
    full.pop("cpu_baseline")
    full["distributed"] = {"collective_backend": "nccl", "rccl_ranks": 8, "ranks": 8, "per_rank_ms_per_step": [1.0] * 8,
                           "per_rank_ms_per_step_without_collectives": [0.99] * 8, "scaling_efficiency": 0.97,
                           "scaling_efficiency_against": "x" * 150}
    rec = json.loads(bench.compact_record(full))
    assert rec["distributed"] == {"ranks": 8, "rccl_ranks": 8, "collective_backend": "nccl", "scaling_efficiency": 0.97}
    assert "cpu_baseline" not in rec


def test_a_line_that_would_outgrow_the_limit_is_refused_not_printed():
    full = _canned()
    full["config"] = dict(full["config"], **{k: "y" * 200 for k in bench._NESTED_KEYS["config"]})
    full["roofline"] = dict(full["roofline"], **{k: "y" * 200 for k in bench._NESTED_KEYS["roofline"]})
    with pytest.raises(AssertionError, match="limit 4096"):
        bench.compact_record(full)


def test_emit_writes_the_detail_file_and_prints_the_compact_line_last(tmp_path, monkeypatch, capsys):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full = _canned()
    bench.emit(full)
    out = capsys.readouterr().out
    last = out.rstrip().splitlines()[-1]
    assert last.startswith("{") and len(last) <= 4096 and json.loads(last)["detail"] == bench.DETAIL_FILE
    with open(tmp_path / bench.DETAIL_FILE) as fh:
        assert json.load(fh)["sub_results"]["cfg2"]["value"] == full["sub_results"]["cfg2"]["value"]


def test_sub_result_headlines_are_flattened_into_config():
    sub = _canned()["sub_results"]
    h = bench.sub_headlines(sub)
    assert h["cfg2_trials_per_sec"] == sub["cfg2"]["value"] and h["cfg5_fit_evals_per_sec"] == sub["cfg5"]["fit"]["evals_per_sec"]
    assert h["potrf_ms"] == sub["potrf"]["ms"] and h["npx69_ms_per_step"] == sub["npx69"]["ms_per_step"]
    assert bench.sub_headlines({"cfg2": {"error": "boom"}})["cfg2_trials_per_sec"] is None      # a failed leg leaves nulls, no raise


def test_sub_results_run_as_child_commands_and_a_failed_one_leaves_nulls(monkeypatch):
    """The default line measures every sub-workload in a child process of its own (`bench.py --sub-result <key>`): the child's
    dict comes back through one marked stdout line; a child that dies leaves an error entry, the headline survives."""
    import subprocess
    import types
    calls = []

    def fake_run(cmd, **kw):
        calls.append(cmd)
        key = cmd[cmd.index("--sub-result") + 1]
        assert cmd[0] == sys.executable and os.path.basename(cmd[1]) == "bench.py" and kw.get("timeout")
        if key == "cfg5":                                          # a child that crashed before printing its result
            return types.SimpleNamespace(stdout="noise\n", stderr="boom", returncode=1)
        d = {"value": 123.0, "ms_per_step": 0.5, "headline": {"ms": 18.0, "frac": 0.4}}
        return types.SimpleNamespace(stdout="import noise\n" + bench.SUB_RESULT_MARK + json.dumps(d) + "\n", stderr="", returncode=0)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("GPCSD_BENCH_SUB_INPROC", raising=False)
    sub = bench.sub_results(types.SimpleNamespace(), 0, "nccl", [])
    assert [c[c.index("--sub-result") + 1] for c in calls] == ["cfg2", "cfg3fit", "cfg5", "npx69", "aud24", "potrf"]
    assert sub["cfg2"]["value"] == 123.0 and "error" in sub["cfg5"] and "child process" in sub["measured_in"]
    h = bench.sub_headlines(sub)
    assert h["cfg2_ms_per_step"] == 0.5 and h["cfg5_evals_per_sec"] is None and h["potrf_frac"] == 0.4
    assert set(bench.SUB_RESULT_KEYS) == {"cfg2", "cfg3fit", "cfg5", "npx69", "aud24", "potrf"}
