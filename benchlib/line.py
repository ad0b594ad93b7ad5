"""The ONE line bench.py prints: scalars only, one level deep, at most LINE_LIMIT bytes; everything else goes to DETAIL_FILE."""
import json


def sub_headlines(sub):
    """Headline scalars of the sub-results, flat, for `config` of the compact line."""
    h = {}
    g = lambda d, *ks: (g(d.get(ks[0]), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
    h["cfg2_trials_per_sec"] = g(sub, "cfg2", "value")
    h["cfg2_ms_per_step"] = g(sub, "cfg2", "ms_per_step")
    h["cfg3fit_evals_per_sec"] = g(sub, "cfg3fit", "value")
    h["cfg3fit_single_eval_ms"] = g(sub, "cfg3fit", "single_eval_ms")
    h["cfg3fit_single_eval_over_fenced_loglik"] = g(sub, "cfg3fit", "single_eval_over_fenced_loglik")
    h["cfg3fit_batch4_evals_per_sec"] = g(sub, "cfg3fit", "evals_by_lockstep_batch", "4", "evals_per_sec")
    h["cfg3fit_fit_evals_per_sec"] = g(sub, "cfg3fit", "fit", "evals_per_sec")
    h["cfg3fit_frac"] = g(sub, "cfg3fit", "roofline_frac_step_executed")
    h["cfg3fit_grad_err_vs_oracle"] = g(sub, "cfg3fit", "parity", "gradient_worst_component_rel_err_vs_oracle_closed_form")
    h["cfg3fit_cpu_evals_per_sec"] = g(sub, "cfg3fit", "cpu_baseline", "value")
    h["cfg5_evals_per_sec"] = g(sub, "cfg5", "value")
    h["cfg5_fit_evals_per_sec"] = g(sub, "cfg5", "fit", "evals_per_sec")
    for k in ("potrf", "npx69"):
        for kk, vv in (g(sub, k, "headline") or {}).items():
            h["%s_%s" % (k, kk)] = vv
    h["aud24_evals_per_sec"] = g(sub, "aud24", "value")
    h["aud24_fit_evals_per_sec"] = g(sub, "aud24", "fit", "evals_per_sec")
    h["aud24_predict_trials_per_sec"] = g(sub, "aud24", "predict_trials_per_sec")
    h["aud24_grad_err_vs_oracle_fd"] = g(sub, "aud24", "parity", "gradient_max_err_over_max_component_vs_oracle_fd")
    return h


# ------------------------------------------------------------------------------------------------------- the printed line
LINE_LIMIT = 4096          # bytes; the driver keeps an 8 KB stdout tail and parses the last line (round 4's 20 KB line was lost)
DETAIL_FILE = "bench_detail.json"

_TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "setup_steps", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "loglik", "parity_rel_err_loglik_vs_oracle", "parity_rel_err_predict_vs_oracle",
             "n", "ms", "tflops", "frac_of_fp64_mfma_peak", "evals_per_sec_one_at_a_time_per_gpu", "batched_over_sequential",
             "only_value")
_NESTED_KEYS = {
    "config": ("workload", "n_elec", "n_t", "trials_per_gpu", "total_trials", "parallelism", "restarts_total", "restarts_per_gpu",
               "lockstep_batch", "class_api_predict_trials_per_sec", "class_api_predict_host_gb_per_sec",
               "class_api_predict_cached_trials_per_sec", "fenced_loglik_ms", "fenced_predict_ms", "two_steps_in_flight_ms",
               "next_step_announced", "pair_shares_spatial_side", "unannounced_ms_per_step", "library_default_ms_per_step",
               "cfg2_trials_per_sec", "cfg2_ms_per_step", "cfg3fit_evals_per_sec", "cfg3fit_single_eval_ms",
               "cfg3fit_single_eval_over_fenced_loglik", "cfg3fit_batch4_evals_per_sec", "cfg3fit_fit_evals_per_sec", "cfg3fit_frac",
               "cfg3fit_grad_err_vs_oracle", "cfg3fit_cpu_evals_per_sec", "single_eval_ms", "single_eval_over_fenced_loglik",
               "batch4_evals_per_sec", "cfg5_evals_per_sec", "cfg5_fit_evals_per_sec",
               "potrf_ms", "potrf_frac", "potrf_trailing_update_frac", "npx69_trials_per_sec", "npx69_ms_per_step",
               "npx69_fit_evals_per_sec", "npx69_step_over_symmetric_control", "aud24_evals_per_sec", "aud24_fit_evals_per_sec",
               "aud24_predict_trials_per_sec", "aud24_grad_err_vs_oracle_fd", "fit_evals_per_sec", "fit_restarts_per_sec",
               "predict_trials_per_sec", "predict100_trials_per_sec"),
    "roofline": ("bound", "unit", "peak", "achieved", "frac", "executed_gflop_per_step", "dominant_kernel_name",
                 "dominant_kernel_frac", "dominant_kernel_avg_ms", "dominant_kernel_share", "largest_gemm_frac", "all_gemm_frac",
                 "traffic", "algorithmic_bytes_per_step", "traffic_over_algorithmic", "measured_mfma_f64_peak_tflops",
                 "reference_algorithm_frac"),
    "cpu_baseline": ("value", "unit", "kind", "cores", "blas_threads", "host_cpus", "loglik_evals_per_sec", "predict_trials_per_sec",
                     "reference_layout_loglik_evals_per_sec", "single_thread_trials_per_sec", "sample"),
    "distributed": ("ranks", "rccl_ranks", "collective_backend", "scaling_efficiency"),
}


def compact_record(full):
    """The ONE line bench.py prints: scalars only, one level deep inside config / roofline / cpu_baseline / distributed, a fixed set
    of keys, at most LINE_LIMIT bytes.  Everything else (notes, per-kernel tables, sub-result dicts) lives in DETAIL_FILE."""
    def scalar(v):
        return v is None or isinstance(v, (bool, int, float)) or (isinstance(v, str) and len(v) <= 200)
    rec = {k: full[k] for k in _TOP_KEYS if k in full and scalar(full[k])}
    for obj, keys in _NESTED_KEYS.items():
        src = full.get(obj)
        if isinstance(src, dict):
            rec[obj] = {k: src[k] for k in keys if k in src and scalar(src[k])}
        elif obj in full:
            rec[obj] = None
    rec["detail"] = DETAIL_FILE
    line = json.dumps(rec)
    if len(line) > LINE_LIMIT:
        raise AssertionError("bench.py: the result line is %d bytes (limit %d): move keys to the detail file" % (len(line), LINE_LIMIT))
    return line
