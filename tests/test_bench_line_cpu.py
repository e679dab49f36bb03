"""bench.py's final stdout line must stay parsable by the driver (its capture keeps 8 KB): round 3's single 20.5 KB line was
cut off and the round went unmeasured.  `bench.compact_line` is pure, so the line is built here from a canned full result --
every mode, every parity block, both secondary configs, the N > 1 blocks, full-precision floats -- and measured."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _parity(total):
    return {"idx_total": total, "idx_mismatches": 0, "idx_agree": 1.0, "mismatch_gap_max": 0.0, "slice_elems": 4096,
            "z_rel_max": 1.2917650205980314e-05, "z_rel_l2": 1.0917650205980314e-05, "z_sum_rel": 4.0917650205980314e-07,
            "recon_rel_max": 2.060955206217151e-05, "recon_rel_l2": 2.0704237236324295e-05, "recon_sum_rel": 5.971921694517551e-07,
            "vq_loss_rel": 1.4108397925106133e-07, "recon_error_rel": 1.1918131494413317e-07, "grad_rel_max": 0.0423888689822478,
            "grad_rel_l2_median": 0.0004983737811415364, "grad_rel_l2_max": 0.0054983737811415364, "grad_sum_rel_max": 0.00032661314412749505,
            "encoder_grad_rel_max": 0.017317037682857164, "golden": "tests/golden/g3_speech_b64.npz"}


def _mode(mode, value):
    fams = {f: {"launches": 180, "ms_per_step": 1.2345678901234567, "tflops": 534.1234567890123} for f in bench.CONV_FAMILIES[:6]}
    return {"value": value, "unit": "spectrograms/s", "ms_per_step": 64e3 / value, "steps": 20, "dtype": mode, "launch": "hipGraph replay",
            "model_tflops": value * 99.95 / 1e3, "final_loss": 1.2345678901234567, "allreduce_calls_per_step": 0,
            "step_peak_tflops": bench.step_peak_tflops(mode), "step_frac_of_peak": 0.4071234567890123,
            "step_frac_of_dense_16bit_peak": 0.3071234567890123, "skipped_steps": 0, "fp16_range_flag": 0,
            "roofline": {"bound": "mfma", "scope": "kernel-only: the conv family with the largest share of the step",
                         "kernel": "conv1d_wgrad_bf16_v2_kernel", "achieved": 539.1234567890123, "peak": 1250.0,
                         "peak_is": "2500/2: one fp16 + one equal-length block-scaled fp8 MFMA per product TFLOP/s", "unit": "TFLOP/s",
                         "frac": 0.4315123456789012, "frac_of_dense_16bit_peak": 0.2157123456789012, "traffic": 481912924.5538461,
                         "traffic_source": "NOT measured in this run: profiles/traffic.json (rocprofv3 --pmc passes, tools/profile_bench.sh)",
                         "launches": 180, "avg_launch_ms": 0.2231234567890123, "algorithmic_gflop_per_launch": 119.03123456789012},
            "kernel_families": fams, "other_kernels_ms_per_step": 0.5012345678901234, "other_kernels_share": 0.0598123456789012,
            "what": bench.MODE_TEXT[mode], "parity_b2": _parity(1000)}


def canned_full():
    head = _mode("x3mx_hb", 7640.123456789012)
    full = {"metric": "spectrograms/sec (train step), speech VQ-VAE default config", "value": head["value"], "unit": "spectrograms/s",
            "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "x3mx_hb", "data": "synthetic",
            "config": {"workload": "speech VQ-VAE train step (fwd+bwd+Adam), ctor [201, 1024, 128, 3, 1024, 0.25, 1024], B=64 per GPU x "
                                   "(201,500), jitter on; " + bench.MODE_TEXT["x3mx_hb"],
                       "global_batch": 512, "parallelism": "dp8", "algorithmic_gflop_per_spectrogram": 99.94931200000001}}
    for k in ("model_tflops", "step_peak_tflops", "step_frac_of_peak", "step_frac_of_dense_16bit_peak", "final_loss", "launch",
              "allreduce_calls_per_step", "roofline", "kernel_families", "other_kernels_ms_per_step", "other_kernels_share",
              "fp16_range_flag", "skipped_steps"):
        full[k] = head[k]
    full["modes"] = {m: _mode(m, 1000.0 + 1234.56789 * i) for i, m in enumerate(bench.ALL_MODES)}
    full["parity"] = {"b2": _parity(1000), "b16": _parity(8000), "b64": _parity(32000), "rir_b2": _parity(402), "rir_b32": _parity(6432),
                      "echoed_b32": {"recon_rel_max": 2.3e-5, "recon_error_rel": 1e-7}}
    full["script_loop_mode"] = {"value": 7187.123456789, "unit": "spectrograms/s", "ms_per_step": 8.9, "steps": 10, "dtype": "x3mx_hb",
                                "launch": "eager, module API + torch.optim.Adam (train_speech.py:62-74,88-91)", "vs_trainer_eager": 0.9791234567}
    full["trainer_eager_mode"] = {"value": 7341.1, "unit": "spectrograms/s", "ms_per_step": 8.7, "steps": 10, "dtype": "x3mx_hb", "launch": "eager"}
    full["vq_stress"] = {"workload": "alvq_vq_argmin_f32 ...", "ms": 6.141234567, "tflops": 87.41234567, "peak": 157.3, "frac": 0.5551234567,
                         "rows_per_s": 4.1e7, "idx_bit_exact_on_sample": True, "sample_rows": 257, "algorithmic_mbytes": 268.3}
    for key in ("rir_config", "echoed_config"):
        full[key] = {"workload": "...", "value": 17674.123456, "unit": "spectrograms/s", "ms_per_step": 1.8112345678, "steps": 20,
                     "launch": "hipGraph replay", "model_tflops": 187.123456, "algorithmic_gflop_per_spectrogram": 10.59,
                     "throughput_mode": {"dtype": "bf16", "value": 23405.1, "ms_per_step": 1.37, "model_tflops": 247.9}}
    full["grad_exchange"] = {"default": {"allreduce_calls_per_step": 1, "value": 61234.123456789, "ms_per_step": 8.361234567},
                             "two_spans": {"allreduce_calls_per_step": 2, "value": 61834.123456789, "ms_per_step": 8.281234567}}
    full["rccl"] = {"world": 8, "backend": "nccl (RCCL)", "nranks_seen": 8, "allreduce_bytes": 67348224, "allreduce_ms": 0.4123456789,
                    "allreduce_ms_min": 0.4023456789, "bus_GBps": 285.81234567, "ranks_bit_identical": True, "flat_checksum": -123456789012345,
                    "per_rank_ms_per_step": {"min": 8.301234567, "max": 8.361234567}}
    full["cpu_baseline"] = {"value": 25.212345678, "unit": "spectrograms/s", "cores": 16, "kind": "port",
                            "sample": "60 train steps of the CPU oracle (the reference's ATen op sequence), speech ctor, B=4 x (201,500) "
                                      "fp32, jitter on, Adam; median 0.159 s/step"}
    full["gpu_over_cpu"] = full["value"] / full["cpu_baseline"]["value"]
    full["detail_file"] = "bench_detail.json"
    return full


def test_compact_line_fits_the_drivers_capture_and_keeps_the_contract():
    full = canned_full()
    line = bench.compact_line(full)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < bench.LINE_LIMIT <= 6000, len(text)
    assert len(json.dumps(line)) < 6500                                     # default separators too
    assert len(json.dumps(full)) > 3 * len(text)                            # the detail really is elsewhere
    back = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in back, key
    assert set(back["config"]) >= {"workload", "global_batch", "parallelism"} and "model" not in back["config"]
    assert set(back["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert abs(back["roofline"]["frac"] - back["roofline"]["achieved"] / back["roofline"]["peak"]) < 1e-3
    assert set(back["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert back["parity_headline"]["speech_b64_the_timed_workload"] == "0/32000" and back["parity_headline"]["rir_b32"] == "0/6432"
    assert set(back["modes"]) == set(bench.ALL_MODES) and back["rccl"]["ranks_bit_identical"] is True
    assert abs(back["value"] - full["value"]) / full["value"] < 1e-5


def test_structural_peaks():
    """2500 * flops / sum(flops * units): f16mx_hb 1875 (forward third at two units, backward at one), x3mx_hb below it."""
    assert abs(bench.step_peak_tflops("f16mx_hb") - 1875.0) < 12.0
    assert abs(bench.step_peak_tflops("bf16x3_hb") - 1500.0) < 10.0
    assert bench.step_peak_tflops("bf16x3_hb") < bench.step_peak_tflops("x3mx_hb") < bench.step_peak_tflops("f16mx_hb")
    assert bench.step_peak_tflops("bf16") == 2500.0 and bench.step_peak_tflops("f32") == 157.3
    assert abs(bench.algorithmic_gflop_per_spectrogram(bench.SPEECH_CFG, 500) - 99.95) < 0.01
    assert abs(bench.algorithmic_gflop_per_spectrogram(bench.RIR_CFG, 201, 1) - 10.59) < 0.01


def test_committed_detail_files_compact_to_a_parsable_line():
    """Every bench_detail the repo keeps under profiles/ (real runs) must compact to a line inside the limit."""
    prof = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(prof)):
        if name.startswith("r04") and name.endswith("bench_detail.json"):
            full = json.load(open(os.path.join(prof, name)))
            assert len(json.dumps(bench.compact_line(full), separators=(",", ":"))) < bench.LINE_LIMIT, name
