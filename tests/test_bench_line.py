"""CPU: the shape of bench.py's final stdout line (BENCH_r04.json had `parsed: null` because the line had grown to 25 KB).

`bench.compact_line` / `bench.emit` are pure functions of the full result object; here they get a canned object of the size the real run
produces (every per-mode object, a 20-entry kernels[] list, calibration, closed loop) and must return strict JSON under 4 KB with the
contract's keys, roofline and cpu_baseline."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def canned(n_kernels=20):
    import bench

    kern = [{"kernel": f"conv3x3_halo_kernel<{i}>", "achieved": 324.8643619855192 - i, "frac": 0.3898372343826231, "traffic": 163371228.44444445 + i,
             "launches_per_step": 27.0, "avg_launch_ms": 0.12123506323054985, "flop_per_launch_avg": 39384951466.666664, "ms_per_step": 3.27} for i in range(n_kernels)]
    roof = {"kernel": "conv3x3_halo_kernel", "bound": "mfma", "achieved": 324.8643619855192, "peak": 2500.0 / 3, "peak_basis": bench.PEAK_BASIS["f16x3"], "unit": "TFLOP/s",
            "frac": 0.3898372343826231, "frac_of_fp16_peak": 0.12994574479420768, "traffic": 163371228.44444445, "hbm_gbps": 1347.5575802172448, "hbm_frac": 0.1684446975271556,
            "mfma_util_pmc": 0.4213908014990542, "launches_per_step": 27.0, "avg_launch_ms": 0.12123506323054985, "flop_per_launch_avg": 39384951466.666664,
            "share_of_forward": 0.505, "timing": "HIP events on the launch stream, 10 single-stream forwards of this run", "end_to_end_frac": 0.3717122391878488,
            "conv_family": {"achieved": 284.0, "frac": 0.34}, "kernels": kern, "class_ms_per_step": {"conv": 6.4, "pool": 0.05, "head": 0.018},
            "provenance": {"traffic, hbm_gbps, hbm_frac": {"file": "profiles/r05_conv_traffic_f16x3.json", "src_sha": "c90f31db38d64436", "note": "x" * 80},
                           "mfma_util_pmc": {"file": "profiles/r05_pmc_mfma_util_f16x3.json", "src_sha": "c90f31db38d64436", "note": "x" * 80}}}
    win = {"n": 10, "steps_each": 20, "median_ms": 117.48615249962313, "min_ms": 117.07613799080718, "max_ms": 117.86104000930209, "first_ms": 117.9, "value_best": 1.0, "value_worst": 1.0}
    acc = {"frames": 128, "index_match": 128, "index_match_rate": 1.0, "iou_matched": {"min": 0.999997552638024, "p01": 0.99999, "p50": 1.0}, "iou_all": {"min": float("nan")}}
    mode = {"dtype": "f16x3", "value": 10894.900996984356, "unit": "frames/s", "ms_per_step": 5.8743076249811566, "windows": win, "roofline": roof, "note": "n" * 400}
    d = {"metric": "frames/sec YOLOv8s+ResMLP sim loop @640x640", "value": 10894.900996984356, "unit": "frames/s", "n_gpus": 1, "steps": 20, "warmup": 5,
         "ms_per_step": 5.8743076249811566, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16x3", "data": "synthetic",
         "config": {"workload": "BASELINE configs[2]: full sim loop, YOLOv8s (nc=1, seeded synthetic weights) + ResMLP(imaging-100ms_pred-40ms_moving-50ms, reference weights)",
                    "frame": "640x640 uint8 gray, resident in HBM", "batch_per_gpu": 64, "lanes_per_gpu": 2, "global_batch": 64, "timing_ms": [100, 40, 50], "conf": 0.1,
                    "headline_mode": "f16x3", "headline_reason": "r" * 200, "parallelism": "single GPU"},
         "windows": win, "roofline": roof, "headline_exactness_verified": True,
         "headline_check": {"against": "a" * 150, "frames": 128, "index_mismatches": 0, "box_abs_diff_max_px": 0.000823974609375, "verified": True},
         "end_to_end": {"frac": 0.37}, "closed_loop": {"what": "w" * 300, "f16x3_eager": {"frames_per_s": 5893.0}},
         "latency": {"what": "w" * 300, "rows": [{"mode": m, "shape": s, "device_ms": 0.4, "host_ms": 0.5} for m in ("f16x3", "fp32") for s in ("b1_384", "b15_384", "b1_640")]},
         "latency_b1_384_f16x3_ms": 0.41234567, "latency_b15_384_f16x3_ms": 0.71234567, "latency_b1_384_fp32_ms": 0.61234567, "latency_b15_384_fp32_ms": 2.21234567,
         "latency_b1_640_f16x3_ms": 0.64234567, "latency_b1_640_fp32_ms": 0.88234567, "closed_loop_f16x3_frames_per_s": 8309.536123,
         "closed_loop_f16x3_deferred_log_frames_per_s": 11488.971456,
         "cpu_baseline": {"value": 15.374184052472774, "unit": "frames/s", "cores": 32, "kind": "port", "by_batch": {"64": {}}, "sample": "s" * 300},
         "parity": {"checker": "c" * 100, "frames": "f", "headline_gate": "g", "floors_asserted_in_tests": "t" * 300, "fp16": acc, "fp32": acc, "f16x3": acc, "hybrid": acc},
         "hybrid_calibration": {"mismatch_margins_sorted_desc": [0.009] * 40}}
    for dt in ("f16x3", "fp16", "hybrid", "fp32"):
        d[f"value_{dt}"] = 10894.900996984356
        d[f"ms_per_step_{dt}"] = 5.87
        d["fp16_throughput" if dt == "fp16" else dt] = mode
    return d


def test_final_line_is_short_strict_json_with_the_contract_keys(tmp_path, monkeypatch):
    import bench

    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.mkdir(tmp_path / "gpurun_out")
    d = canned()
    assert len(json.dumps(d)) > 20000  # the full object is what broke the driver's parser
    s = bench.emit(bench._finite(d))
    assert len(s) < bench.LINE_LIMIT == 4096 and "\n" not in s and "NaN" not in s
    line = json.loads(s)
    for k in CONTRACT:
        assert k in line, k
    assert line["dtype"] == "f16x3" and line["value"] == pytest.approx(d["value"], rel=1e-6) and line["ms_per_step"] == pytest.approx(d["ms_per_step"], rel=1e-6)
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "peak_basis", "unit", "frac", "frac_of_fp16_peak", "traffic", "avg_launch_ms", "end_to_end_frac"):
        assert k in r, k
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3) and "/ 3" in r["peak_basis"]
    assert r["provenance"] == ["profiles/r05_conv_traffic_f16x3.json@c90f31db38d64436", "profiles/r05_pmc_mfma_util_f16x3.json@c90f31db38d64436"]
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 32 and c["value"] > 0 and c["unit"] == "frames/s" and c["sample"]
    assert line["parity"]["f16x3"] == {"index_match_rate": 1.0, "iou_matched_min": pytest.approx(0.9999976, abs=1e-6)}
    assert all(f"value_{dt}" in line for dt in ("f16x3", "fp16", "hybrid", "fp32"))
    assert line["latency_b1_384_f16x3_ms"] == pytest.approx(0.41235, rel=1e-4)
    # BASELINE configs[1] (YoloController, 640 x 640, batch 1) travels in the line the driver parses
    assert line["latency_b1_640_f16x3_ms"] == pytest.approx(0.64235, rel=1e-4) and line["latency_b1_640_fp32_ms"] == pytest.approx(0.88235, rel=1e-4)
    assert line["closed_loop_f16x3_frames_per_s"] == pytest.approx(8309.5, rel=1e-4) and line["closed_loop_f16x3_deferred_log_frames_per_s"] == pytest.approx(11489.0, rel=1e-4)
    assert line["config"]["workload"].startswith("BASELINE configs[2]") and "model" not in line["config"]
    for k in ("closed_loop", "latency", "hybrid_calibration", "f16x3", "fp32", "hybrid", "fp16_throughput"):
        assert k not in line, k  # detail only
    # the full object travels next to the script and under gpurun_out/ (which a GPU box carries back)
    for where in (tmp_path, tmp_path / "gpurun_out"):
        full = json.load(open(where / line["detail"]))
        assert full["closed_loop"] and len(full["roofline"]["kernels"]) == 20 and full["parity"]["fp16"]["iou_all"]["min"] is None  # NaN -> null


def test_line_stays_under_the_limit_when_optional_parts_grow(tmp_path, monkeypatch):
    import bench

    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    d = canned(n_kernels=200)
    d["config"]["workload"] = "w" * 1500
    d["dist"] = {"world_size": 8, "backend": "nccl", "track_rows": 10240, "track_checksum": [1.0, 2.0], "checksum_equal_on_all_ranks": True}
    s = bench.emit(bench._finite(d))
    assert len(s) < 4096
    line = json.loads(s)
    assert all(k in line for k in CONTRACT) and line["roofline"] and line["cpu_baseline"]
