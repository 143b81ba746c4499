"""The harness (wtracker_amd/sim.py), CsvController and the oracle MLP controller reproduce the logs
the REAL reference wrote for the same seeded input (tests/golden/*.csv, sim_moves.json).  CPU only."""
import csv
import json
import os

import numpy as np
import pytest

from oracle import resmlp_oracle
from oracle.controllers_oracle import OracleMLPController
from wtracker_amd.controllers import CsvController, OptimalController, PolyfitConfig, PolyfitController
from wtracker_amd.resmlp import make_training_pairs
from wtracker_amd.sim import ExperimentConfig, TimingConfig, TrackLogger, box_center, discretize, xyxy_to_xywh, yolo_to_xywh
from harness.sim_harness import Simulator

EXP0 = dict(name="exp0", num_frames=200, frames_per_sec=60, orig_resolution=(1600, 1400), px_per_mm=90, init_position=(1300, 1200))


def read_log(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def assert_rows_equal(rows, golden):
    assert len(rows) == len(golden)
    for r, g in zip(rows, golden):
        for k, gv in g.items():
            v = r[k]
            if k == "phase":
                assert v == gv
            elif gv in ("", "nan"):
                assert not np.isfinite(float(v))
            else:
                assert float(v) == float(gv), (k, r["frame"], v, gv)


def test_timing_arithmetic(golden_dir):
    for t in json.load(open(os.path.join(golden_dir, "timing.json"))):
        ec = ExperimentConfig("t", 10, t["fps"], (1600, 1400), t["px_per_mm"], (10, 10))
        tc = TimingConfig(ec, t["imaging_ms"], t["pred_ms"], t["moving_ms"], (4, 4), (0.32, 0.32))
        assert (tc.imaging_frame_num, tc.pred_frame_num, tc.moving_frame_num, tc.cycle_frame_num) == (
            t["imaging_frame_num"], t["pred_frame_num"], t["moving_frame_num"], t["cycle_frame_num"])
        assert list(tc.camera_size_px) == t["camera_size_px"] and list(tc.micro_size_px) == t["micro_size_px"]
        assert tc.ms_per_frame == t["ms_per_frame"]


def run(ctrl_factory, timing=(100, 40, 50)):
    ec = ExperimentConfig(**EXP0)
    tc = TimingConfig(ec, *timing, (4, 4), (0.32, 0.32))
    ctrl = ctrl_factory(tc)
    moves = []
    inner = ctrl.provide_movement_vector

    def wrapped(sim):
        dx, dy = inner(sim)
        moves.append([int(sim.frame_number), int(dx), int(dy)])
        return dx, dy

    ctrl.provide_movement_vector = wrapped
    log = TrackLogger(ctrl)
    Simulator(tc, ec, log).run()
    return log.rows, moves


def test_csv_controller_loop_matches_reference_log(golden_dir):
    init = os.path.join(golden_dir, "sim_init_bboxes.csv")
    rows, moves = run(lambda tc: CsvController(tc, init))
    assert_rows_equal(rows, read_log(os.path.join(golden_dir, "sim_csv_bboxes.csv")))
    assert moves == json.load(open(os.path.join(golden_dir, "sim_moves.json")))["sim_csv_bboxes.csv"]
    assert len(rows) == 198  # the last cycle is never logged


@pytest.mark.parametrize("tag,timing,name", [("100ms", (100, 40, 50), "sim_mlp_bboxes.csv"), ("200ms", (200, 40, 50), "sim_mlp200_bboxes.csv")])
def test_oracle_mlp_controller_loop_matches_reference_log(golden_dir, tag, timing, name):
    init = os.path.join(golden_dir, "sim_init_bboxes.csv")
    st = resmlp_oracle.load_state(os.path.join(golden_dir, f"resmlp_{tag}.npz"))
    rows, moves = run(lambda tc: OracleMLPController(tc, init, st), timing)
    assert moves == json.load(open(os.path.join(golden_dir, "sim_moves.json")))[name]
    assert_rows_equal(rows, read_log(os.path.join(golden_dir, name)))


def test_optimal_and_polyfit_controllers_match_reference_moves(golden_dir):
    """the comparison baselines around the ResMLP — per-cycle (dx, dy) equal to what the
    real reference's OptimalController / PolyfitController returned on the same track."""
    init = os.path.join(golden_dir, "sim_init_bboxes.csv")
    gold = json.load(open(os.path.join(golden_dir, "sim_moves.json")))
    _, moves = run(lambda tc: OptimalController(tc, init))
    assert moves == gold["sim_optimal"]
    cfg = PolyfitConfig(degree=2, sample_times=[-9, -6, -3, 0, 2, 4], weights=[1, 1, 2, 3, 4, 5])
    _, moves = run(lambda tc: PolyfitController(tc, cfg, init))
    assert moves == gold["sim_polyfit"]
    assert PolyfitConfig(1, [0, 1, 2]).weights == [1.0, 1.0, 1.0]
    with pytest.raises(AssertionError):
        PolyfitConfig(1, [0, 1, 2], [1.0])


def test_training_pairs_match_reference_dataset(golden_dir):
    """NumpyDataset.create_from_config on the golden track (neural/dataset.py:42-96), bit for bit."""
    g = np.load(os.path.join(golden_dir, "dataset_100ms.npz"))
    m = np.load(os.path.join(golden_dir, "resmlp_100ms.npz"))
    X, y = make_training_pairs(os.path.join(golden_dir, "sim_init_bboxes.csv"), m["input_frames"].tolist(), m["pred_frames"].tolist())
    assert X.dtype == np.float32 and y.dtype == np.float32
    assert X.shape == g["X"].shape and y.shape == g["y"].shape
    assert np.array_equal(X, g["X"]) and np.array_equal(y, g["y"])
    X0, y0 = make_training_pairs(os.path.join(golden_dir, "sim_init_bboxes.csv"), [-300, 0], [400])
    assert X0.shape == (0, 8) and y0.shape == (0, 2)


def test_csv_predict_edge_cases(golden_dir):
    ec = ExperimentConfig(**EXP0)
    tc = TimingConfig(ec, 100, 40, 50, (4, 4), (0.32, 0.32))
    c = CsvController(tc, os.path.join(golden_dir, "sim_init_bboxes.csv"))
    out = c.predict([-3, 0, 57, 199, 200], relative=False)
    assert np.isnan(out[0]).all() and np.isnan(out[2]).all() and np.isnan(out[4]).all()  # OOB, missed detection, OOB
    assert np.isfinite(out[1]).all() and np.isfinite(out[3]).all()
    with pytest.raises(AssertionError):
        c.predict([])


def test_discretize():
    b = np.array([[10.2, 5.7, 3.1, 4.0], [np.nan, 1, 1, 1], [-5.0, -5.0, 3.0, 3.0], [1395.5, 1590.0, 20.0, 20.0]])
    d, ok = discretize(b, (1600, 1400))
    assert d.dtype == np.int32
    assert d.tolist() == [[10, 5, 4, 5], [0, 0, 0, 0], [0, 0, 0, 0], [1395, 1590, 5, 10]]
    assert ok.tolist() == [True, False, False, True]


def test_box_utilities_match_reference_vectors(golden_dir):
    """BoxUtils.discretize / center and BoxConverter.to_xywh of the REAL reference (tests/golden/bbox_utils.npz, written by
    make_golden.py --r2 from wtracker/utils/bbox_utils.py:76-167,232-260) on boxes with NaN rows, negative corners, boxes
    hanging over the bounds, empty boxes: bit for bit."""
    g = np.load(os.path.join(golden_dir, "bbox_utils.npz"))
    bounds = tuple(int(v) for v in g["bounds"])
    d, ok = discretize(g["xywh"], bounds)
    assert d.dtype == g["disc"].dtype == np.int32
    np.testing.assert_array_equal(d, g["disc"])
    np.testing.assert_array_equal(ok, g["legal"])
    assert 10 < ok.sum() < len(ok)
    d32, _ = discretize(g["xywh"].astype(np.float32), bounds)  # the detector's float32 boxes (no NaN row -> float32 stack)
    np.testing.assert_array_equal(d32, g["disc_f32"])
    finite = np.isfinite(g["xywh"]).all(axis=1)
    np.testing.assert_array_equal(box_center(g["xywh"])[finite], g["center"][finite])
    np.testing.assert_array_equal(xyxy_to_xywh(g["xyxy"])[finite], g["xywh_from_xyxy"][finite])
    np.testing.assert_array_equal(yolo_to_xywh(g["yolo"])[finite], g["xywh_from_yolo"][finite])
    # the reference zeroes NaN rows of the CALLER's array (bbox_utils.py:139-140); the harness logger reproduces the visible
    # consequence (missed detections are logged as 0,0,0,0) without mutating its input
    assert (g["disc_input_after"][~finite] == 0).all()


def test_polyfit_unsorted_times_and_weights_match_reference(golden_dir):
    """polyfit_controller.py:28 sorts sample_times but NOT weights: weight i belongs to the i-th smallest time."""
    init = os.path.join(golden_dir, "sim_init_bboxes.csv")
    cases = json.load(open(os.path.join(golden_dir, "polyfit_cases.json")))
    assert set(cases) >= {"unsorted_weighted", "unsorted_default_weights", "sorted_cubic"}
    for name, c in cases.items():
        cfg = PolyfitConfig(**c["config"])
        assert list(cfg.sample_times) == c["sample_times_after"] and [float(w) for w in cfg.weights] == c["weights_after"]
        _, moves = run(lambda tc: PolyfitController(tc, cfg, init))
        assert moves == c["moves"], name
    assert cases["unsorted_weighted"]["sample_times_after"] == [-9, -6, -3, 0, 2, 4]
    assert cases["unsorted_weighted"]["weights_after"] == [1, 1, 2, 3, 4, 5]


def test_polyfit_at_the_highest_admitted_degree_matches_reference(golden_dir):
    """Degree 7 over 16 sample times on both sides of zero, a weighted quintic, and a quartic whose first cycles have fewer
    samples than coefficients (tests/golden/polyfit_highdeg.json, the real reference's PolyfitController): integer moves equal."""
    import warnings

    init = os.path.join(golden_dir, "sim_init_bboxes.csv")
    cases = json.load(open(os.path.join(golden_dir, "polyfit_highdeg.json")))
    assert set(cases) == {"deg7_16_times", "deg5_weighted", "deg4_short_history"}
    for name, c in cases.items():
        cfg = PolyfitConfig(**c["config"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # numpy's RankWarning on the under-determined first cycles (the reference gets it too)
            _, moves = run(lambda tc: PolyfitController(tc, cfg, init))
        assert moves == c["moves"], name


def test_deferred_track_log_writes_the_same_rows(golden_dir, tmp_path):
    """TrackLogger(deferred=True) with a controller that offers the asynchronous cycle batch (here: CsvController behind a token that is only
    evaluated at collection time) writes the rows of the reference's log, one cycle late and the last batch at on_sim_end; without the
    asynchronous pair it behaves as the immediate logger."""
    init = os.path.join(golden_dir, "sim_init_bboxes.csv")
    golden = read_log(os.path.join(golden_dir, "sim_csv_bboxes.csv"))
    ec = ExperimentConfig(**EXP0)
    tc = TimingConfig(ec, 100, 40, 50, (4, 4), (0.32, 0.32))

    class AsyncCsv(CsvController):
        launched = collected = 0

        def _cycle_predict_all_async(self, sim):
            AsyncCsv.launched += 1
            return {"rows": CsvController._cycle_predict_all(self, sim)}  # a real controller would only enqueue here

        def _cycle_collect(self, token):
            AsyncCsv.collected += 1
            return token["rows"]

    path = tmp_path / "deferred.csv"
    log = TrackLogger(AsyncCsv(tc, init), csv_path=str(path), deferred=True)
    seen = []
    inner = log.on_cycle_end

    def spy(sim):
        inner(sim)
        seen.append(len(log.rows))

    log.on_cycle_end = spy
    Simulator(tc, ec, log).run()
    n = tc.cycle_frame_num
    assert seen[:3] == [0, n, 2 * n]  # a cycle's rows appear at the NEXT cycle's end
    assert AsyncCsv.launched == AsyncCsv.collected and AsyncCsv.launched >= 3
    assert_rows_equal(log.rows, golden)
    assert_rows_equal(read_log(str(path)), golden)
    plain = TrackLogger(CsvController(tc, init), deferred=True)  # no asynchronous pair: immediate rows
    Simulator(tc, ec, plain).run()
    assert_rows_equal(plain.rows, golden)
