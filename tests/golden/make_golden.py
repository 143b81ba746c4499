#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box).  Nothing of the reference's source is copied: the outputs are
numeric data only (model parameters, input/output vectors, CSV logs).

The reference imports `tkinter`, `cv2` and `ultralytics` at module import time
(wtracker/utils/gui_utils.py:1, wtracker/utils/io_utils.py:1,
wtracker/sim/sim_controllers/yolo_controller.py:6).  None of them is installed
here and none of their functions is executed on the CSV/ResMLP path with all
LogConfig.save_* flags off, so empty placeholder modules are registered in
sys.modules before the import.

Outputs
  resmlp_100ms.npz / resmlp_200ms.npz   raw state-dict tensors, IOConfig, golden (x, y) pairs
  sim_init_bboxes.csv                   seeded synthetic pre-detected track (input)
  sim_mlp_bboxes.csv                    reference LoggingController(MLPController) log
  sim_csv_bboxes.csv                    reference LoggingController(CsvController) log
  sim_moves.json                        per-cycle (dx, dy) returned by provide_movement_vector
  timing.json                           TimingConfig numbers for several (imaging, pred, moving) triples
Round 2 (`--r2`: writes ONLY these, the fixtures above stay byte-identical):
  bbox_utils.npz                        BoxUtils.discretize / center / round and BoxConverter.to_xywh / to_xyxy of the real
                                        reference (wtracker/utils/bbox_utils.py:76-167,232-260) on seeded boxes with NaN rows,
                                        negative corners, boxes hanging over the bounds and empty boxes
  polyfit_cases.json                    per-cycle (dx, dy) of the reference PolyfitController for several PolyfitConfigs,
                                        among them UNSORTED sample_times with non-uniform weights (polyfit_controller.py:28
                                        sorts the times, not the weights), and the weights / sample_times the config ends up with
Round 3 (`--r3`: writes ONLY these):
  dropin_reference_driver.json          the PRODUCT's host controllers (wtracker_amd.controllers.CsvController / OptimalController /
                                        PolyfitController, and HipMLPController's CPU twin is not needed: the MLP log is pinned
                                        above) driven by the REAL reference's Simulator + LoggingController
                                        (simulator.py:140-194, logging_controller.py:64-224): for every controller the sha256 and
                                        row count of the bboxes.csv the reference's LoggingController wrote with the reference's
                                        own controller inside and with the product's controller inside, and whether the two files
                                        are byte-equal — the "simulate.ipynb is a drop-in" claim checked where it can be checked
  polyfit_highdeg.json                  numpy-identical fits at the highest degree the device solver admits (7, 16 sample times on
                                        both sides of zero): the reference PolyfitController's per-cycle integer moves
"""
import hashlib
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _register_placeholders():
    tk = types.ModuleType("tkinter")
    tk.Tk = object
    fd = types.ModuleType("tkinter.filedialog")
    tk.filedialog = fd
    cv2 = types.ModuleType("cv2")
    cv2.IMREAD_GRAYSCALE = 0
    cv2.IMREAD_COLOR = 1
    ul = types.ModuleType("ultralytics")
    ul.YOLO = type("YOLO", (), {})
    for name, mod in (("tkinter", tk), ("tkinter.filedialog", fd), ("cv2", cv2), ("ultralytics", ul)):
        sys.modules.setdefault(name, mod)


def synthetic_track(num_frames: int, seed: int, start=(1300.0, 1200.0), nan_rows=(57,)) -> np.ndarray:
    """Seeded random-walk head track, SURVEY.md §8d: speed mean 0.54 px/frame, sigma 0.28,
    head box ~13.8 x 14.6 px.  Returns [N,4] xywh float64 with NaN rows = missed detections."""
    rng = np.random.default_rng(seed)
    heading = rng.uniform(0, 2 * np.pi)
    pos = np.array(start, dtype=np.float64)
    out = np.empty((num_frames, 4), dtype=np.float64)
    for i in range(num_frames):
        speed = max(0.0, rng.normal(0.54, 0.28))
        heading += rng.normal(0.0, 0.15)
        pos = pos + speed * np.array([np.cos(heading), np.sin(heading)])
        w = 13.8 + rng.normal(0, 0.6)
        h = 14.6 + rng.normal(0, 0.6)
        out[i] = (pos[0] - w / 2, pos[1] - h / 2, w, h)
    for r in nan_rows:
        if 0 <= r < num_frames:
            out[r] = np.nan
    return out


def model_inputs(io_config, n: int, seed: int) -> np.ndarray:
    """Inputs shaped like MLPController builds them (mlp_controllers.py:50-56): 7 xywh boxes,
    x/y relative to the first box's corner."""
    rng = np.random.default_rng(seed)
    k = len(io_config.input_frames)
    x = np.zeros((n, k, 4), dtype=np.float64)
    for i in range(n):
        v = rng.normal(0, 0.8, size=2)
        for j, f in enumerate(io_config.input_frames):
            x[i, j, 0:2] = v * f + rng.normal(0, 0.3, size=2)
            x[i, j, 2] = 13.8 + rng.normal(0, 0.8)
            x[i, j, 3] = 14.6 + rng.normal(0, 0.8)
        x[i, :, 0] -= x[i, 0, 0]
        x[i, :, 1] -= x[i, 0, 1]
    return x.reshape(n, k * 4).astype(np.float32)


def main():
    _register_placeholders()
    sys.path.insert(0, REF)
    import torch

    torch.manual_seed(0)
    torch.set_num_threads(1)

    from wtracker.sim.config import TimingConfig, ExperimentConfig
    from wtracker.sim.simulator import Simulator
    from wtracker.sim.sim_controllers.csv_controller import CsvController
    from wtracker.sim.sim_controllers.mlp_controllers import MLPController
    from wtracker.sim.sim_controllers.logging_controller import LoggingController, LogConfig

    models = {
        "100ms": "ResMLP(imaging-100ms_pred-40ms_moving-50ms).pt",
        "200ms": "ResMLP(imaging-200ms_pred-40ms_moving-50ms).pt",
    }
    loaded = {}
    for tag, fname in models.items():
        m = torch.load(os.path.join(REF, "models", fname), weights_only=False, map_location="cpu")
        m.eval()
        loaded[tag] = m
        sd = m.state_dict()
        arrays = {"sd::" + k: v.detach().cpu().numpy() for k, v in sd.items()}
        digest = hashlib.sha256(b"".join(np.ascontiguousarray(v).tobytes() for v in arrays.values())).hexdigest()
        x = model_inputs(m.io_config, 256, seed=1234)
        with torch.no_grad():
            y_batch = m.forward(torch.from_numpy(x)).numpy()
            y_single = np.concatenate([m.forward(torch.from_numpy(x[i : i + 1])).numpy() for i in range(x.shape[0])])
            y_zero = m.forward(torch.zeros(1, x.shape[1])).numpy()
        np.savez(
            os.path.join(HERE, f"resmlp_{tag}.npz"),
            input_frames=np.asarray(m.io_config.input_frames, dtype=np.int64),
            pred_frames=np.asarray(m.io_config.pred_frames, dtype=np.int64),
            x=x,
            y_batch=y_batch,
            y_single=y_single,
            y_zero=y_zero,
            sha256=np.frombuffer(digest.encode(), dtype=np.uint8),
            **arrays,
        )
        print(tag, "params", sum(v.size for k, v in arrays.items() if "num_batches" not in k), "f(0)=", y_zero, digest[:16])

    # ---- timing arithmetic fixtures (sim/config.py:41-67)
    exp_cfg_json = json.load(open(os.path.join(REF, "experiments/exp0/exp_config.json")))
    timing_rows = []
    for fps, ppm in ((60, 90), (60, 88), (30, 92), (50, 91.5)):
        for trip in ((100, 40, 50), (200, 40, 50), (250, 60, 70), (16, 16, 17), (83.4, 33.3, 50.1)):
            ec = ExperimentConfig(name="t", num_frames=10, frames_per_sec=fps, orig_resolution=(1600, 1400), px_per_mm=ppm, init_position=(10, 10))
            tc = TimingConfig(ec, trip[0], trip[1], trip[2], (4, 4), (0.32, 0.32))
            timing_rows.append(
                dict(fps=fps, px_per_mm=ppm, imaging_ms=trip[0], pred_ms=trip[1], moving_ms=trip[2],
                     imaging_frame_num=tc.imaging_frame_num, pred_frame_num=tc.pred_frame_num,
                     moving_frame_num=tc.moving_frame_num, cycle_frame_num=tc.cycle_frame_num,
                     camera_size_px=list(tc.camera_size_px), micro_size_px=list(tc.micro_size_px),
                     ms_per_frame=tc.ms_per_frame))
    json.dump(timing_rows, open(os.path.join(HERE, "timing.json"), "w"), indent=1)

    # ---- golden sim loop (SURVEY.md §3.2, §8d "Synthetic tracks for C1")
    num_frames = 200
    track = synthetic_track(num_frames, seed=0)
    init_csv = os.path.join(HERE, "sim_init_bboxes.csv")
    with open(init_csv, "w") as f:
        f.write("frame,wrm_x,wrm_y,wrm_w,wrm_h\n")
        for i, r in enumerate(track):
            f.write(f"{i}," + ",".join("" if not np.isfinite(v) else repr(float(v)) for v in r) + "\n")

    moves = {}

    def run(kind: str, out_name: str, timing=(100, 40, 50), model_tag="100ms"):
        ec = ExperimentConfig(
            name="exp0", num_frames=num_frames, frames_per_sec=exp_cfg_json["frames_per_sec"],
            orig_resolution=tuple(exp_cfg_json["orig_resolution"]), px_per_mm=exp_cfg_json["px_per_mm"],
            init_position=tuple(exp_cfg_json["init_position"]))
        tc = TimingConfig(ec, timing[0], timing[1], timing[2], (4, 4), (0.32, 0.32))
        if kind == "mlp":
            ctrl = MLPController(tc, init_csv, loaded[model_tag], max_speed=0.9)
        else:
            ctrl = CsvController(tc, init_csv)
        rec = []
        orig = ctrl.provide_movement_vector

        def wrapped(sim):
            dx, dy = orig(sim)
            rec.append([int(sim.frame_number), int(dx), int(dy)])
            return dx, dy

        ctrl.provide_movement_vector = wrapped
        tmp = tempfile.mkdtemp(prefix="wtk_golden_")
        try:
            lc = LogConfig(root_folder=tmp, save_mic_view=False, save_cam_view=False, save_err_view=False, save_wrm_view=False)
            sim = Simulator(tc, ec, LoggingController(ctrl, lc))
            sim.run()
            shutil.copy(lc.bbox_file_path, os.path.join(HERE, out_name))
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        moves[out_name] = rec

    from wtracker.sim.sim_controllers.optimal_controller import OptimalController
    from wtracker.sim.sim_controllers.polyfit_controller import PolyfitConfig, PolyfitController

    def run_other(kind: str, out_name: str, timing=(100, 40, 50)):
        ec = ExperimentConfig(
            name="exp0", num_frames=num_frames, frames_per_sec=exp_cfg_json["frames_per_sec"],
            orig_resolution=tuple(exp_cfg_json["orig_resolution"]), px_per_mm=exp_cfg_json["px_per_mm"],
            init_position=tuple(exp_cfg_json["init_position"]))
        tc = TimingConfig(ec, timing[0], timing[1], timing[2], (4, 4), (0.32, 0.32))
        if kind == "optimal":
            ctrl = OptimalController(tc, init_csv)
        else:
            ctrl = PolyfitController(tc, PolyfitConfig(degree=2, sample_times=[-9, -6, -3, 0, 2, 4], weights=[1, 1, 2, 3, 4, 5]), init_csv)
        rec = []
        orig = ctrl.provide_movement_vector

        def wrapped(sim):
            dx, dy = orig(sim)
            rec.append([int(sim.frame_number), int(dx), int(dy)])
            return dx, dy

        ctrl.provide_movement_vector = wrapped
        tmp = tempfile.mkdtemp(prefix="wtk_golden_")
        try:
            lc = LogConfig(root_folder=tmp, save_mic_view=False, save_cam_view=False, save_err_view=False, save_wrm_view=False)
            Simulator(tc, ec, LoggingController(ctrl, lc)).run()
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        moves[out_name] = rec

    run_other("optimal", "sim_optimal")
    run_other("polyfit", "sim_polyfit")

    # ---- training-pair builder (neural/dataset.py:42-96) on the same synthetic log
    from wtracker.neural.config import DatasetConfig
    from wtracker.neural.dataset import NumpyDataset

    ds = NumpyDataset.create_from_config(DatasetConfig(list(loaded["100ms"].io_config.input_frames), list(loaded["100ms"].io_config.pred_frames), init_csv))
    np.savez(os.path.join(HERE, "dataset_100ms.npz"), X=ds.X.numpy(), y=ds.y.numpy())
    print("dataset", tuple(ds.X.shape), tuple(ds.y.shape))

    run("mlp", "sim_mlp_bboxes.csv")
    run("csv", "sim_csv_bboxes.csv")
    run("mlp", "sim_mlp200_bboxes.csv", timing=(200, 40, 50), model_tag="200ms")
    json.dump(moves, open(os.path.join(HERE, "sim_moves.json"), "w"))
    for k, v in moves.items():
        print(k, "cycles", len(v), "last", v[-1])


def main_r2():
    """Round-2 fixtures: numeric outputs of the real reference only (see the module docstring)."""
    _register_placeholders()
    sys.path.insert(0, REF)
    from wtracker.sim.config import ExperimentConfig, TimingConfig
    from wtracker.sim.sim_controllers.logging_controller import LogConfig, LoggingController
    from wtracker.sim.sim_controllers.polyfit_controller import PolyfitConfig, PolyfitController
    from wtracker.sim.simulator import Simulator
    from wtracker.utils.bbox_utils import BoxConverter, BoxFormat, BoxUtils

    # ---- bbox utilities
    rng = np.random.default_rng(77)
    n = 64
    xywh = np.stack([rng.uniform(-30, 1420, n), rng.uniform(-30, 1620, n), rng.uniform(0.0, 40, n), rng.uniform(0.0, 40, n)], axis=1)
    xywh[3] = np.nan
    xywh[9, 2] = np.nan
    xywh[12] = [-5.0, -5.0, 3.0, 3.0]            # entirely outside (negative side)
    xywh[13] = [1395.5, 1590.0, 20.0, 20.0]      # hangs over both bounds
    xywh[14] = [10.0, 10.0, 0.0, 5.0]            # zero width
    xywh[15] = [1400.0, 20.0, 5.0, 5.0]          # starts exactly at the bound
    xywh[16] = [10.2, 5.7, 3.1, 4.0]
    xywh[17] = [7.0, 8.0, 2.0, 2.0]              # integer box: floor / ceil leave it alone
    bounds = (1600, 1400)
    d_in = xywh.copy()
    disc, legal = BoxUtils.discretize(d_in, bounds, BoxFormat.XYWH)  # zeroes NaN rows of d_in in place (bbox_utils.py:139-140)
    xyxy = BoxConverter.to_xyxy(xywh.copy(), BoxFormat.XYWH)
    yolo = BoxConverter.to_yolo(xywh.copy(), BoxFormat.XYWH)
    np.savez(os.path.join(HERE, "bbox_utils.npz"), xywh=xywh, bounds=np.asarray(bounds), disc=disc, legal=legal, disc_input_after=d_in,
             xyxy=xyxy, yolo=yolo, xywh_from_xyxy=BoxConverter.to_xywh(xyxy.copy(), BoxFormat.XYXY),
             xywh_from_yolo=BoxConverter.to_xywh(yolo.copy(), BoxFormat.YOLO), center=BoxUtils.center(xywh.copy()),
             round_xywh=BoxUtils.round(np.nan_to_num(xywh.copy(), nan=1.5), BoxFormat.XYWH),
             disc_f32=BoxUtils.discretize(xywh.astype(np.float32), bounds, BoxFormat.XYWH)[0])
    print("bbox_utils", disc.dtype, int(legal.sum()), "legal of", n)

    # ---- PolyfitController with unsorted times / non-uniform weights
    exp_cfg_json = json.load(open(os.path.join(REF, "experiments/exp0/exp_config.json")))
    init_csv = os.path.join(HERE, "sim_init_bboxes.csv")
    cases = {
        "unsorted_weighted": dict(degree=2, sample_times=[2, -9, 0, -3, 4, -6], weights=[1, 1, 2, 3, 4, 5]),
        "unsorted_default_weights": dict(degree=1, sample_times=[4, 0, -6], weights=None),
        "sorted_cubic": dict(degree=3, sample_times=[-12, -9, -6, -3, 0, 2, 4, 5], weights=[0.5, 1, 1, 2, 3, 4, 5, 8]),
    }
    out = {}
    for name, kw in cases.items():
        ec = ExperimentConfig(name="exp0", num_frames=200, frames_per_sec=exp_cfg_json["frames_per_sec"],
                              orig_resolution=tuple(exp_cfg_json["orig_resolution"]), px_per_mm=exp_cfg_json["px_per_mm"],
                              init_position=tuple(exp_cfg_json["init_position"]))
        tc = TimingConfig(ec, 100, 40, 50, (4, 4), (0.32, 0.32))
        cfg = PolyfitConfig(**kw)
        ctrl = PolyfitController(tc, cfg, init_csv)
        rec = []
        orig = ctrl.provide_movement_vector

        def wrapped(sim, orig=orig, rec=rec):
            dx, dy = orig(sim)
            rec.append([int(sim.frame_number), int(dx), int(dy)])
            return dx, dy

        ctrl.provide_movement_vector = wrapped
        tmp = tempfile.mkdtemp(prefix="wtk_golden_")
        try:
            lc = LogConfig(root_folder=tmp, save_mic_view=False, save_cam_view=False, save_err_view=False, save_wrm_view=False)
            Simulator(tc, ec, LoggingController(ctrl, lc)).run()
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        out[name] = dict(config=dict(degree=kw["degree"], sample_times=kw["sample_times"], weights=kw["weights"]),
                         sample_times_after=[int(t) for t in cfg.sample_times], weights_after=[float(w) for w in cfg.weights], moves=rec)
        print(name, "cycles", len(rec), "last", rec[-1])
    json.dump(out, open(os.path.join(HERE, "polyfit_cases.json"), "w"))


def dropin_check(verbose: bool = True) -> dict:
    """Run the reference's own driver objects (Simulator.run + LoggingController) twice per controller kind — once around the
    reference's controller, once around the product's — on the seeded track, and compare the bboxes.csv files byte for byte.
    Importable: tests/test_dropin_reference.py calls it when /root/reference is present."""
    _register_placeholders()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    repo = os.path.dirname(os.path.dirname(HERE))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from wtracker.sim.config import ExperimentConfig, TimingConfig
    from wtracker.sim.sim_controllers.csv_controller import CsvController as RefCsv
    from wtracker.sim.sim_controllers.logging_controller import LogConfig, LoggingController
    from wtracker.sim.sim_controllers.optimal_controller import OptimalController as RefOptimal
    from wtracker.sim.sim_controllers.polyfit_controller import PolyfitConfig as RefPolyfitConfig
    from wtracker.sim.sim_controllers.polyfit_controller import PolyfitController as RefPolyfit
    from wtracker.sim.simulator import Simulator

    from wtracker_amd import controllers as ours

    exp_cfg_json = json.load(open(os.path.join(REF, "experiments/exp0/exp_config.json")))
    init_csv = os.path.join(HERE, "sim_init_bboxes.csv")
    pf = dict(degree=2, sample_times=[2, -9, 0, -3, 4, -6], weights=[1, 1, 2, 3, 4, 5])
    kinds = {
        "csv": (lambda tc: RefCsv(tc, init_csv), lambda tc: ours.CsvController(tc, init_csv)),
        "optimal": (lambda tc: RefOptimal(tc, init_csv), lambda tc: ours.OptimalController(tc, init_csv)),
        "polyfit": (lambda tc: RefPolyfit(tc, RefPolyfitConfig(**pf), init_csv), lambda tc: ours.PolyfitController(tc, ours.PolyfitConfig(**pf), init_csv)),
    }
    out = {"driver": "wtracker.sim.simulator.Simulator.run + wtracker.sim.sim_controllers.logging_controller.LoggingController (the real reference)",
           "track": "tests/golden/sim_init_bboxes.csv", "timings_ms": [[100, 40, 50], [200, 40, 50]], "controllers": {}}
    for name, (make_ref, make_ours) in kinds.items():
        for timing in ((100, 40, 50), (200, 40, 50)):
            logs = []
            for make in (make_ref, make_ours):
                ec = ExperimentConfig(name="exp0", num_frames=200, frames_per_sec=exp_cfg_json["frames_per_sec"],
                                      orig_resolution=tuple(exp_cfg_json["orig_resolution"]), px_per_mm=exp_cfg_json["px_per_mm"],
                                      init_position=tuple(exp_cfg_json["init_position"]))
                tc = TimingConfig(ec, *timing, (4, 4), (0.32, 0.32))
                tmp = tempfile.mkdtemp(prefix="wtk_dropin_")
                try:
                    lc = LogConfig(root_folder=tmp, save_mic_view=False, save_cam_view=False, save_err_view=False, save_wrm_view=False)
                    Simulator(tc, ec, LoggingController(make(tc), lc)).run()
                    logs.append(open(lc.bbox_file_path, "rb").read())
                finally:
                    shutil.rmtree(tmp, ignore_errors=True)
            key = f"{name}_{timing[0]}ms"
            out["controllers"][key] = {"reference_sha256": hashlib.sha256(logs[0]).hexdigest(), "product_sha256": hashlib.sha256(logs[1]).hexdigest(),
                                       "rows": logs[0].count(b"\n") - 1, "bytes": len(logs[0]), "byte_equal": logs[0] == logs[1]}
            if verbose:
                print("dropin", key, out["controllers"][key])
    return out


def main_r3():
    """Round-3 fixtures (see the module docstring)."""
    out = dropin_check()
    assert all(c["byte_equal"] for c in out["controllers"].values()), "a product controller's log differs from the reference controller's"
    json.dump(out, open(os.path.join(HERE, "dropin_reference_driver.json"), "w"), indent=1)

    from wtracker.sim.config import ExperimentConfig, TimingConfig
    from wtracker.sim.sim_controllers.logging_controller import LogConfig, LoggingController
    from wtracker.sim.sim_controllers.polyfit_controller import PolyfitConfig, PolyfitController
    from wtracker.sim.simulator import Simulator

    exp_cfg_json = json.load(open(os.path.join(REF, "experiments/exp0/exp_config.json")))
    init_csv = os.path.join(HERE, "sim_init_bboxes.csv")
    cases = {
        # the widest problem the device solver admits: degree 7, 16 sample times on both sides of zero (scaled Vandermonde condition ~1e9)
        "deg7_16_times": dict(degree=7, sample_times=[-45, -40, -36, -31, -27, -22, -18, -15, -12, -9, -6, -3, 0, 2, 4, 5], weights=None),
        "deg5_weighted": dict(degree=5, sample_times=[-27, -20, -18, -11, -9, -6, -3, 0, 2, 4], weights=[0.3, 0.5, 0.8, 1, 1, 2, 2, 3, 4, 6]),
        # more coefficients than samples in the first cycles (history not there yet): minimum-norm solutions
        "deg4_short_history": dict(degree=4, sample_times=[-30, -20, -10, 0, 3], weights=None),
    }
    res = {}
    for name, kw in cases.items():
        ec = ExperimentConfig(name="exp0", num_frames=200, frames_per_sec=exp_cfg_json["frames_per_sec"],
                              orig_resolution=tuple(exp_cfg_json["orig_resolution"]), px_per_mm=exp_cfg_json["px_per_mm"],
                              init_position=tuple(exp_cfg_json["init_position"]))
        tc = TimingConfig(ec, 100, 40, 50, (4, 4), (0.32, 0.32))
        cfg = PolyfitConfig(**kw)
        ctrl = PolyfitController(tc, cfg, init_csv)
        rec = []
        orig = ctrl.provide_movement_vector

        def wrapped(sim, orig=orig, rec=rec):
            dx, dy = orig(sim)
            rec.append([int(sim.frame_number), int(dx), int(dy)])
            return dx, dy

        ctrl.provide_movement_vector = wrapped
        tmp = tempfile.mkdtemp(prefix="wtk_golden_")
        try:
            lc = LogConfig(root_folder=tmp, save_mic_view=False, save_cam_view=False, save_err_view=False, save_wrm_view=False)
            Simulator(tc, ec, LoggingController(ctrl, lc)).run()
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        res[name] = dict(config=dict(degree=kw["degree"], sample_times=kw["sample_times"], weights=kw["weights"]),
                         sample_times_after=[int(t) for t in cfg.sample_times], weights_after=[float(w) for w in cfg.weights], moves=rec)
        print(name, "cycles", len(rec), "last", rec[-1])
    json.dump(res, open(os.path.join(HERE, "polyfit_highdeg.json"), "w"))


if __name__ == "__main__":
    if "--r3" in sys.argv:
        main_r3()
    elif "--r2" in sys.argv:
        main_r2()
    else:
        main()
        main_r2()
