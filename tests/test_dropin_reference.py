"""Drop-in proof against the reference's OWN driver objects (CPU only).

tests/golden/dropin_reference_driver.json was written by `tests/golden/make_golden.py --r3` in the build container: the product's
host controllers (wtracker_amd.controllers.CsvController / OptimalController / PolyfitController) were driven by the REAL reference's
`Simulator.run` + `LoggingController` (wtracker/sim/simulator.py:140-194, wtracker/sim/sim_controllers/logging_controller.py:64-224)
and the bboxes.csv files compared byte for byte with the ones the reference's own controllers produce there.

  test_fixture_*           what the fixture records
  test_harness_logs_*      ties the fixture to the code of THIS checkout: the same product controllers, driven by the repo's own
                           harness (wtracker_amd.sim.Simulator + TrackLogger), write a file whose sha256 is the reference log's
  test_live_*              where /root/reference exists (the build container, never the GPU box) the comparison is run again, live
"""
import hashlib
import json
import os

import pytest

from wtracker_amd.controllers import CsvController, OptimalController, PolyfitConfig, PolyfitController
from wtracker_amd.sim import ExperimentConfig, TimingConfig, TrackLogger
from harness.sim_harness import Simulator

EXP0 = dict(name="exp0", num_frames=200, frames_per_sec=60, orig_resolution=(1600, 1400), px_per_mm=90, init_position=(1300, 1200))
PF = dict(degree=2, sample_times=[2, -9, 0, -3, 4, -6], weights=[1, 1, 2, 3, 4, 5])


def _fixture(golden_dir):
    return json.load(open(os.path.join(golden_dir, "dropin_reference_driver.json")))


def test_fixture_records_byte_equal_logs_for_every_controller(golden_dir):
    fx = _fixture(golden_dir)
    assert "wtracker.sim.simulator.Simulator" in fx["driver"] and "LoggingController" in fx["driver"]
    want = {f"{k}_{t}ms" for k in ("csv", "optimal", "polyfit") for t in (100, 200)}
    assert set(fx["controllers"]) == want
    for name, c in fx["controllers"].items():
        assert c["byte_equal"] is True and c["reference_sha256"] == c["product_sha256"], name
        assert c["rows"] in (198, 195) and c["bytes"] > 20000  # 200 frames: 22 nine-frame cycles / 13 fifteen-frame cycles, the last one never logged


@pytest.mark.parametrize("kind", ["csv", "optimal", "polyfit"])
@pytest.mark.parametrize("timing", [(100, 40, 50), (200, 40, 50)])
def test_harness_logs_hash_to_the_reference_logs(golden_dir, tmp_path, kind, timing):
    init = os.path.join(golden_dir, "sim_init_bboxes.csv")
    ec = ExperimentConfig(**EXP0)
    tc = TimingConfig(ec, *timing, (4, 4), (0.32, 0.32))
    ctrl = {"csv": lambda: CsvController(tc, init), "optimal": lambda: OptimalController(tc, init),
            "polyfit": lambda: PolyfitController(tc, PolyfitConfig(**PF), init)}[kind]()
    path = str(tmp_path / "bboxes.csv")
    Simulator(tc, ec, TrackLogger(ctrl, csv_path=path)).run()
    got = hashlib.sha256(open(path, "rb").read()).hexdigest()
    assert got == _fixture(golden_dir)["controllers"][f"{kind}_{timing[0]}ms"]["reference_sha256"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/wtracker"), reason="the reference only exists in the build container")
def test_live_product_controllers_inside_the_reference_driver(golden_dir):
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(golden_dir, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    out = mg.dropin_check(verbose=False)
    fx = _fixture(golden_dir)
    for name, c in out["controllers"].items():
        assert c["byte_equal"], name
        assert c["reference_sha256"] == fx["controllers"][name]["reference_sha256"], name
