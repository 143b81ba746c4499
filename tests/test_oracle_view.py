"""CPU: oracle/view_oracle.py (restatement of view_controller.py:45-61,93-172) against hand-computed windows and against
the product harness' ViewController (wtracker_amd/sim.py) — two independent statements of the same slicing."""
import numpy as np

from oracle import view_oracle as vo
from harness.sim_harness import ArrayReader, ViewController


def test_hand_computed_window_with_replicate_border():
    f = np.arange(20, dtype=np.uint8).reshape(4, 5)  # rows 0..3, cols 0..4
    # camera (w, h) = (3, 3): pad 1; position (0, 0) -> window rows/cols -1..1 -> replicate of row 0 / col 0
    v = vo.camera_view(f, (0, 0), (3, 3))
    np.testing.assert_array_equal(v, [[0, 0, 1], [0, 0, 1], [5, 5, 6]])
    v = vo.camera_view(f, (4, 3), (3, 3))  # bottom-right corner
    np.testing.assert_array_equal(v, [[13, 14, 14], [18, 19, 19], [18, 19, 19]])
    # non-square view: rows are sliced by w, columns by h (view_controller.py:171)
    v = vo.custom_view(f, (2, 2), (5, 5), (1, 3))
    assert v.shape == (1, 3)
    # pad 2: x = 2 + 2 - 1 // 2 = 4, y = 2 + 2 - 3 // 2 = 3 (padded) -> rows 3:4 = frame row 1, cols 4:7 = frame cols 2..4
    np.testing.assert_array_equal(v, [[7, 8, 9]])
    assert vo.clamp_position(-3, 99, (4, 5)) == (0, 3)


def test_view_oracle_equals_harness_view_controller():
    rng = np.random.default_rng(0)
    for shape, cam, mic in [((90, 120), (36, 36), (9, 9)), ((64, 48, 3), (21, 33), (5, 7)), ((50, 50), (50, 50), (11, 11))]:
        frames = rng.integers(0, 256, size=(5, *shape), dtype=np.uint8)
        vc = ViewController(ArrayReader(frames), camera_size=cam, micro_size=mic)
        H, W = shape[:2]
        for i, (x, y) in enumerate([(0, 0), (W - 1, H - 1), (W // 2, H // 2), (3, H - 2), (W + 40, -7)]):
            vc.seek(i)
            vc.set_position(x, y)
            pos = vo.clamp_position(x, y, (H, W))
            assert tuple(int(p) for p in vc.position) == pos
            np.testing.assert_array_equal(vc.camera_view(), vo.camera_view(frames[i], pos, cam))
            np.testing.assert_array_equal(vc.micro_view(), vo.micro_view(frames[i], pos, cam, mic))
