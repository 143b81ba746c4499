"""Per-wave interval stamps of conv3x3_ws64_kernel (diagnostic build: WTK_EXTRA_HIPCC_FLAGS=-DWTK_WS64_STAMPS, env WTK_WS64_STAMPS=1).
Prints, for the first two blocks, every wave's per-interval cycles: work (stamp 0 -> 1 or 2), vmcnt wait, barrier wait."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["WTK_WS64_STAMPS"] = "1"
os.environ.setdefault("WTK_NO_SIDE_STREAM", "1")
from wtracker_amd import frames as fr, hip, yolo_spec as ys  # noqa: E402

w = ys.synthetic_weights("s", 1, seed=0)
det = hip.HipYolo(w, (640, 640), 64, dtype="fp16")
frames, _ = fr.synthetic_frames(4, 640, seed=1)
frames = np.concatenate([frames] * 16)
for _ in range(3):
    det.predict_host(frames)
lib = hip.load()
buf = np.zeros(2 * 8 * 16 * 4, dtype=np.uint64)
lib.wtk_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.wtk_debug_stamps(buf.ctypes.data, buf.size) == 0
st = buf.reshape(2, 8, 16, 4).astype(np.int64)
for b in range(2):
    for wv in range(8):
        s = st[b, wv]
        t0 = s[15, 3]
        rows = []
        for i in range(15):
            if s[i, 0] == 0 or s[i, 3] == 0:
                break
            work_end = s[i, 1] if s[i, 1] else s[i, 2]
            rows.append("%d:[%s w%5d v%5d b%5d]" % (i, "Q" if s[i, 1] else "P", work_end - s[i, 0], s[i, 2] - work_end, s[i, 3] - s[i, 2]))
        print("blk%d wave%d start+%d  " % (b, wv, st[b, wv, 0, 0] - t0) + " ".join(rows))
print("total cycles loop (wave0 blk0):", st[0, 0, :15, 3].max() - st[0, 0, 15, 3])
