R=$GRAFT_REPO_ROOT
cd $R
python3 - <<'P'
import os, subprocess, sys, numpy as np
code = '''
import sys, numpy as np
sys.path.insert(0, "%s")
from wtracker_amd import hip, yolo_spec as ys, frames as fr
w = ys.synthetic_weights("s", 1, seed=0)
det = hip.HipYolo(w, (640, 640), 8, dtype="f16x3", plan="throughput")
f = fr.diverse_frames(8, 640, seed=5)
x, c, a = det.predict_host(f, conf=0.1)
b, k = det.debug_head(8)
np.savez(sys.argv[1], x=x, a=a, b=b, k=k)
''' % os.getcwd()
outs = []
for v in ("0", "1"):
    env = dict(os.environ, WTK_HALO_2CU=v, WTK_NO_SK_MIXED="1", WTK_SMALL_NARROW="0")
    p = f"/tmp/2cu_{v}.npz"
    subprocess.run([sys.executable, "-c", code, p], env=env, check=True)
    outs.append(np.load(p))
for key in ("x", "a", "b", "k"):
    print(key, "equal" if np.array_equal(outs[0][key], outs[1][key]) else f"DIFFERENT max {np.abs(outs[0][key]-outs[1][key]).max()}")
P
for V in 0 1 0 1; do echo "== WTK_HALO_2CU=$V"; WTK_HALO_2CU=$V python3 tools/gpu_sessions/time_mode.py --dtype f16x3 --batch 64 --size 640 --steps 30 --lanes 1 2>&1 | grep "ms per step"; WTK_HALO_2CU=$V GPU_MAX_HW_QUEUES=8 python3 tools/gpu_sessions/time_mode.py --dtype f16x3 --batch 64 --size 640 --steps 30 --lanes 2 2>&1 | grep "ms per step"; done
