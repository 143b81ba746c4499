# six-slab ring of the split 64-cout window tiles (WTK_HALO_DEEP) + the bias wait in front of the loop: bit-identity, then timing against the previous library
R=$GRAFT_REPO_ROOT
cd $R
python3 - <<'P'
import os, subprocess, sys, numpy as np
code = '''
import sys, numpy as np
sys.path.insert(0, "%s")
from wtracker_amd import hip, yolo_spec as ys, frames as fr
B, S, cap = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
w = ys.synthetic_weights("s", 1, seed=0)
det = hip.HipYolo(w, (S, S), cap, dtype=sys.argv[5], plan="throughput")
f = fr.diverse_frames(B, S, seed=5)
x, c, a = det.predict_host(f, conf=0.1)
b, k = det.debug_head(B)
np.savez(sys.argv[1], x=x, a=a, b=b, k=k)
''' % os.getcwd()
def run(tag, env, B, S, cap, dtype="f16x3"):
    p = f"/tmp/deep_{tag}.npz"
    subprocess.run([sys.executable, "-c", code, p, str(B), str(S), str(cap), dtype], env=dict(os.environ, **env), check=True)
    return np.load(p)
prev = {"WTK_HIP_LIB": os.getcwd() + "/tools/_bin/libwtk_prev.so"}
for name, B, S, cap, envs, dtype in (("cycle batch", 15, 384, 15, ({"WTK_HALO_DEEP": "0"}, {"WTK_HALO_DEEP": "1"}, prev), "f16x3"),
                              ("cycle batch 640", 15, 640, 16, ({"WTK_HALO_DEEP": "0"}, {"WTK_HALO_DEEP": "1"}, prev), "f16x3"),
                              ("B=24 @640 on a 64-frame handle", 24, 640, 64, ({"WTK_HALO_DEEP": "0"}, {"WTK_HALO_DEEP": "2"}, prev), "f16x3"),
                              ("fp16 B=24 @640", 24, 640, 64, ({}, prev), "fp16"),
                              ("fp32 B=15 @384", 15, 384, 15, ({}, prev), "fp32"),
                              ("fp32 B=24 @640", 24, 640, 64, ({}, prev), "fp32")):
    outs = [run(f"{i}", e, B, S, cap, dtype) for i, e in enumerate(envs)]
    for i in range(1, len(outs)):
        print(name, f"variant {i} vs 0:", " ".join(f"{key}:" + ("equal" if np.array_equal(outs[0][key], outs[i][key]) else f"DIFFERENT({np.abs(outs[0][key].astype(np.float64)-outs[i][key]).max():.3g})") for key in ("x", "a", "b", "k")), flush=True)
P
T="python3 tools/gpu_sessions/time_mode.py"
for rep in 1 2; do
for V in prev 0 1; do
  if [ $V = prev ]; then export WTK_HIP_LIB=$R/tools/_bin/libwtk_prev.so; unset WTK_HALO_DEEP; else unset WTK_HIP_LIB; export WTK_HALO_DEEP=$V; fi
  echo "== $V"; $T --dtype f16x3 --batch 15 --size 384 --steps 300 2>&1 | grep "ms per step"
  $T --dtype f16x3 --batch 15 --size 640 --steps 200 2>&1 | grep "ms per step"
done
for V in prev 0 2; do
  if [ $V = prev ]; then export WTK_HIP_LIB=$R/tools/_bin/libwtk_prev.so; unset WTK_HALO_DEEP; else unset WTK_HIP_LIB; export WTK_HALO_DEEP=$V; fi
  echo "== $V"; $T --dtype f16x3 --batch 256 --size 640 --steps 20 2>&1 | grep "ms per step"
  $T --dtype f16x3 --batch 256 --size 384 --steps 30 2>&1 | grep "ms per step"
done
done
unset WTK_HALO_DEEP
for V in prev new prev new; do
  if [ $V = prev ]; then export WTK_HIP_LIB=$R/tools/_bin/libwtk_prev.so; else unset WTK_HIP_LIB; fi
  echo "== $V"; $T --dtype fp16 --batch 256 --size 640 --steps 20 2>&1 | grep "ms per step"
  $T --dtype fp32 --batch 64 --size 640 --steps 10 2>&1 | grep "ms per step"
  $T --dtype fp32 --batch 15 --size 384 --steps 100 2>&1 | grep "ms per step"
done
