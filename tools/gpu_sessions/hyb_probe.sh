# Does a pipeline worker leave the GPU usable for the next process?  `hyb_probe.sh [worker args...]`
R=$GRAFT_REPO_ROOT
cd $R
probe() { timeout 120 python3 -c "
import torch
try:
    torch.zeros(1, device='cuda'); print('probe $1: GPU ok')
except Exception as e:
    print('probe $1: GPU GONE', str(e)[:80])
"; }
probe before
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout -k 10 200 python3 tests/dist_worker.py --out /tmp/w.npz --backend gloo "$@" > gpurun_out/hyb_probe_worker.log 2>&1; echo "worker rc=$?"; tail -5 gpurun_out/hyb_probe_worker.log
probe after
