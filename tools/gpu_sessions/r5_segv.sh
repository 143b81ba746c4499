R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1000 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint" -ex run -ex bt -ex "thread apply all bt 12" --args python3 -m pytest tests -x -q -m gpu > gpurun_out/r5_segv_gdb.log 2>&1; echo "gdb rc $?"; grep -n "SIGSEGV\|^#\|passed\|failed" gpurun_out/r5_segv_gdb.log | head -60
