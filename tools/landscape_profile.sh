cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --marker-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_f_land -- python3 $GRAFT_REPO_ROOT/bench.py --config landscape --steps 2 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/r3_f_land.log 2>&1
echo rc=$?
tail -c 1500 $GRAFT_REPO_ROOT/gpurun_out/r3_f_land.log
ls $GRAFT_REPO_ROOT/gpurun_out/r3_f_land/*/
