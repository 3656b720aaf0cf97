cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in encode sched; do
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/slam_$w -o t -- python $R/tools/slam_trace.py $w 20 2>&1 | grep "per call"
done
ls $R/gpurun_out/slam_encode
