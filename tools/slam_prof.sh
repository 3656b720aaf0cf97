#!/bin/bash
# rocprofv3 kernel trace of one SLAM-scale stage: tools/slam_prof.sh encode|sched|replay [tag] [reps]   -> gpurun_out/slam_<stage><tag>/t_results.db
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
w=${1:-encode}; tag=${2:-}
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/slam_$w$tag -o t -- python $R/tools/slam_trace.py $w ${3:-20} 2>&1 | grep "per call"
