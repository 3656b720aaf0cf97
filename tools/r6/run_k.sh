#!/bin/bash
# round 6, GPU call K: fused tail at the benchmark size - halo kernel (switch 0 = 0) vs implicit GEMM (= 1), whole step, same process
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6k; rm -rf $O; mkdir -p $O
timeout 400 python tools/ab_option.py 0 0 1 --rounds 5 > $O/ab_opt0.txt 2>&1
tail -3 $O/ab_opt0.txt
