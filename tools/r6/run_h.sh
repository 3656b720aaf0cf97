#!/bin/bash
# round 6, GPU call H: fused DPT tail with transposed accumulators + head.4 on the matrix pipe - goldens, A/B, per-launch time
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6h; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "large_tile or full_size or mid_size or outlier or overflow" > $O/tests_tail.txt 2>&1; echo "rc $?" >> $O/tests_tail.txt
for b in 8 4; do AB_B=$b timeout 300 python tools/ab_inproc.py 2>&1 | tail -1; done > $O/ab.txt
timeout 200 python tools/gemm_tiles.py shapes 0 > $O/gemm_shapes.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "pytest rc $?" >> $O/tests.txt
tail -3 $O/tests_tail.txt; cat $O/ab.txt; grep "head\|786432" $O/gemm_shapes.txt; tail -3 $O/tests.txt
