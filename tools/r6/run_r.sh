#!/bin/bash
# round 6, GPU call R: conv3h 128-column tiles with TWO taps per K step (half the barriers, 32-KiB weight stages)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6r; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "pytest rc $?" >> $O/tests.txt
for cfg in "8 384 512" "4 384 512" "8 224 224"; do set -- $cfg; AB_B=$1 AB_H=$2 AB_W=$3 timeout 400 python tools/ab_inproc.py tools/ab/libsta_prev.so f16x3h 4 2>&1 | tail -1 | sed "s/^/B=$1 @$2x$3: /"; done > $O/ab_prev.txt
timeout 200 python tools/gemm_tiles.py shapes 0 > $O/gemm_shapes.txt 2>&1
tail -2 $O/tests.txt; cat $O/ab_prev.txt; grep "conv\*" $O/gemm_shapes.txt
