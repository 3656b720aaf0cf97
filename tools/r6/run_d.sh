#!/bin/bash
# round 6, GPU call D: hybrid schedule of the decoder's N = 768 residual GEMMs - suite, switch A/B (option 3: 1 = off), A/B against round start
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6d; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "pytest rc $?" >> $O/tests.txt
timeout 300 python tools/ab_option.py 3 1 0 > $O/ab_opt3.txt 2>&1
AB_B=8 timeout 300 python tools/ab_inproc.py 2>&1 | tail -1 > $O/ab.txt
timeout 200 python tools/gemm_tiles.py shapes 0 > $O/gemm_shapes.txt 2>&1
tail -3 $O/tests.txt; tail -3 $O/ab_opt3.txt; cat $O/ab.txt; grep " 768 " $O/gemm_shapes.txt
