#!/bin/bash
# round 6, GPU call E: LayerNorm folded into the consuming GEMM (encoder) - tests, switch A/B (option 3: 1 = off), shapes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6e; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "layernorm_fold or full_size or mid_size" > $O/tests_fold.txt 2>&1; echo "rc $?" >> $O/tests_fold.txt
timeout 300 python tools/ab_option.py 3 1 0 > $O/ab_opt3.txt 2>&1
AB_B=4 timeout 300 python tools/ab_option.py 3 1 0 2>&1 | tail -2 > $O/ab_opt3_b4.txt
timeout 200 python tools/gemm_tiles.py shapes 0 > $O/gemm_shapes.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "pytest rc $?" >> $O/tests.txt
tail -5 $O/tests_fold.txt; tail -3 $O/ab_opt3.txt; cat $O/ab_opt3_b4.txt; head -8 $O/gemm_shapes.txt; tail -3 $O/tests.txt
