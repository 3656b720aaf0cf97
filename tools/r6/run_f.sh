#!/bin/bash
# round 6, GPU call F: LayerNorm fold after the consumer's loads were batched - tests, switch A/B, shapes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6f; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "layernorm_fold or full_size" > $O/tests_fold.txt 2>&1; echo "rc $?" >> $O/tests_fold.txt
timeout 300 python tools/ab_option.py 3 1 0 --rounds 4 > $O/ab_opt3.txt 2>&1
timeout 200 python tools/gemm_tiles.py shapes 0 > $O/gemm_shapes.txt 2>&1
tail -3 $O/tests_fold.txt; tail -3 $O/ab_opt3.txt; head -8 $O/gemm_shapes.txt
