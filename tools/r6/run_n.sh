#!/bin/bash
# round 6, GPU call N: mlp.fc1 epilogue on transposed accumulators - kernel tests, goldens, A/B, stamps
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6n; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" > $O/tests_kernels.txt 2>&1; echo "rc $?" >> $O/tests_kernels.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "pytest rc $?" >> $O/tests.txt
for b in 8 2; do AB_B=$b timeout 300 python tools/ab_inproc.py 2>&1 | tail -1; done > $O/ab.txt
AB_B=8 AB_H=224 AB_W=224 timeout 300 python tools/ab_inproc.py 2>&1 | tail -1 >> $O/ab.txt
timeout 200 python tools/model_stamps.py 2>&1 | grep -v "^/opt" | head -14 > $O/stamps.txt
tail -2 $O/tests_kernels.txt; tail -2 $O/tests.txt; cat $O/ab.txt; cat $O/stamps.txt
