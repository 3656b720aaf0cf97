#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6m; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "pytest rc $?" >> $O/tests.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -q -s -k "batch8_stress" 2>&1 | grep "b8 stress" > $O/b8_stress.txt
tail -3 $O/tests.txt; cat $O/b8_stress.txt
bash tools/final_profiles.sh
