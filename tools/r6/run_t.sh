#!/bin/bash
# round 6, GPU call T: sta_decode_pos (foreign positions) parity + the whole suite + A/B against the build before it
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6t; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "positions" > $O/tests_pos.txt 2>&1; echo "pytest rc $?" >> $O/tests_pos.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "pytest rc $?" >> $O/tests.txt
for cfg in "8 384 512" "2 384 512" "8 224 224"; do set -- $cfg; AB_B=$1 AB_H=$2 AB_W=$3 timeout 400 python tools/ab_inproc.py tools/ab/libsta_prev.so f16x3h 4 2>&1 | tail -1 | sed "s/^/B=$1 @$2x$3: /"; done > $O/ab_prev.txt
grep "worst\|passed\|failed\|rc" $O/tests_pos.txt; tail -3 $O/tests.txt; cat $O/ab_prev.txt
