#!/bin/bash
# persistent fused tail (NEW) against the build one commit earlier (libsta_prev.so: one tile per workgroup, 99 VGPRs, no spills)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6q; rm -rf $O; mkdir -p $O
for cfg in "8 384 512" "4 384 512" "8 224 224"; do set -- $cfg; AB_B=$1 AB_H=$2 AB_W=$3 timeout 400 python tools/ab_inproc.py tools/ab/libsta_prev.so f16x3h 5 2>&1 | tail -1 | sed "s/^/B=$1 @$2x$3: /"; done > $O/ab_prev.txt
cat $O/ab_prev.txt
