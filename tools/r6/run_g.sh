#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6g; rm -rf $O; mkdir -p $O
timeout 200 python tools/model_stamps.py > $O/stamps_fold_on.txt 2>&1
STA_TOOL_OPT=3:1 timeout 200 python tools/model_stamps.py > $O/stamps_fold_off.txt 2>&1
AB_B=8 timeout 300 python tools/ab_inproc.py 2>&1 | tail -1 > $O/ab.txt
head -8 $O/stamps_fold_on.txt; head -8 $O/stamps_fold_off.txt; cat $O/ab.txt
