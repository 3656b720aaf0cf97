#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6p; rm -rf $O; mkdir -p $O
timeout 400 python tools/ab_option.py 0 2 0 --rounds 5 > $O/ab_persist_b8.txt 2>&1
AB_B=4 timeout 400 python tools/ab_option.py 0 2 0 --rounds 4 > $O/ab_persist_b4.txt 2>&1
AB_B=8 AB_H=224 AB_W=224 timeout 400 python tools/ab_option.py 0 2 0 --rounds 4 > $O/ab_persist_b8_224.txt 2>&1
tail -2 $O/ab_persist_b8.txt; tail -2 $O/ab_persist_b4.txt; tail -2 $O/ab_persist_b8_224.txt
