#!/bin/bash
# round 6, GPU call U: Cout = 256 halo convolutions as two 128-column two-tap tiles (switch 0 = 3) against the 256-column one-tap tile
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6u; rm -rf $O; mkdir -p $O
timeout 400 python tools/ab_option.py 0 0 3 --rounds 4 > $O/ab_b8.txt 2>&1
AB_B=2 timeout 400 python tools/ab_option.py 0 0 3 --rounds 4 > $O/ab_b2.txt 2>&1
tail -2 $O/ab_b8.txt; tail -2 $O/ab_b2.txt
