#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6l; rm -rf $O; mkdir -p $O
timeout 1500 python tools/tile_table.py 3 > $O/tile_table.txt 2> $O/tile_table.err
grep -c "|" $O/tile_table.txt; grep "head" $O/tile_table.txt | cut -c1-200
