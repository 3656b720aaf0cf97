#!/bin/bash
# round 6, GPU call J: the tile table under the final cost model (profiles/r06_tile_table.txt), A/B at the mid sizes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6j; rm -rf $O; mkdir -p $O
timeout 1500 python tools/tile_table.py 3 > $O/tile_table.txt 2> $O/tile_table.err
for cfg in "8 384 512" "4 384 512" "2 384 512" "1 384 512" "8 224 224" "2 224 224"; do set -- $cfg; AB_B=$1 AB_H=$2 AB_W=$3 timeout 300 python tools/ab_inproc.py 2>&1 | tail -1 | sed "s/^/B=$1 @$2x$3: /"; done > $O/ab.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "pytest rc $?" >> $O/tests.txt
cat $O/ab.txt; grep "head" $O/tile_table.txt; grep "^# B" $O/tile_table.txt | cut -c1-150; tail -3 $O/tests.txt
