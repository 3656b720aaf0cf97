#!/bin/bash
# round 6, GPU call I: transposed tail on the implicit-GEMM EPI_HEAD kernel too; fresh tile table
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6i; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "large_tile or full_size or mid_size or outlier or overflow or full_goldens or portrait" > $O/tests_tail.txt 2>&1; echo "rc $?" >> $O/tests_tail.txt
for b in 8 4 2; do AB_B=$b timeout 300 python tools/ab_inproc.py 2>&1 | tail -1; done > $O/ab.txt
AB_B=8 AB_H=224 AB_W=224 timeout 300 python tools/ab_inproc.py 2>&1 | tail -1 >> $O/ab.txt
timeout 1500 python tools/tile_table.py 3 > $O/tile_table.txt 2> $O/tile_table.err
tail -3 $O/tests_tail.txt; cat $O/ab.txt; grep "head" $O/tile_table.txt
