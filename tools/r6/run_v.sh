#!/bin/bash
# round 6, GPU call V: what the driver runs at round end - smoke, the GPU suite, the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6v; rm -rf $O; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "rc $?" >> $O/smoke.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "pytest rc $?" >> $O/tests.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
tail -2 $O/smoke.txt; tail -2 $O/tests.txt; head -c 400 $O/bench.json; echo; tail -1 $O/bench.err
