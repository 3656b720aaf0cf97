#!/bin/bash
# round 6, GPU call A: the suite on the hidden-visibility build, A/B against the round-start build, attention's LDS conflicts
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "pytest rc $?" >> $O/tests.txt
AB_B=8 timeout 300 python tools/ab_inproc.py > $O/ab8.txt 2>&1
timeout 200 python tools/attn_bench.py > $O/attn_bench.txt 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 250 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_lds -o t -- python $R/tools/model_steps.py 0 3 f16x3h > /dev/null 2> $O/pmc_lds.err
F=$(ls $O/pmc_lds/*/*counter_collection.csv $O/pmc_lds/*counter_collection.csv 2>/dev/null | head -1)
python $R/tools/pmc_lds_summary.py $F > $O/lds_util_summary.txt 2>&1
rm -rf $O/pmc_lds
tail -4 $O/tests.txt; tail -2 $O/ab8.txt; head -3 $O/attn_bench.txt; head -6 $O/lds_util_summary.txt
