#!/bin/bash
# round 6, GPU call C: compile-time ReLU / running weight pointer in the convolution loops - A/B against the round-start build; LDS counters
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6c; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" > $O/tests_conv.txt 2>&1; echo "rc $?" >> $O/tests_conv.txt
for b in 8 2; do AB_B=$b timeout 300 python tools/ab_inproc.py 2>&1 | tail -1; done > $O/ab.txt
timeout 200 python tools/gemm_tiles.py shapes 0 > $O/gemm_shapes.txt 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 250 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_lds -o t -- python $R/tools/model_steps.py 0 3 f16x3h > $O/pmc_lds.out 2> $O/pmc_lds.err
F=$(ls $O/pmc_lds/*/*counter_collection.csv $O/pmc_lds/*counter_collection.csv 2>/dev/null | head -1)
python $R/tools/pmc_lds_summary.py $F > $O/lds_util_summary.txt 2>&1
rm -rf $O/pmc_lds
cd $R; tail -2 $O/tests_conv.txt; cat $O/ab.txt; grep -i "attn\|conv3h" $O/lds_util_summary.txt; grep "conv" $O/gemm_shapes.txt
