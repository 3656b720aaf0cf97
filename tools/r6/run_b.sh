#!/bin/bash
# round 6, GPU call B: mlp.fc2 in f16mx (precision f16x3m): kernel tests, precision table, A/B; new sequence goldens; LDS counters again
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6b; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "mlp_f16mx" > $O/tests_kernels.txt 2>&1; echo "rc $?" >> $O/tests_kernels.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "t075 or sharp or checkpoint_kit or reserve or pipeline_streams" > $O/tests_new.txt 2>&1; echo "rc $?" >> $O/tests_new.txt
timeout 300 python tools/model_ab.py f16x3h:0 f16x3m:0 > $O/ab_prec.txt 2>&1
for fam in "" --stress --outlier --fullstress; do timeout 600 python tools/prec_check.py $fam f16x3m f16x3h >> $O/precision_table.txt 2>&1; done
cd /tmp; export TMPDIR=/tmp
timeout 250 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_lds -o t -- python $R/tools/model_steps.py 0 3 f16x3h > $O/pmc_lds.out 2> $O/pmc_lds.err
F=$(ls $O/pmc_lds/*/*counter_collection.csv $O/pmc_lds/*counter_collection.csv 2>/dev/null | head -1)
python $R/tools/pmc_lds_summary.py $F > $O/lds_util_summary.txt 2>&1
wc -l $F > $O/pmc_csv_info.txt; cut -d, -f1-12 $F | grep -i "attn" | head -5 >> $O/pmc_csv_info.txt; head -2 $F >> $O/pmc_csv_info.txt
rm -rf $O/pmc_lds
cd $R
tail -3 $O/tests_kernels.txt; tail -3 $O/tests_new.txt; cat $O/ab_prec.txt | tail -4; grep -c ok $O/precision_table.txt; grep FAIL $O/precision_table.txt | head
