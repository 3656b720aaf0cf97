#!/bin/bash
# Everything profiles/rNN_* is made of, on ONE box:  bash tools/profile_round.sh   (run through gpurun; ~3 GPU-minutes)
#   bench line; rocprofv3 kernel trace of `python bench.py` (the same command) and of 4 plain steps; per-shape GEMM table;
#   three separate --pmc passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) of tools/model_steps.py.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 280 python $R/bench.py > $O/bench_1gpu.json 2> $O/bench.err
timeout 280 rocprofv3 --kernel-trace -d $O/trace_bench -o t -- python $R/bench.py --no-cpu-baseline --no-slam-probe --slam-frames 0 > $O/bench_traced.json 2> $O/trace_bench.err
timeout 200 rocprofv3 --kernel-trace -d $O/trace_steps -o t -- python $R/tools/model_steps.py 0 6 f16x3h > /dev/null 2> $O/trace_steps.err
timeout 200 python $R/tools/gemm_tiles.py shapes 0 > $O/gemm_shapes.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  d=$O/pmc_$(echo $c | cut -d' ' -f1)
  timeout 250 rocprofv3 --pmc $c --output-format csv -d $d -o t -- python $R/tools/model_steps.py 0 3 f16x3h > /dev/null 2> $d.err
done
python $R/tools/rocpd_stats.py $O/trace_bench/*/*.db > $O/kernel_stats.txt 2>&1 || python $R/tools/rocpd_stats.py $O/trace_bench/*.db > $O/kernel_stats.txt
python $R/tools/rocpd_stats.py --by-grid $O/trace_bench/*/*.db > $O/kernel_stats_by_grid.txt 2>&1 || python $R/tools/rocpd_stats.py --by-grid $O/trace_bench/*.db > $O/kernel_stats_by_grid.txt
python $R/tools/rocpd_stats.py $O/trace_steps/*/*.db > $O/kernel_stats_model_steps.txt 2>&1 || python $R/tools/rocpd_stats.py $O/trace_steps/*.db > $O/kernel_stats_model_steps.txt
F=$(ls $O/pmc_FETCH_SIZE/*/*counter_collection.csv $O/pmc_FETCH_SIZE/*counter_collection.csv 2>/dev/null | head -1)
W=$(ls $O/pmc_WRITE_SIZE/*/*counter_collection.csv $O/pmc_WRITE_SIZE/*counter_collection.csv 2>/dev/null | head -1)
M=$(ls $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES/*/*counter_collection.csv $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES/*counter_collection.csv 2>/dev/null | head -1)
python $R/tools/pmc_summary.py $F $W > $O/pmc_traffic.json
python $R/tools/pmc_util_summary.py $M $O/pmc_traffic.json $O/kernel_stats_model_steps.txt > $O/mfma_hbm_summary.txt
L=$(ls $O/pmc_SQ_LDS_IDX_ACTIVE/*/*counter_collection.csv $O/pmc_SQ_LDS_IDX_ACTIVE/*counter_collection.csv 2>/dev/null | head -1)
python $R/tools/pmc_lds_summary.py $L $O/kernel_stats_model_steps.txt > $O/lds_util_summary.txt 2>&1
rm -rf $O/trace_bench $O/trace_steps $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES $O/pmc_SQ_LDS_IDX_ACTIVE
ls -la $O; head -c 300 $O/bench_1gpu.json; echo; head -8 $O/mfma_hbm_summary.txt
