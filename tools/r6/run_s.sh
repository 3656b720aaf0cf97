#!/bin/bash
# round 6, GPU call S: kernel-level A/B on ONE box - rocprofv3 kernel trace of 6 steps with the previous build, then with the new one, twice (ABAB)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6s; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for arm in old new; do
    L=""; [ $arm = old ] && L=$R/tools/ab/libsta_prev.so
    STA_AB_LIB=$L timeout 200 rocprofv3 --kernel-trace -d $O/tr_${arm}_$rep -o t -- python $R/tools/model_steps.py 0 6 f16x3h > /dev/null 2> $O/tr_${arm}_$rep.err
    python $R/tools/rocpd_stats.py $O/tr_${arm}_$rep/*/*.db > $O/stats_${arm}_$rep.txt 2>&1 || python $R/tools/rocpd_stats.py $O/tr_${arm}_$rep/*.db > $O/stats_${arm}_$rep.txt
    rm -rf $O/tr_${arm}_$rep
  done
done
grep -h "conv3h\|TOTAL" $O/stats_*.txt
