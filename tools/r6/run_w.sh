#!/bin/bash
# round 6, GPU call W: kernel trace of the mid-size regime (1 and 2 pairs @512x384)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6w; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in 1 2; do
  AB_B=$b timeout 200 rocprofv3 --kernel-trace -d $O/tr_$b -o t -- python $R/tools/model_steps.py 0 12 f16x3h > /dev/null 2> $O/tr_$b.err
  python $R/tools/rocpd_stats.py --by-grid $O/tr_$b/*/*.db > $O/stats_b$b.txt 2>&1 || python $R/tools/rocpd_stats.py --by-grid $O/tr_$b/*.db > $O/stats_b$b.txt
  python $R/tools/rocpd_stats.py --tail 400 $O/tr_$b/*/*.db > $O/tail_b$b.txt 2>&1 || python $R/tools/rocpd_stats.py --tail 400 $O/tr_$b/*.db > $O/tail_b$b.txt
  rm -rf $O/tr_$b
done
head -30 $O/stats_b1.txt
