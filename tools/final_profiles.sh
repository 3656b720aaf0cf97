R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/profile_round.sh > $R/gpurun_out/profile_round.log 2>&1
for w in encode sched; do
  bash $R/tools/slam_prof.sh $w _fin 20 > $R/gpurun_out/slam_${w}_percall.txt 2>&1
  db=$(ls $R/gpurun_out/slam_${w}_fin/*/*.db $R/gpurun_out/slam_${w}_fin/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_stats.py --tail 2000 $db > $R/gpurun_out/round/slam_${w}_timeline.txt 2>&1
  rm -rf $R/gpurun_out/slam_${w}_fin
done
cd $R
timeout 600 python tools/prec_check.py f16x3h f16x3 > gpurun_out/round/precision_table.txt 2>&1
timeout 600 python tools/prec_check.py --stress f16x3h f16x3 >> gpurun_out/round/precision_table.txt 2>&1
timeout 600 python tools/prec_check.py --outlier f16x3h f16x3 >> gpurun_out/round/precision_table.txt 2>&1
timeout 600 python tools/prec_check.py --fullstress f16x3h f16x3 >> gpurun_out/round/precision_table.txt 2>&1
timeout 300 python tools/ab_slam_libs.py > gpurun_out/round/ab_vs_round_start_slam.txt 2>&1
for b in 8 4 2 1; do AB_B=$b timeout 200 python tools/ab_inproc.py 2>&1 | tail -1; done > gpurun_out/round/ab_vs_round_start.txt
tail -5 gpurun_out/profile_round.log
