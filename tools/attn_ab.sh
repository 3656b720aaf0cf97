#!/bin/bash
# attention kernel alone, NEW (libsta_mi355.so) vs OLD (libsta_old.so): rocprofv3 durations
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { rm -rf /tmp/ab_$1; timeout 120 rocprofv3 --kernel-trace -d /tmp/ab_$1 -o t -- python $R/tools/attn_one.py 16 16 768 768 20 > /tmp/ab_$1.log 2>&1
  echo "$1: $(python $R/tools/rocpd_stats.py $(find /tmp/ab_$1 -name '*.db' | head -1) 2>&1 | grep -i 'attn' | cut -c105-175)"; }
cp $R/vista_slam_amd/libsta_mi355.so /tmp/new.so
one NEW; cp $R/tools/ab/libsta_old.so $R/vista_slam_amd/libsta_mi355.so; one OLD; cp /tmp/new.so $R/vista_slam_amd/libsta_mi355.so; one NEW; 
cp $R/tools/ab/libsta_old.so $R/vista_slam_amd/libsta_mi355.so; one OLD; cp /tmp/new.so $R/vista_slam_amd/libsta_mi355.so
