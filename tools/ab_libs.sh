#!/bin/bash
# Same-box A/B of two builds of the library: NEW = vista_slam_amd/libsta_mi355.so, OLD = tools/ab/libsta_old.so
# (kept out of git).  Runs new, old, new so box drift shows.   usage: tools/ab_libs.sh [model_ab.py configs...]
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cfg="${@:-f16x3h:0:0}"
cp vista_slam_amd/libsta_mi355.so /tmp/new.so
timeout 150 python tools/model_ab.py $cfg 2>&1 | grep "pass 1" | sed 's/^/NEW /'
cp tools/ab/libsta_old.so vista_slam_amd/libsta_mi355.so
timeout 150 python tools/model_ab.py $cfg 2>&1 | grep "pass 1" | sed 's/^/OLD /'
cp /tmp/new.so vista_slam_amd/libsta_mi355.so
timeout 150 python tools/model_ab.py $cfg 2>&1 | grep "pass 1" | sed 's/^/NEW /'
