#!/bin/bash
# Same-box A/B of two library builds at the SLAM scale (B = 1 @224x224 split entry points + 5-edge scheduler):
# NEW = vista_slam_amd/libsta_mi355.so, OLD = tools/ab/libsta_old.so.
set -u
cd "$(dirname "$0")/.."
run() { timeout 200 python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench, torch
from vista_slam_amd import weights as W
from vista_slam_amd.sta_frontend import STAFrontend
m = STAFrontend(W.FULL, "cuda:0").load_procedural(seed=43)
for _ in range(2):
    r = bench.slam_probe(m, "cuda:0")
print({k: v for k, v in r.items() if k != "note"})
PY
}
cp vista_slam_amd/libsta_mi355.so /tmp/new.so
echo NEW; run 2>&1 | tail -1
cp tools/ab/libsta_old.so vista_slam_amd/libsta_mi355.so
echo OLD; run 2>&1 | tail -1
cp /tmp/new.so vista_slam_amd/libsta_mi355.so
echo NEW; run 2>&1 | tail -1
