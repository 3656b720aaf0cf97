#!/bin/bash
# round-5 GPU pass C: the 192x96 tile family (sta_debug_set_option 3: 1 = off) - goldens, per-shape table, same-process A/B
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
STA_DEBUG_OPT=3:1 timeout 300 python tools/gemm_tiles.py shapes 0 2>/dev/null | grep -E "  768 |sum of" > gpurun_out/c_shapes_off.txt
timeout 300 python tools/gemm_tiles.py shapes 0 2>/dev/null | grep -E "  768 |sum of" > gpurun_out/c_shapes_on.txt
echo "--- family 4 off"; cat gpurun_out/c_shapes_off.txt; echo "--- family 4 on"; cat gpurun_out/c_shapes_on.txt
timeout 400 python tools/ab_option.py 3 1 0 --rounds 4 2>&1 | tail -3 | tee gpurun_out/c_ab_b8.txt
AB_B=4 timeout 300 python tools/ab_option.py 3 1 0 --rounds 3 2>&1 | tail -2 | tee gpurun_out/c_ab_b4.txt
AB_B=2 timeout 300 python tools/ab_option.py 3 1 0 --rounds 3 2>&1 | tail -2 | tee gpurun_out/c_ab_b2.txt
