#!/bin/bash
# round-5 GPU pass A: full GPU suite, boundary probe, bench line, same-box A/B against the round-start build, precision tables
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 > gpurun_out/a_pytest.log; tail -5 gpurun_out/a_pytest.log
python tools/boundary_probe.py 1000 > gpurun_out/boundary_probe.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; tail -c 600 gpurun_out/a_bench.json
timeout 300 python tools/ab_inproc.py > gpurun_out/a_ab_headline.txt 2>&1; tail -2 gpurun_out/a_ab_headline.txt
timeout 300 python tools/ab_slam_libs.py > gpurun_out/a_ab_slam.txt 2>&1; tail -3 gpurun_out/a_ab_slam.txt
(python tools/prec_check.py f16x3h f16x3; python tools/prec_check.py --stress f16x3h f16x3; python tools/prec_check.py --outlier f16x3h; python tools/prec_check.py --fullstress f16x3h f16x3) > gpurun_out/a_precision_table.txt 2>&1
tail -3 gpurun_out/a_precision_table.txt
