#!/bin/bash
# round-5 GPU pass B: full GPU suite (seq lines kept), the cost of the dominant kernel's event pairs in the bench line
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests -m gpu -q -s > gpurun_out/b_pytest_full.log 2>&1; tail -4 gpurun_out/b_pytest_full.log
grep "^\[seq\]" gpurun_out/b_pytest_full.log > gpurun_out/b_seq_lines.txt; grep -c . gpurun_out/b_seq_lines.txt
grep -E "FAILED|ERROR|Error" gpurun_out/b_pytest_full.log | head -20
for e in 1 4 1 4 1000000; do
  timeout 300 python bench.py --no-cpu-baseline --no-slam-probe --timed-every $e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('timed-every $e', d['value'], d['ms_per_step'], d['roofline']['launches'] if d['roofline'] else None, d['roofline']['avg_launch_us'] if d['roofline'] else None)"
done | tee gpurun_out/b_timed_every.txt
