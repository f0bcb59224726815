#!/bin/bash
# Runs on the GPU box (through gpurun): the short loop behind a kernel change -- GPU suite, headline, phase profile, 1024-agent swarm.
#   tools/quick_ab.sh TAG [notests]
TAG=${1:-ab}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
if [ "${2:-}" != "notests" ]; then
  (timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $OUT/gpu_tests.log
fi
B="python bench.py --no-cpu-baseline --no-latency-leg --sweep-agents 0 --missions 0 --no-ip-leg"
for i in 1 2; do timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d[k] for k in ('value','ms_per_step','tick_solve_ms')}))" | cut -c1-400; done > $OUT/headline.jsonl
timeout 300 python tools/phase_profile.py > $OUT/phase_profile_64.log 2>/dev/null
timeout 300 python tools/config_runs.py --only random1024,circle20 2>/dev/null | cut -c1-330 > $OUT/config_runs.jsonl
tail -3 $OUT/gpu_tests.log 2>/dev/null; cat $OUT/headline.jsonl; grep -v "0.00 us" $OUT/phase_profile_64.log; cat $OUT/config_runs.jsonl
