#!/bin/bash
# Runs on the GPU box (through gpurun): the measurements DESIGN.md section 5 / 6 quote, into gpurun_out/$TAG/ for copying into profiles/.
#   tools/round_measurements.sh TAG
set -u
TAG=${1:-meas}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
B="python bench.py"
timeout 900 $B > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 300 $B --steps 20 --warmup 5 --no-latency-leg --sweep-agents 0 > $OUT/bench_driver_style_20steps.json 2>> $OUT/bench_default.err      # what the driver runs
timeout 300 $B --start-tick 21 --steps 190 --no-cpu-baseline --no-latency-leg --sweep-agents 0 > $OUT/bench_whole_mission_window.json 2>> $OUT/bench_default.err
timeout 300 $B --reset-threshold 0 --no-cpu-baseline --no-latency-leg --sweep-agents 0 > $OUT/bench_no_checks.json 2>> $OUT/bench_default.err
timeout 300 $B --unfused --no-cpu-baseline --no-latency-leg --sweep-agents 0 > $OUT/bench_unfused.json 2>> $OUT/bench_default.err
for mode in "--planner bvc" "--planner bvc --slack collision_constraint" "--planner bvc --slack dynamical_limit"; do
  timeout 300 $B $mode --no-cpu-baseline --no-latency-leg --sweep-agents 0 2>> $OUT/bench_default.err
done > $OUT/bench_modes.jsonl
timeout 600 $B --workload random1024 --steps 60 --warmup 10 > $OUT/bench_random1024.json 2>> $OUT/bench_default.err
timeout 600 $B --workload forest256 --steps 60 --warmup 10 > $OUT/bench_forest256_prior_based.json 2>> $OUT/bench_default.err
timeout 600 $B --workload forest256 --static-goal --steps 60 --warmup 10 > $OUT/bench_forest256_static.json 2>> $OUT/bench_default.err
timeout 900 python tools/config_runs.py > $OUT/config_runs.jsonl 2>> $OUT/bench_default.err
timeout 600 python tools/shard_emulation.py --workload random1024 > $OUT/shard_emulation_random1024.jsonl 2>> $OUT/bench_default.err
timeout 600 python tools/shard_emulation.py --workload forest256 > $OUT/shard_emulation_forest256.jsonl 2>> $OUT/bench_default.err
# round 6: swarms beyond one GPU's 1024 agents at random1024's density, with the neighbour lists and (LSC_NO_NEIGHBOUR_LISTS) with round 5's walks
for n in 2048 4096 8192; do
  timeout 600 python tools/shard_emulation.py --agents $n --shards 1,8 --ticks 40 2>> $OUT/bench_default.err
  LSC_NO_NEIGHBOUR_LISTS=1 timeout 600 python tools/shard_emulation.py --agents $n --shards 1,8 --ticks 40 2>> $OUT/bench_default.err
done > $OUT/shard_emulation_weak.jsonl
LSC_NEIGH_ALWAYS=1 timeout 600 python tools/shard_emulation.py --workload random1024 > $OUT/shard_emulation_random1024_lists_forced.jsonl 2>> $OUT/bench_default.err
LSC_NEIGH_PROFILE=1 timeout 300 python bench.py --workload random1024 --steps 30 --warmup 10 --no-cpu-baseline --no-latency-leg --sweep-agents 0 2>&1 > /dev/null | grep "\[lsc\]" > $OUT/neighbour_query_stages.log
timeout 600 python tools/phase_profile.py --random --agents 1024 --ticks 40 --from-tick 11 > $OUT/phase_profile_1024.log 2>> $OUT/bench_default.err
LSC_NO_NEIGHBOUR_LISTS=1 timeout 600 python tools/phase_profile.py --random --agents 1024 --ticks 40 --from-tick 11 > $OUT/phase_profile_1024_walks.log 2>> $OUT/bench_default.err
timeout 300 python bench.py --mission-list --steps 100 --no-cpu-baseline > $OUT/bench_mission_list.json 2>> $OUT/bench_default.err
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_atomics tools/microbench/launch_atomics.hip 2>/dev/null && timeout 60 /tmp/launch_atomics > $OUT/microbench_launch_atomics.log 2>&1
timeout 600 python tools/general_profile.py > $OUT/general_profile.jsonl 2>> $OUT/bench_default.err
timeout 600 python tools/phase_profile.py > $OUT/phase_profile_64.log 2>> $OUT/bench_default.err
timeout 600 python tools/goal_profile.py > $OUT/goal_profile.log 2>> $OUT/bench_default.err
timeout 600 python tools/multi_eval.py > $OUT/multi_eval.log 2>> $OUT/bench_default.err
ls -la $OUT
