#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel stats and PMC passes of the bench command and of the octomap
# configuration, summarised into gpurun_out/$TAG/ for copying into profiles/.
#   tools/profile_round.sh TAG
set -u
TAG=${1:-prof}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
BENCH="python $ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-latency-leg --missions 0 --no-ip-leg"      # default --start-tick 60: timed launches = dispatches 59 .. 158 of the plan kernel
FOREST="python $ROOT/tools/config_runs.py --only forest256p,forest256,forest256x4p,forest256x4 --ticks 30 --warmup 5"
LARGE="python $ROOT/tools/config_runs.py --only random1024 --ticks 30 --warmup 5"
GENERAL="python $ROOT/tools/general_profile.py --modes bvc,collision_constraint,gust"      # lsc_general_kernel under load
cd /tmp
# 1. kernel trace + stats (no counters)
rocprofv3 --kernel-trace --stats -d $OUT/stats_bench -o bench -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/stats_bench.err
rocprofv3 --kernel-trace --stats -d $OUT/stats_forest -o forest -- $FOREST > $OUT/forest_under_rocprof.jsonl 2> $OUT/stats_forest.err
rocprofv3 --kernel-trace --stats -d $OUT/stats_large -o large -- $LARGE > $OUT/large_under_rocprof.jsonl 2> $OUT/stats_large.err
rocprofv3 --kernel-trace --stats -d $OUT/stats_general -o general -- $GENERAL --ticks 30 > $OUT/general_under_rocprof.jsonl 2> $OUT/stats_general.err
# 2. counters, each group in its own pass, kernel trace only
# (fourth pass, round 5: the LDS side north_star asks for -- SQ_LDS_IDX_ACTIVE = all LDS-array cycles, SQ_LDS_BANK_CONFLICT = the extra
#  cycles of bank conflicts, SQ_ACTIVE_INST_LDS / SQ_WAIT_INST_LDS = wave-cycles on / waiting for LDS instructions; MI355X_MICROARCH.md, LDS section)
# (fifth and sixth pass, round 6: the instruction stream -- requests / hits / misses of the instruction cache the CUs share in pairs, fetches
#  and the requests that reach the L2: what a 99 KB kernel executed nearly straight-line pays for its text)
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_ANY SQC_TC_INST_REQ SQC_TC_REQ"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/pmc_bench_$tag -o bench -- $BENCH --steps 30 > /dev/null 2> $OUT/pmc_bench_$tag.err
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/pmc_forest_$tag -o forest -- $FOREST --ticks 10 > /dev/null 2> $OUT/pmc_forest_$tag.err
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/pmc_large_$tag -o large -- $LARGE --ticks 10 > /dev/null 2> $OUT/pmc_large_$tag.err
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/pmc_general_$tag -o general -- $GENERAL --ticks 10 > /dev/null 2> $OUT/pmc_general_$tag.err
done
cd $ROOT
for d in stats_bench stats_forest stats_large stats_general; do
  db=$(find $OUT/$d -name "*.db" | head -1)
  if [ "$d" = stats_bench ]; then win="lsc_plan_alt_kernel 59 100"; else win=""; fi
  [ -n "$db" ] && python profiles/summarize_rocpd.py stats $db $win > $OUT/$d.csv
done
python profiles/summarize_rocpd.py pmc $(find $OUT/pmc_bench_* -name "*.db") > $OUT/pmc_bench.json
python profiles/summarize_rocpd.py pmc $(find $OUT/pmc_forest_* -name "*.db") > $OUT/pmc_forest.json
python profiles/summarize_rocpd.py pmc $(find $OUT/pmc_large_* -name "*.db") > $OUT/pmc_large.json
python profiles/summarize_rocpd.py pmc $(find $OUT/pmc_general_* -name "*.db") > $OUT/pmc_general.json
# the databases are large: keep only the summaries
find $OUT -name "*.db" -delete
ls -la $OUT
