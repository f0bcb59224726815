#!/bin/bash
# Large-count fuzz / stress runs on the GPU box (through gpurun): the generators of tests/test_gpu_fuzz.py at 10-100x their counts,
# new seeds every round.  They use the oracle (test infrastructure), hence tests/fuzz_*.py.   tools/fuzz_round.sh TAG SEEDBASE
set -u
TAG=${1:-fuzz}
S=${2:-3000000}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
{
echo "== tests/fuzz_lsc.py $((S+100000)) 1000"; timeout 1500 python tests/fuzz_lsc.py $((S+100000)) 1000 2>&1 | tail -1
echo "== tests/fuzz_lsc.py $((S+200000)) 300 45"; timeout 1500 python tests/fuzz_lsc.py $((S+200000)) 300 45 2>&1 | tail -1
echo "== tests/fuzz_lsc.py $((S+300000)) 300 14 {warm_start_mu: 0}"; timeout 1500 python tests/fuzz_lsc.py $((S+300000)) 300 14 '{"warm_start_mu": 0}' 2>&1 | tail -1
echo "== tests/fuzz_modes.py $((S+400000)) 1000"; timeout 2400 python tests/fuzz_modes.py $((S+400000)) 1000 2>&1 | tail -1
echo "== tests/fuzz_variants.py $((S+500000)) 300 all"; timeout 2400 python tests/fuzz_variants.py $((S+500000)) 300 all 2>&1 | grep -v "^MISMATCH" | tail -3;
echo "== tools/fuzz_device_chain.py"; timeout 900 python tools/fuzz_device_chain.py 2>&1 | tail -1
echo "== tools/stress_missions.py"; timeout 900 python tools/stress_missions.py 2>&1 | tail -1
} > $OUT/fuzz_stress.log 2>&1
cat $OUT/fuzz_stress.log
