#!/bin/bash
# Long runs of the randomised route fuzzers on both builds of the library (shipped: arguments only; developer: + tuning knobs).
# usage (GPU box): bash tools/soak_all.sh [trials]      -> gpurun_out/soak_*.txt
T=${1:-1000}
for L in "" "$PWD/diffqcqp_amd/lib/tuning/libdiffqcqp_hip.so"; do
  tag=$([ -z "$L" ] && echo shipped || echo tuning)
  for job in "fuzz_small.py $T 101" "fuzz_small.py $T 102 big" "fuzz_bwd.py $T 103" "fuzz_bwd.py $T 104 big" "fuzz_bwd.py $((T/4)) 105 lane" "soak_segmented.py $((T/5)) 106"; do
    echo "== [$tag] tools/$job"
    DQQ_LIB=$L python tools/$job 2>&1 | grep -v amdgpu.ids | tail -4
  done
done
