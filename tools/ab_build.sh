#!/bin/bash
# A/B on ONE box (boxes of the pool differ by several per cent): the shipped build (A) against builds with
# DQQ_EXTRA_FLAGS (B, C, ...), bench config $1, alternating.   gpurun -- 'bash tools/ab_build.sh 4 -DX=1 -DX=2'
set -u
cd $GRAFT_REPO_ROOT
L=diffqcqp_amd/lib
CFG=$1; shift
cp $L/libdiffqcqp_hip.so /tmp/lib_0.so
i=0
for f in "$@"; do
  i=$((i+1))
  DQQ_EXTRA_FLAGS="$f" python -m diffqcqp_amd.build > /tmp/build_$i.log 2>&1 || tail -5 /tmp/build_$i.log
  cp $L/libdiffqcqp_hip.so /tmp/lib_$i.so
  LAST="$f"
done
export DQQ_EXTRA_FLAGS="$LAST"
for r in 1 2 3; do
  for v in $(seq 0 $i); do
    cp /tmp/lib_$v.so $L/libdiffqcqp_hip.so
    python bench.py --config $CFG --no-cpu-baseline --no-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('build $v', round(d['ms_per_step'],4), {k:round(v['mean_us'],1) for k,v in d['kernels'].items()})"
  done
done
