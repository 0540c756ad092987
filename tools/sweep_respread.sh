#!/bin/bash
# Developer build: the re-spread thresholds of the N = 8 forward at the bench shape (round 6: (16, 0) read 0.6 us faster than
# (16, 8) on the well-conditioned headline, hence the gate fwd_respread2_from).  usage: bash tools/sweep_respread.sh
R=$PWD; export DQQ_LIB=$R/diffqcqp_amd/lib/tuning/libdiffqcqp_hip.so
for rep in 1 2; do for rs in "16 8" "12 8" "8 8" "16 4" "12 6" "16 0" "0 0"; do set -- $rs; for cfg in 2 3; do
  python tools/bench_opt.py fwd_respread=$1 fwd_respread2=$2 fwd_respread2_from=0 -- --config $cfg --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline --no-check --no-hot --details /tmp/sw.json > /dev/null 2>&1
  python -c "import json; d=json.load(open('/tmp/sw.json')); print('rep $rep respread $1 $2 config $cfg:', {k: round(v['mean_us'],2) for k,v in d['kernels'].items()})"
done; done; done
