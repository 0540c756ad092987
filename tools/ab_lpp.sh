#!/bin/bash
# Developer build: lanes per problem of the N = 8 diagonal forward at the bench batch (the shipped rule: 4 below B = 57344, 2 from
# there on -- settled in round 3, re-checked in round 6 after the loop bodies changed).  -> gpurun_out/ab_lpp.txt
R=$PWD; export DQQ_LIB=$R/diffqcqp_amd/lib/tuning/libdiffqcqp_hip.so
OUT=$R/gpurun_out/ab_lpp.txt; mkdir -p $R/gpurun_out; : > $OUT
for rep in 1 2 3; do for lpp in 2 4; do for cfg in 2 3 0; do
  python tools/bench_opt.py fwd_lpp=$lpp -- --config $cfg --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline --no-check --no-hot --details /tmp/ab_lpp.json > /dev/null 2>&1
  python -c "import json; d=json.load(open('/tmp/ab_lpp.json')); print('rep $rep lpp $lpp config $cfg: ms_per_step %.5f' % d['ms_per_step'], {k: round(v['mean_us'],2) for k,v in d['kernels'].items()})" >> $OUT
done; done; done; cat $OUT
