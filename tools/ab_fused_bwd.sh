#!/bin/bash
# Developer build only: what dropping the backward's drain launch would buy.  fuse_fallback = 1 makes
# bwd_diag_kernel solve non-diagonal tiles itself (FUSE = true: no drain launch behind it; its launch bounds cap the VGPRs);
# -1 is the shipped route (bwd_diag_kernel + the drain of the work-list).  Alternates the two on ONE box, three times:
# qp_pair (--config 8, one stream) and the two-stream headline (--config 0).  Output: gpurun_out/ab_fused_bwd.txt
R=$PWD
export DQQ_LIB=$R/diffqcqp_amd/lib/tuning/libdiffqcqp_hip.so
OUT=$R/gpurun_out/ab_fused_bwd.txt
mkdir -p $R/gpurun_out
: > $OUT
for rep in 1 2 3; do
  for opt in -1 1; do
    for cfg in 8 0; do
      python tools/bench_opt.py fuse_fallback=$opt -- --config $cfg --steps 100 --warmup 10 --repeats 5 --no-cpu-baseline --no-check --no-hot --details /tmp/ab_$cfg.json 2>/dev/null | \
        python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('rep $rep fuse_fallback=$opt config $cfg: ms_per_step %.5f' % d['ms_per_step'])" >> $OUT
      python -c "import json; d=json.load(open('/tmp/ab_$cfg.json')); print('      kernels_us', {k: round(v['mean_us'],2) for k,v in d['kernels'].items()})" >> $OUT
    done
  done
done
cat $OUT
