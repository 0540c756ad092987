#!/bin/bash
# A/B of builds of the C ABI on ONE box through bench.py: tools/ab_libs_bench.sh "<bench args>" name=lib.so [name=lib.so ...]
# (alternating, three rounds; per-launch durations and the step).  Output: gpurun_out/ab_libs_bench.txt
R=$PWD
ARGS=$1; shift
OUT=$R/gpurun_out/ab_libs_bench.txt
mkdir -p $R/gpurun_out; : > $OUT
for rep in 1 2 3; do
  for nl in "$@"; do
    name=${nl%%=*}; lib=${nl#*=}
    DQQ_LIB=$lib python bench.py $ARGS --no-cpu-baseline --no-hot --details /tmp/ab_lb.json > /dev/null 2>/tmp/ab_lb.err || tail -3 /tmp/ab_lb.err >> $OUT
    python -c "import json; d=json.load(open('/tmp/ab_lb.json')); print('rep $rep %-10s' % '$name', 'ms_per_step %.5f' % d['ms_per_step'], {k: round(v['mean_us'],2) for k,v in d['kernels'].items()}, d.get('parity_max_abs_err_vs_oracle_sample',''))" >> $OUT
  done
done
cat $OUT
