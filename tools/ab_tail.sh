#!/bin/bash
# A/B of builds on the tail-heavy workloads (the reference's figure workload, the stress variant) next to the headline:
#   tools/ab_tail.sh name=lib.so [name=lib.so ...]     (GPU box; alternating, 3 rounds)  -> gpurun_out/ab_tail.txt
R=$PWD; OUT=$R/gpurun_out/ab_tail.txt; mkdir -p $R/gpurun_out; : > $OUT
for rep in 1 2 3; do for nl in "$@"; do name=${nl%%=*}; lib=${nl#*=}
  DQQ_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check --details /tmp/ab_tail.json > /dev/null 2>&1
  python -c "
import json; d=json.load(open('/tmp/ab_tail.json')); e=d['survey_8d_extras']; f=e['reference_figure_workload']
print('rep $rep %-9s' % '$name', 'figure qp %.3f qcqp %.3f | bench-eps qp %.3f qcqp %.3f | stress %.4f | headline cold %.5f hot %.5f qp_pair %.5f large %.4f' % (f['qp_fwd_ms'], f['qcqp_fwd_ms'], f['qp_fwd_ms_eps1e-7_maxiter1000'], f['qcqp_fwd_ms_eps1e-7_maxiter1000'], e['stress_p_u(0,1)_qp_fwd']['ms_per_call'], d['ms_per_step'], d['hot']['ms_per_step'], d['qp_pair']['ms_per_step'], d['qp_pair_large']['ms_per_step']))" >> $OUT
done; done; cat $OUT
