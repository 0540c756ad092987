#!/bin/bash
# Developer build: the iteration from which the second re-spread applies (fwd_respread2_from) -- headline-type and tail-heavy
# workloads side by side.  usage: bash tools/sweep_gate.sh   -> gpurun_out/sweep_gate.txt
R=$PWD; export DQQ_LIB=$R/diffqcqp_amd/lib/tuning/libdiffqcqp_hip.so
OUT=$R/gpurun_out/sweep_gate.txt; : > $OUT
for rep in 1 2 3; do for g in 0 24 32 40 48; do
  python tools/bench_opt.py fwd_respread2_from=$g -- --steps 20 --warmup 5 --no-cpu-baseline --no-check --details /tmp/sg.json > /dev/null 2>&1
  python -c "
import json; d=json.load(open('/tmp/sg.json')); e=d['survey_8d_extras']; f=e['reference_figure_workload']
print('rep $rep gate %2d' % $g, 'figure qp %.3f qcqp %.3f | bench-eps qp %.3f qcqp %.3f | stress %.4f | cold %.5f qp_pair %.5f cfg2 %.5f cfg3 %.5f' % (f['qp_fwd_ms'], f['qcqp_fwd_ms'], f['qp_fwd_ms_eps1e-7_maxiter1000'], f['qcqp_fwd_ms_eps1e-7_maxiter1000'], e['stress_p_u(0,1)_qp_fwd']['ms_per_call'], d['ms_per_step'], d['qp_pair']['ms_per_step'], d['per_config']['config_2']['ms_per_step'], d['per_config']['config_3']['ms_per_step']))" >> $OUT
done; done; cat $OUT
