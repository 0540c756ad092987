#!/bin/bash
# The same question for the N = 8 diagonal forwards (the headline's dominant kernels): does a scalar instruction in the ADMM
# loop cost time?  Variants with K extra scalar instructions per iteration (-DDQQ_SALU_PROBE_DIAG=K, fwd_diag.hip only).
#   build: tools/ab_salu_probe_diag.sh build      run (GPU box): tools/ab_salu_probe_diag.sh
R=$PWD
V=$R/diffqcqp_amd/lib/variants
if [ "$1" = "build" ]; then
  for K in 8 16; do
    mkdir -p $V/dsalu$K
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fvisibility=hidden -Wall -Wno-unused-function \
      -ffp-contract=fast-honor-pragmas -DDQQ_SALU_PROBE_DIAG=$K -I $R/include -c $R/diffqcqp_amd/csrc/fwd_diag.hip -o $V/dsalu$K/fwd_diag.o || exit 1
    OBJS=$(ls $R/diffqcqp_amd/lib/obj/*.o | grep -v fwd_diag.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/dsalu$K/libdiffqcqp_hip.so $OBJS $V/dsalu$K/fwd_diag.o || exit 1
  done
  ls -la $V/dsalu*/libdiffqcqp_hip.so; exit 0
fi
OUT=$R/gpurun_out/ab_salu_probe_diag.txt
mkdir -p $R/gpurun_out; : > $OUT
for rep in 1 2 3; do
  for lib in $R/diffqcqp_amd/lib/libdiffqcqp_hip.so $V/dsalu8/libdiffqcqp_hip.so $V/dsalu16/libdiffqcqp_hip.so; do
    DQQ_LIB=$lib python bench.py --config 0 --streams 1 --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline --no-check --no-hot --details /tmp/ab_dsalu.json > /dev/null 2>&1
    python -c "import json; d=json.load(open('/tmp/ab_dsalu.json')); print('rep $rep', '$lib'.split('/')[-2], 'ms_per_step %.5f' % d['ms_per_step'], {k: round(v['mean_us'],2) for k,v in d['kernels'].items()})" >> $OUT
  done
done
cat $OUT
