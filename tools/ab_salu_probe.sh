#!/bin/bash
# is the scalar instruction stream of fwd_dense_wave64_kernel (BASELINE configs[4]: 278 M SALU against 1.26 G
# VALU per launch, ~57 per ADMM iteration) on the critical path?  Variants of the library with K extra scalar instructions per
# iteration (-DDQQ_SALU_PROBE=K, dense_wave64.hip only; everything else linked from the shipped objects), the configs[4] step on
# ONE box, alternating.   build:  tools/ab_salu_probe.sh build    (build container)      run: tools/ab_salu_probe.sh   (GPU box)
R=$PWD
V=$R/diffqcqp_amd/lib/variants
if [ "$1" = "build" ]; then
  for K in 32 64; do
    mkdir -p $V/salu$K
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fvisibility=hidden -Wall -Wno-unused-function \
      -ffp-contract=fast -DDQQ_SALU_PROBE=$K -I $R/include -c $R/diffqcqp_amd/csrc/dense_wave64.hip -o $V/salu$K/dense_wave64.o || exit 1
    OBJS=$(ls $R/diffqcqp_amd/lib/obj/*.o | grep -v dense_wave64.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/salu$K/libdiffqcqp_hip.so $OBJS $V/salu$K/dense_wave64.o || exit 1
  done
  ls -la $V/*/libdiffqcqp_hip.so; exit 0
fi
OUT=$R/gpurun_out/ab_salu_probe.txt
mkdir -p $R/gpurun_out; : > $OUT
for rep in 1 2 3; do
  for lib in $R/diffqcqp_amd/lib/libdiffqcqp_hip.so $V/salu32/libdiffqcqp_hip.so $V/salu64/libdiffqcqp_hip.so; do
    DQQ_LIB=$lib python bench.py --config 5 --steps 5 --warmup 2 --repeats 3 --no-cpu-baseline --no-check --no-hot --details /tmp/ab_salu.json > /dev/null 2>&1
    python -c "import json; d=json.load(open('/tmp/ab_salu.json')); print('rep $rep', '$lib'.split('/')[-2], 'ms_per_step %.4f' % d['ms_per_step'], {k: round(v['mean_us'],1) for k,v in d['kernels'].items()})" >> $OUT
  done
done
cat $OUT
