#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of the bench command, then the PMC
# passes (SQ counters, FETCH_SIZE, WRITE_SIZE in separate runs, kernel-trace only), then condenses
# everything into gpurun_out/prof_summary/.   Usage: tools/profile.sh <tag> [bench args...]
# Profiled with --streams 1 so that per-kernel durations are those of kernels running alone (the
# roofline brackets of bench.py are taken the same way).
set -u
TAG=${1:-r01}; shift || true
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT $R/gpurun_out/prof_summary
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --streams 1 "$@" > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/sq -o sq -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check --streams 1 "$@" > /dev/null 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fe -o fe -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check --streams 1 "$@" > /dev/null 2> $OUT/fe.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/wr -o wr -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check --streams 1 "$@" > /dev/null 2> $OUT/wr.err
cd $R
python tools/summarize_prof.py $OUT/kt/kt_results.db $OUT/sq/sq_results.db $OUT/fe/fe_results.db $OUT/wr/wr_results.db gpurun_out/prof_summary $TAG
cp $OUT/bench_under_rocprof.json gpurun_out/prof_summary/${TAG}_bench_under_rocprof.json
rm -rf $OUT/kt $OUT/sq $OUT/fe $OUT/wr
