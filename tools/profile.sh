#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of the bench command, then the PMC
# passes (SQ counters, FETCH_SIZE, WRITE_SIZE in separate runs, kernel-trace only), then condenses
# everything into gpurun_out/prof_summary/.   Usage: tools/profile.sh <tag> <suffix|-> [bench args...]
#   tools/profile.sh r02 -                  headline step      -> r02_*,      pmc_latest.json
#   tools/profile.sh r02_cfg5 _cfg5 --config 5                 -> r02_cfg5_*, pmc_latest_cfg5.json
# Profiled with --streams 1 so that per-kernel durations are those of kernels running alone (the
# roofline brackets of bench.py are taken the same way).
set -u
TAG=${1:-r02}; shift || true
SUF=${1:--}; shift || true
[ "$SUF" = "-" ] && SUF=""
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT $R/gpurun_out/prof_summary
export TMPDIR=/tmp
cd /tmp
COMMON="--no-cpu-baseline --no-hot --no-per-config --streams 1"
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $R/bench.py --steps 20 --warmup 5 --repeats 3 $COMMON "$@" > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/sq -o sq -- python $R/bench.py --steps 2 --warmup 1 --repeats 1 --no-check $COMMON "$@" > /dev/null 2> $OUT/sq.err
# second SQ pass (8 slots per pass): VALU lane utilisation and where the wave-cycles go.  SQ_WAVE_CYCLES,
# SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS -d $OUT/sq2 -o sq2 -- python $R/bench.py --steps 2 --warmup 1 --repeats 1 --no-check $COMMON "$@" > /dev/null 2> $OUT/sq2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fe -o fe -- python $R/bench.py --steps 2 --warmup 1 --repeats 1 --no-check $COMMON "$@" > /dev/null 2> $OUT/fe.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/wr -o wr -- python $R/bench.py --steps 2 --warmup 1 --repeats 1 --no-check $COMMON "$@" > /dev/null 2> $OUT/wr.err
cd $R
python tools/summarize_prof.py $OUT/kt/kt_results.db $OUT/sq/sq_results.db $OUT/fe/fe_results.db $OUT/wr/wr_results.db gpurun_out/prof_summary $TAG "$SUF" $OUT/sq2/sq2_results.db
cp $OUT/bench_under_rocprof.json gpurun_out/prof_summary/${TAG}_bench_under_rocprof.json
tail -3 $OUT/sq.err; tail -3 $OUT/sq2.err
rm -rf $OUT/kt $OUT/sq $OUT/sq2 $OUT/fe $OUT/wr
