#!/bin/bash
# Everything profiles/ holds for ONE tree, in one GPU-box run:  tools/final_evidence.sh <tag> [fuzz trials]
#   smoke + GPU test log, the driver's bench command (line + details), tools/profile.sh for the headline and workloads 2-7
#   (rocprofv3 kernel trace + separate SQ / FETCH_SIZE / WRITE_SIZE passes), the configs[4] iteration match, the fuzz soak.
# Output: gpurun_out/prof_summary/<tag>_*  (+ pmc_latest*.json) -- copy into profiles/.
TAG=${1:-r07}; T=${2:-1000}
R=$PWD; O=$R/gpurun_out/prof_summary; mkdir -p $O
# (fuzz trials 0: the measurement half only -- bench + profiles; the test / fuzz logs of the same csrc stay valid)
if [ "$T" != "0" ]; then
python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; tail -1 $O/${TAG}_smoke.log
timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/${TAG}_gpu_tests.log; tail -1 $O/${TAG}_gpu_tests.log
fi
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_line.json 2> /dev/null; cp bench_details.json $O/${TAG}_bench_details.json; wc -c $O/${TAG}_bench_line.json
tools/profile.sh $TAG - > /dev/null 2>&1
for k in 2 3 4 5 6 7; do tools/profile.sh ${TAG}_cfg$k _cfg$k --config $k > /dev/null 2>&1; done
# the default line again, now quoting the counter summaries of THIS build (pmc_matches_build: true)
mkdir -p $R/profiles; cp $O/pmc_latest*.json $R/profiles/
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_line.json 2> /dev/null; cp bench_details.json $O/${TAG}_bench_details.json
if [ "$T" != "0" ]; then
python tools/cfg5_iteration_match.py > $O/${TAG}_cfg5_match.txt 2>&1; tail -1 $O/${TAG}_cfg5_match.txt
bash tools/soak_all.sh $T > $O/${TAG}_fuzz_soak.txt 2>&1; grep -c "== \[" $O/${TAG}_fuzz_soak.txt; grep -i "fail\|error" $O/${TAG}_fuzz_soak.txt | head -5
fi
rm -f $O/*_bench_under_rocprof.json
ls $O | head -60
