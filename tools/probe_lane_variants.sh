#!/bin/bash
# lane-per-problem backward: the shipped build against tools/ubench/bin/libdqq_*.so variants (tools/build_variant.sh)
cd $GRAFT_REPO_ROOT
echo "== shipped"; python tools/probe_lane_bwd.py 2>&1 | grep -E "N=8 B= 65536|N=8 B=262144|N=6"
for f in tools/ubench/bin/libdqq_*.so; do
  echo "== $f"; DQQ_LIB=$PWD/$f python tools/probe_lane_bwd.py 2>&1 | grep -E "N=8 B= 65536|N=8 B=262144|N=6"
done
