#!/bin/bash
# Developer A/B without rebuilding on the GPU box: one translation unit recompiled with extra -D flags and linked with the
# shipped objects into tools/ubench/bin/libdqq_<tag>.so (git-ignored, travels with the snapshot).  Use with
#   DQQ_LIB=tools/ubench/bin/libdqq_<tag>.so python tools/probe_...py
# usage: tools/build_variant.sh <tag> <unit.hip> <contract: off|fast|fast-honor-pragmas> [-Dflags...]
set -e
cd "$(dirname "$0")/.."
TAG=$1; UNIT=$2; FPC=$3; shift 3
mkdir -p tools/ubench/bin/obj_$TAG
OBJ=tools/ubench/bin/obj_$TAG/${UNIT%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fvisibility=hidden -Wall -Wno-unused-function \
  -ffp-contract=$FPC "$@" -I include -c diffqcqp_amd/csrc/$UNIT -o $OBJ
OTHERS=$(ls diffqcqp_amd/lib/obj/*.o | grep -v "/${UNIT%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/bin/libdqq_$TAG.so $OBJ $OTHERS
echo tools/ubench/bin/libdqq_$TAG.so
