#!/bin/bash
# A/B of library builds on one box: tools/ab_many.sh libA.so libB.so [repeats]   (each repeat = tools/ab_libs.py once)
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do echo "== $A vs $B (run $i)"; python tools/ab_libs.py $A $B 2>&1 | grep -v amdgpu.ids | cut -c1-62; done
