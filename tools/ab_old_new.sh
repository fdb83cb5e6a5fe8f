#!/bin/bash
# A/B of two builds of the library on one box, alternating processes:  bash tools/ab_old_new.sh [other.so]
# (tinman_sandbox_amd/csrc/<other.so> built from another revision; CAAR_LIBRARY_PATH selects the build)
OTHER=${1:-libcaar_hip_old.so}
mkdir -p gpurun_out/r03
L=gpurun_out/r03/kbench_old_vs_new.log
: > $L
for i in 1 2 3; do
  for lib in $OTHER libcaar_hip.so; do
    echo "== $lib" >> $L
    CAAR_LIBRARY_PATH=$PWD/tinman_sandbox_amd/csrc/$lib timeout -k 10 200 python tools/kbench.py --variants 0,1 --skeletons 0 --rounds 3 2>/dev/null | grep "^variant" | cut -c1-120 >> $L || exit 1
  done
done
cat $L
