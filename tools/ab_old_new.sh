#!/bin/bash
# A/B of several builds of the library on one box, alternating processes:
#     bash tools/ab_old_new.sh <tag> <lib.so> [<lib.so> ...]        (files in tinman_sandbox_amd/csrc/; CAAR_LIBRARY_PATH selects one)
# 1. tools/ab_bits.py under every build: all must agree with the first in every bit (a difference is reported, not fatal);
# 2. kbench (default kernel + all-streaming twin) and steps_bench (the driver loop as one launch), three alternating rounds.
# Environment: AB_CASES = space-separated subset of "np4_72 np4_128 np8" (default np4_72), AB_BITS=0 skips step 1, ROUND.
TAG=$1; shift
LIBS="$@"
R=${ROUND:-r04}
CASES=${AB_CASES:-np4_72}
mkdir -p gpurun_out/$R
L=gpurun_out/$R/${TAG}_kbench.log
S=gpurun_out/$R/${TAG}_steps.log
B=gpurun_out/$R/${TAG}_bits
C=$PWD/tinman_sandbox_amd/csrc
if [ "${AB_BITS:-1}" != 0 ]; then
  : > ${B}.log
  first=""
  for lib in $LIBS; do
    CAAR_LIBRARY_PATH=$C/$lib timeout -k 10 500 python tools/ab_bits.py > ${B}_$lib.txt 2> ${B}_$lib.err || { tail -5 ${B}_$lib.err; exit 1; }
    if [ -z "$first" ]; then first=$lib; continue; fi
    if diff -q ${B}_$first.txt ${B}_$lib.txt > /dev/null; then echo "$lib vs $first: bit-identical, $(wc -l < ${B}_$lib.txt) fingerprints agree" | tee -a ${B}.log
    else echo "$lib vs $first: BITS DIFFER in $(diff ${B}_$first.txt ${B}_$lib.txt | grep -c '^>') of $(wc -l < ${B}_$lib.txt) fingerprints" | tee -a ${B}.log; diff ${B}_$first.txt ${B}_$lib.txt | head -6; fi
  done
fi
: > $L; : > $S
for i in 1 2 3; do
  for lib in $LIBS; do
    for c in $CASES; do
      case $c in
        np4_72) KA="--variants 0,1"; SA="--elems 10000 --reps 5";;
        np4_128) KA="--nlev 128 --elems 12500 --variants 0,1 --reps 10"; SA="--nlev 128 --elems 12500 --reps 3";;
        np8) KA="--np 8 --elems 20000 --variants 0 --reps 10"; SA="--np 8 --elems 20000 --reps 3";;
      esac
      echo "== $lib $c" >> $L; echo "== $lib $c" >> $S
      CAAR_LIBRARY_PATH=$C/$lib timeout -k 10 200 python tools/kbench.py $KA --skeletons 0 --rounds 3 2>/dev/null | grep "^variant [0-9]  " | cut -c1-120 >> $L || exit 1
      CAAR_LIBRARY_PATH=$C/$lib timeout -k 10 200 python tools/steps_bench.py $SA 2>/dev/null | grep "^E=" | cut -c1-150 >> $S || exit 1
    done
  done
done
cat $L $S
