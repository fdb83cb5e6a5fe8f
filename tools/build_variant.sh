#!/bin/bash
# A library build that differs from the default one by -D flags on some translation units (for A/B runs with
# CAAR_LIBRARY_PATH / tools/ab_old_new.sh):   bash tools/build_variant.sh <out.so> "<flags>" <tu.hip> [<tu.hip> ...]
# The other objects are the default build's (tinman_sandbox_amd/csrc/build/*.o: run the normal build first).
set -e
OUT=$1; FLAGS=$2; shift 2
C=$(cd "$(dirname "$0")/../tinman_sandbox_amd/csrc" && pwd)
T=$(mktemp -d /tmp/caar_variant.XXXXXX)
OBJS=""
for f in $C/build/*.o; do
  case "$f" in *.debug.o|*.extra.o) continue;; esac
  b=$(basename $f .o); skip=0
  for tu in "$@"; do [ "$b" = "$(basename $tu .hip)" ] && skip=1; done
  [ $skip = 0 ] && OBJS="$OBJS $f"
done
for tu in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc $FLAGS -c $C/$tu -o $T/$(basename $tu .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $OBJS $T/*.o -o $C/$OUT
rm -rf $T
echo built $C/$OUT
