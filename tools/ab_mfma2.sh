mkdir -p gpurun_out/r03
L=gpurun_out/r03/mfma_vs_dpp_other.log
: > $L
for i in 1 2; do
for lib in libcaar_hip_mfma.so libcaar_hip.so; do
  echo "== $lib" >> $L
  CAAR_LIBRARY_PATH=$PWD/tinman_sandbox_amd/csrc/$lib timeout -k 10 200 python tools/kbench.py --nlev 128 --elems 12500 --variants 0,1 --skeletons 0 --rounds 3 2>/dev/null | grep "^variant" | cut -c1-110 >> $L
done
done
for lib in libcaar_hip_mfma.so libcaar_hip.so; do
  echo "== $lib" >> $L
  CAAR_LIBRARY_PATH=$PWD/tinman_sandbox_amd/csrc/$lib timeout -k 10 300 python tools/eulerian_bench.py 2>/dev/null | grep "^np=" | cut -c1-330 >> $L
  CAAR_LIBRARY_PATH=$PWD/tinman_sandbox_amd/csrc/$lib timeout -k 10 300 python tools/anylev_bench.py 2>/dev/null | grep "nlev" | cut -c1-120 >> $L
  CAAR_LIBRARY_PATH=$PWD/tinman_sandbox_amd/csrc/$lib timeout -k 10 200 python tools/steps_bench.py --nlev 128 --elems 1024,12500 2>/dev/null | grep "variant  0" | cut -c1-100 >> $L
done
cat $L
