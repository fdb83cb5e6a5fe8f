mkdir -p gpurun_out/r03
bash tools/ab_old_new.sh libcaar_hip_mfma.so > /dev/null
grep "==\|^variant 0 \|^variant 1 " gpurun_out/r03/kbench_old_vs_new.log | cut -c1-110
cp gpurun_out/r03/kbench_old_vs_new.log gpurun_out/r03/kbench_mfma_vs_dpp_4w.log
for lib in libcaar_hip_mfma.so libcaar_hip.so libcaar_hip_mfma.so libcaar_hip.so; do echo "== $lib"; CAAR_LIBRARY_PATH=$PWD/tinman_sandbox_amd/csrc/$lib timeout -k 10 200 python tools/steps_bench.py --elems 64,1024,10000 2>/dev/null | grep "variant  0" | cut -c1-100; done | tee gpurun_out/r03/steps_bench_mfma_vs_dpp_4w.log
