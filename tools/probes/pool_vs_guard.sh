mkdir -p gpurun_out/r05
# memory churn first, like a test session: a few allocate/free cycles
python tools/perf_guard.py --np 8 --nlev 72 --elems 20000 > /dev/null 2>&1
for i in 1 2; do
  echo "== default pool"; python tools/perf_guard.py --np 4 --nlev 72 --elems 10000 --twin
  echo "== pool 16 GiB"; CAAR_PLACEMENT_POOL_GIB=16 python tools/perf_guard.py --np 4 --nlev 72 --elems 10000 --twin
  echo "== pool 16 GiB nlev128"; CAAR_PLACEMENT_POOL_GIB=16 python tools/perf_guard.py --np 4 --nlev 128 --elems 12500
  echo "== default nlev128"; python tools/perf_guard.py --np 4 --nlev 128 --elems 12500
done
