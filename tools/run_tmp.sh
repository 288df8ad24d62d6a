mkdir -p gpurun_out/r04o
timeout -k 10 900 python -m pytest tests/test_gpu_literal.py tests/test_gpu_hits.py tests/test_gpu_configs.py -x -q -k "literal or hits or exact or erratic or aimed or two_threads or surface" > gpurun_out/r04o/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r04o/tests.log; [ $rc = 0 ] || exit 1
export CHROMA_BENCH_GEOMETRY_CACHE=/dev/shm/geo CHROMA_BENCH_NO_EXACT=1
tools/ab_env.sh "--steps 3 --warmup 1" base,CHROMA_WALK=literal base,CHROMA_WALK=literal,CHROMA_BENCH_SORT=1 2>&1 | tee gpurun_out/r04o/ab_literal_chained.txt
rm -rf /dev/shm/geo
