mkdir -p gpurun_out/r04l
timeout -k 10 600 python -m pytest tests/test_gpu_hits.py tests/test_gpu_sim_pipeline.py tests/test_gpu_parity.py -x -q > gpurun_out/r04l/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r04l/tests.log; [ $rc = 0 ] || exit 1
export CHROMA_BENCH_GEOMETRY_CACHE=/dev/shm/geo CHROMA_BENCH_NO_EXACT=1
tools/ab_env.sh "--steps 5 --warmup 1" base base,CHROMA_BENCH_SORT=1 base 2>&1 | tee gpurun_out/r04l/ab.txt
tools/ab_env.sh "--config detector --steps 5 --warmup 1" base 2>&1 | tee -a gpurun_out/r04l/ab.txt
rm -rf /dev/shm/geo
