mkdir -p gpurun_out/r04k
timeout -k 10 600 python -m pytest tests/test_gpu_hits.py tests/test_gpu_sim_pipeline.py -x -q > gpurun_out/r04k/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r04k/tests.log; [ $rc = 0 ] || exit 1
export CHROMA_BENCH_GEOMETRY_CACHE=/dev/shm/geo CHROMA_BENCH_NO_EXACT=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04k/rocprof -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> gpurun_out/r04k/rocprof.log || exit 1
python tools/prof_summary.py gpurun_out/r04k/rocprof gpurun_out/r04k/summary.txt bench.py | head -20
rm -rf gpurun_out/r04k/rocprof /dev/shm/geo
