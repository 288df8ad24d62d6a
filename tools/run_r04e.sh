mkdir -p gpurun_out/r04e
timeout -k 10 600 python -m pytest tests/test_gpu_hits.py tests/test_gpu_literal.py tests/test_gpu_sim_pipeline.py -x -q > gpurun_out/r04e/tests.log 2>&1; rc=$?; tail -5 gpurun_out/r04e/tests.log; [ $rc = 0 ] || exit 1
export CHROMA_BENCH_GEOMETRY_CACHE=/dev/shm/geo CHROMA_BENCH_NO_EXACT=1
tools/ab_env.sh "--steps 4 --warmup 1" base base,CHROMA_BENCH_SEPARATE_HITS=1 base,CHROMA_BENCH_SORT=1 base,CHROMA_BENCH_SORT=1,CHROMA_BENCH_SEPARATE_HITS=1 2>&1 | tee gpurun_out/r04e/ab_fused_hits.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04e/rocprof -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> gpurun_out/r04e/rocprof.log || exit 1
python tools/prof_summary.py gpurun_out/r04e/rocprof gpurun_out/r04e/rocprof_c3_fused_hits_summary.txt bench.py generation order fused hits | head -30
rm -rf gpurun_out/r04e/rocprof /dev/shm/geo
