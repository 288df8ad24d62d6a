mkdir -p gpurun_out/r04d
export CHROMA_BENCH_GEOMETRY_CACHE=/dev/shm/geo
timeout -k 10 500 python bench.py --steps 5 --warmup 2 > gpurun_out/r04d/bench_c3.json 2> gpurun_out/r04d/bench_c3.log || { tail -30 gpurun_out/r04d/bench_c3.log; exit 1; }
grep -E "timed|pre-sorted|generation|exact|sort of|cpu base" gpurun_out/r04d/bench_c3.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CHROMA_BENCH_NO_EXACT=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04d/rocprof -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> gpurun_out/r04d/rocprof.log || exit 1
python tools/prof_summary.py gpurun_out/r04d/rocprof gpurun_out/r04d/rocprof_c3_generation_order_summary.txt bench.py generation order
rm -rf gpurun_out/r04d/rocprof /dev/shm/geo
