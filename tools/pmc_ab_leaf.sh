set -u
export CHROMA_BENCH_GEOMETRY_CACHE=/dev/shm/chroma_geo_cache
export PMC_GROUPS="1 2" PMC_TIMEOUT=200
python bench.py --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> gpurun_out/r02_prime.log; tail -2 gpurun_out/r02_prime.log
tools/pmc.sh gpurun_out/r02_pmc_leaf1 python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > gpurun_out/r02_pmc_leaf1.txt 2>&1
CHROMA_HIP_LIBRARY=$PWD/build_variants/lib_leaf0.so tools/pmc.sh gpurun_out/r02_pmc_leaf0 python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > gpurun_out/r02_pmc_leaf0.txt 2>&1
grep -A40 "k_raycast_quad<false>" gpurun_out/r02_pmc_leaf1.txt | head -22
grep -A40 "k_raycast_quad<false>" gpurun_out/r02_pmc_leaf0.txt | head -22
rm -rf /dev/shm/chroma_geo_cache
