#!/bin/bash
# One GPU-box pass that refreshes the judged artefacts: GPU tests, the bench lines of every config, the
# rocprofv3 kernel summary of the default bench command and its HBM-traffic counters (FETCH_SIZE / WRITE_SIZE in
# separate --pmc passes, no trace flags beside them).  usage: tools/refresh_profiles.sh OUTDIR
# (a gpurun call is limited to 20 minutes: PART=1 runs the tests and the bench lines, PART=2 the profiles of the default walk,
#  PART=3 the exact walk's profiles and the measured variants of the contract; default 1 and 2)
set -u
out=$1
part=${PART:-12}
mkdir -p $out
export CHROMA_BENCH_GEOMETRY_CACHE=/dev/shm/chroma_geo_cache
if [[ $part == *1* ]]; then
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/gpu_tests.log 2>&1 || { tail -20 $out/gpu_tests.log; exit 1; }
tail -1 $out/gpu_tests.log
python bench.py > $out/bench_c3_default.json 2> $out/bench_c3_default.log || exit 1
for cfg in tiny lite c5 detector; do
  python bench.py --config $cfg > $out/bench_$cfg.json 2> $out/bench_$cfg.log || exit 1
done
fi
if [[ $part == *2* ]]; then
export CHROMA_BENCH_NO_EXACT=1      # (the profiles are of the default walk: no extra batch through the literal one)
# (the profiles are of the propagate path: the geometry comes from the cache -- filled here if PART=2 runs on a box of its own --
#  so that the builders' kernels, run once per geometry, do not head the per-kernel list)
[ -d /dev/shm/chroma_geo_cache ] || python bench.py --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> $out/cache_fill.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/rocprof_c3 -- python3 bench.py --no-cpu-baseline > $out/rocprof_c3.json 2> $out/rocprof_c3.log || exit 1
python tools/prof_summary.py $out/rocprof_c3 $out/rocprof_c3_default_summary.txt bench.py
python tools/launch_profile.py $out/rocprof_c3 $out/launch_profile_c3.txt > /dev/null      # (per-launch durations of the last call of the same run)
rm -rf $out/rocprof_c3
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 > $out/pmc_$c.stdout 2> $out/pmc_$c.stderr || exit 1
done
python tools/pmc_traffic.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE c3:100000000:100 > $out/pmc_traffic.txt
python tools/pmc_traffic.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE c3:100000000:100:physics k_physics >> $out/pmc_traffic.txt
cat $out/pmc_traffic.txt
rm -rf $out/pmc_FETCH_SIZE/*/*agent_info.csv
# issue-side counters (SQ groups 1 and 2 of tools/pmc.sh) of the same command, into the same JSON under the same hash
PMC_GROUPS="1 2" PMC_TIMEOUT=300 tools/pmc.sh $out/pmc_sq python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 > $out/pmc_sq.txt 2>&1
python tools/pmc_sq.py $out/pmc_sq c3:100000000:100 >> $out/pmc_traffic.txt
rm -rf $out/pmc_sq/pass*/*/*agent_info.csv
# memory-instruction mix and waits of k_physics (tools/pmc_physics.sh)
tools/pmc_physics.sh $out/pmc_physics_raw > $out/pmc_physics.txt 2>&1
rm -rf $out/pmc_physics_raw
CHROMA_BENCH_NO_EXACT= python bench.py --no-cpu-baseline > $out/bench_c3_with_traffic.json 2> $out/bench_c3_with_traffic.log
tools/isa_report.sh > $out/isa_resources.txt 2>/dev/null
# the geometry set-up on its own: phases (CHROMA_TIMING) and the builders' kernels
CHROMA_TIMING=1 python tools/time_geometry_build.py c3 > $out/wide_device_build.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/rocprof_geo -- python3 tools/time_geometry_build.py c3 > /dev/null 2> $out/rocprof_geo.log && python tools/prof_summary.py $out/rocprof_geo $out/rocprof_geometry_build_summary.txt tools/time_geometry_build.py
rm -rf $out/rocprof_geo
rm -rf /dev/shm/chroma_geo_cache
cat $out/bench_*.json
fi
if [[ $part == *3* ]]; then
# the exact walk (CHROMA_WALK=literal): bench line, rocprofv3 summary, SQ + traffic counters; the variants of SURVEY 8(d)
tools/prof_literal.sh $out/literal
mv $out/literal/bench_literal.json $out/bench_c3_literal.json; mv $out/literal/bench_literal.log $out/bench_c3_literal.log
mv $out/literal/rocprof_literal_summary.txt $out/rocprof_c3_literal_summary.txt; mv $out/literal/pmc/summary.txt $out/pmc_literal_sq_traffic.txt
rm -rf $out/literal
tools/bench_variants.sh $out/variants
fi
