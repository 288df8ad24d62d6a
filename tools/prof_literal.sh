#!/bin/bash
# Profiles of the exact walk (CHROMA_WALK=literal) at C3: rocprofv3 kernel summary, SQ issue-side counters (groups 1, 2 of
# tools/pmc.sh) and the ray cast's HBM traffic (FETCH_SIZE / WRITE_SIZE in passes of their own).  usage: tools/prof_literal.sh OUTDIR [bench args]
set -u
out=$1; shift
mkdir -p $out
export CHROMA_WALK=literal CHROMA_BENCH_GEOMETRY_CACHE=/dev/shm/chroma_geo_cache CHROMA_BENCH_NO_EXACT=1
python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $out/bench_literal.json 2> $out/bench_literal.log || exit 1
grep -E "timed|step " $out/bench_literal.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/rocprof_lit -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > /dev/null 2> $out/rocprof_lit.log || exit 1
python tools/prof_summary.py $out/rocprof_lit $out/rocprof_literal_summary.txt bench.py CHROMA_WALK=literal "$@"
rm -rf $out/rocprof_lit
PMC_GROUPS="${PMC_GROUPS:-1 2 4 5}" PMC_TIMEOUT=300 tools/pmc.sh $out/pmc python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" > $out/pmc_literal.txt 2>&1
grep -A40 "k_raycast_literal" $out/pmc/summary.txt | head -45
rm -rf $out/pmc/pass*
rm -rf /dev/shm/chroma_geo_cache
