#!/bin/bash
# Counter passes of one more bench configuration into profiles/pmc_traffic.json (what tools/refresh_profiles.sh PART=2 does for C3):
# FETCH_SIZE / WRITE_SIZE in separate --pmc passes, then the SQ groups.  usage: tools/pmc_config.sh OUTDIR CONFIG PHOTONS [MAX_STEPS]
set -u
out=$1; cfg=$2; n=$3; ms=${4:-100}
mkdir -p $out
export CHROMA_BENCH_GEOMETRY_CACHE=/dev/shm/chroma_geo_cache CHROMA_BENCH_NO_EXACT=1
python bench.py --config $cfg --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> $out/cache_fill_$cfg.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${cfg}_$c -- python3 bench.py --config $cfg --no-cpu-baseline --steps 1 --warmup 1 > $out/pmc_${cfg}_$c.stdout 2> $out/pmc_${cfg}_$c.stderr || exit 1
done
python tools/pmc_traffic.py $out/pmc_${cfg}_FETCH_SIZE $out/pmc_${cfg}_WRITE_SIZE $cfg:$n:$ms > $out/pmc_traffic_$cfg.txt
python tools/pmc_traffic.py $out/pmc_${cfg}_FETCH_SIZE $out/pmc_${cfg}_WRITE_SIZE $cfg:$n:$ms:physics k_physics >> $out/pmc_traffic_$cfg.txt
PMC_GROUPS="1 2" PMC_TIMEOUT=300 tools/pmc.sh $out/pmc_sq_$cfg python3 bench.py --config $cfg --no-cpu-baseline --steps 1 --warmup 1 > $out/pmc_sq_$cfg.txt 2>&1
python tools/pmc_sq.py $out/pmc_sq_$cfg $cfg:$n:$ms >> $out/pmc_traffic_$cfg.txt
rm -rf $out/pmc_${cfg}_FETCH_SIZE $out/pmc_${cfg}_WRITE_SIZE $out/pmc_sq_$cfg/pass*
CHROMA_BENCH_NO_EXACT= python bench.py --config $cfg --no-cpu-baseline > $out/bench_${cfg}_with_traffic.json 2> $out/bench_${cfg}_with_traffic.log
cat $out/pmc_traffic_$cfg.txt
