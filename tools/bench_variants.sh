#!/bin/bash
# The measured variants of SURVEY.md section 8(d) beside the default line: wavelengths U(400, 800) nm (the reference benchmark's
# own choice, chroma/benchmark.py:81) and max_steps = 10 (the default of GPUPhotons.propagate, chroma/gpu/photon.py:194), on C3
# (29 007 PMTs) and C2 (demo.detector()).  usage: tools/bench_variants.sh OUTDIR
set -u
out=$1; mkdir -p $out
export CHROMA_BENCH_GEOMETRY_CACHE=/dev/shm/chroma_geo_cache CHROMA_BENCH_NO_EXACT=1
for cfg in c3 detector; do
  for v in "400nm_steps100:" "u400_800_steps100:--wavelength-hi 800" "400nm_steps10:--max-steps 10" "u400_800_steps10:--wavelength-hi 800 --max-steps 10"; do
    name=${v%%:*}; args=${v#*:}
    python bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline $args > $out/bench_${cfg}_$name.json 2> $out/bench_${cfg}_$name.log || exit 1
    echo "$cfg $name: $(grep timed $out/bench_${cfg}_$name.log)"
  done
done
rm -rf /dev/shm/chroma_geo_cache
python - $out <<'PY'
import json, glob, os, sys
rows = []
for f in sorted(glob.glob(os.path.join(sys.argv[1], 'bench_*_*.json'))):
    j = json.load(open(f)); c = j['config']
    rows.append('%-36s %10.4g photons/s  %8.2f ms/step  steps/photon %.3f  max_steps %d  wavelength %s' % (
        os.path.basename(f)[6:-5], j['value'], j['ms_per_step'], c['steps_per_photon'], c['max_steps'], c['wavelength_nm']))
open(os.path.join(sys.argv[1], 'bench_variants.txt'), 'w').write('\n'.join(rows) + '\n')
print('\n'.join(rows))
PY
