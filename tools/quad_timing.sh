#!/bin/bash
# Where a ray-cast wave's cycles go: the diagnostic build (-DQUAD_TIMING=1, build_variants/lib_timing.so) stamps
# s_memtime around the phases of k_raycast_quad; a few waves of every large launch print their totals.
# usage (GPU box): tools/quad_timing.sh OUT.txt
export CHROMA_BENCH_GEOMETRY_CACHE=${CHROMA_BENCH_GEOMETRY_CACHE:-/dev/shm/chroma_geo_cache}
CHROMA_HIP_LIBRARY=$PWD/build_variants/lib_timing.so python bench.py --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep -E "^QT|timed" > $1
python - "$1" <<'PY'
import sys, collections
rows = collections.defaultdict(list)
for line in open(sys.argv[1]):
    if not line.startswith('QT'): 
        print(line.strip()); continue
    f = line.split()
    d = {f[i]: int(f[i + 1]) for i in range(1, len(f) - 1, 2)}
    rows[d['rays']].append(d)
print('rays/launch  waves | cycles per wave | refill%  pop%  wait%  node%  leaf% other% | iters  active/iter  cyc/iter: pop wait node | leaf rounds  tests/round  cyc/round')
for rays in sorted(rows, reverse=True)[:8]:
    r = rows[rays]; n = len(r)
    s = {k: sum(x[k] for x in r) / n for k in r[0]}
    tot = s['total']; other = tot - s['refill'] - s['pop'] - s['wait'] - s['node'] - s['leaf']
    print('%10d %5d | %12.0f | %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f | %6.0f %6.2f   %5.0f %5.0f %5.0f | %8.0f %8.2f %8.0f' % (
        rays, n, tot, 100 * s['refill'] / tot, 100 * s['pop'] / tot, 100 * s['wait'] / tot, 100 * s['node'] / tot, 100 * s['leaf'] / tot, 100 * other / tot,
        s['iters'], s['active'] / max(1, s['iters']), s['pop'] / max(1, s['iters']), s['wait'] / max(1, s['iters']), s['node'] / max(1, s['iters']),
        s['rounds'], s['tests'] / max(1, s['rounds']), s['leaf'] / max(1, s['rounds'])))
PY
