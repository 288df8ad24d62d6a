"""Sum rocprofv3 --pmc counter_collection csv files per kernel (all dispatches and the largest one)."""
import csv, glob, os, sys, collections
d = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
big = {}
for f in sorted(glob.glob(os.path.join(d, 'pass*', '**', '*counter_collection.csv'), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:48]
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (k, r['Counter_Name'])
        g = int(r['Grid_Size']) if 'Grid_Size' in r else int(r.get('Grid_Size_X', 0) or 0)
        if key not in big or g > big[key][0]:
            big[key] = (g, float(r['Counter_Value']))
lines = []
for k in sorted(tot):
    if not any(x in k for x in ('k_raycast', 'k_physics', 'k_propagate')):
        continue
    lines.append('== %s (sum over dispatches | largest dispatch)' % k)
    for c in sorted(tot[k]):
        lines.append('  %-36s %18.0f | %16.0f (grid %d)' % (c, tot[k][c], big[(k, c)][1], big[(k, c)][0]))
open(os.path.join(d, 'summary.txt'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
