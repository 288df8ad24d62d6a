"""Summarise a rocprofv3 --kernel-trace --stats run (csv output) into a small text file for profiles/."""
import csv, glob, sys, os
d = sys.argv[1]
out = sys.argv[2]
stats = glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True)[0]
trace = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
lines = []
lines.append('# rocprofv3 --kernel-trace --stats summary (%s)' % ' '.join(sys.argv[3:]))
lines.append('')
lines.append('name, calls, total_ms, avg_ms, pct, min_ms, max_ms')
for r in csv.DictReader(open(stats)):
    lines.append('%s, %s, %.3f, %.4f, %s, %.4f, %.4f' % (r['Name'].replace('(anonymous namespace)::', '').split('(')[0][:60], r['Calls'], int(r['TotalDurationNs']) / 1e6,
                 float(r['AverageNs']) / 1e6, r['Percentage'], int(r['MinNs']) / 1e6, int(r['MaxNs']) / 1e6))
lines.append('')
lines.append('# per-launch trace of the propagate path kernels (first 60): name, grid, vgpr, lds, scratch, dur_ms')
n = 0
for r in csv.DictReader(open(trace)):
    name = r['Kernel_Name']
    if any(k in name for k in ('k_propagate', 'k_raycast', 'k_physics', 'k_sort', 'k_morton')):
        lines.append('%s, %s, %s, %s, %s, %.3f' % (name.replace('(anonymous namespace)::', '').split('(')[0][:40], r['Grid_Size_X'], r['VGPR_Count'], r['LDS_Block_Size'], r['Scratch_Size'],
                     (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
        n += 1
        if n >= 60: break
open(out, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[:14]))
