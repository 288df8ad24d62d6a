"""Every kernel of ONE propagate call (between the last two k_load_working launches) of a rocprofv3 --kernel-trace run, with start,
gap to the previous kernel end and duration: where a small batch spends its time.  usage: python tools/trace_call.py TRACE_DIR"""
import csv, glob, os, sys
d = sys.argv[1]
trace = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
rows = []
for r in csv.DictReader(open(trace)):
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith('k_load_working')]
a = starts[-2]; b = starts[-1]
t0 = rows[a][0]
prev_end = None
for s, e, name in rows[a:b+1]:
    print('%9.1f us  +%7.1f gap  %8.1f us  %s' % ((s - t0) / 1e3, 0.0 if prev_end is None else (s - prev_end) / 1e3, (e - s) / 1e3, name))
    prev_end = e
