"""Where the GPU idles between the kernels of a bench run: reads the kernel trace rocprofv3 wrote (csv) and lists the
largest gaps with the kernels on either side.  usage: trace_gaps.py DIR [min_gap_us]"""
import csv, glob, os, sys
d = sys.argv[1]; min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
rows = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]))
rows.sort()
busy = sum(e - s for s, e, _ in rows)
print('%d kernels, busy %.1f ms, span %.1f ms' % (len(rows), busy / 1e6, (rows[-1][1] - rows[0][0]) / 1e6))
gaps = {}
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = (s1 - e0) / 1e3
    if g >= min_gap:
        k = (n0, n1)
        c = gaps.setdefault(k, [0, 0.0]); c[0] += 1; c[1] += g
for (n0, n1), (cnt, tot) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print('%9.1f us total in %4d gaps (avg %7.1f us)  after %-40s before %s' % (tot, cnt, tot / cnt, n0, n1))
