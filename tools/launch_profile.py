"""Per-launch durations of the propagate path from a rocprofv3 --kernel-trace run (csv): one line per step of the LAST
chroma_propagate call in the trace -- ray cast, main physics pass, retry, fix-up physics -- and the kernels around them.
usage: python tools/launch_profile.py TRACE_DIR [OUT]"""
import csv, glob, os, sys
d = sys.argv[1]
trace = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
rows = []
for r in csv.DictReader(open(trace)):
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name))
rows.sort()
# the last call: from the last k_load_working on
start = max(i for i, r in enumerate(rows) if r[2].startswith('k_load_working'))
call = rows[start:]
t0 = call[0][0]
out = ['# %s: the last propagate call, %d kernels, %.3f ms from the first start to the last end' % (trace, len(call), (max(r[1] for r in call) - t0) / 1e6),
       '# step: raycast ms | physics ms | retry ms | fix-up ms | gaps ms (idle between the kernels of the step)']
step, cur, last_end = 0, None, None
steps = []
others = []
for s, e, name in call:
    dur = (e - s) / 1e6
    if name.startswith(('k_raycast_quad', 'k_raycast_literal', 'k_raycast_pair')):
        cur = {'cast': dur, 'phys': [], 'retry': 0.0, 'gap': 0.0, 'end': e}
        steps.append(cur)
    elif cur is not None and name.startswith('k_physics'):
        cur['phys'].append(dur)
    elif cur is not None and name.startswith('k_raycast_retry'):
        cur['retry'] += dur
    else:
        others.append((name, dur, (s - t0) / 1e6))
    if cur is not None and last_end is not None and s > last_end:
        cur['gap'] += (s - last_end) / 1e6
    last_end = max(last_end or e, e)
tot = [0.0, 0.0, 0.0, 0.0, 0.0]
for i, st in enumerate(steps):
    ph = st['phys'] + [0.0, 0.0]
    vals = [st['cast'], ph[0], st['retry'], sum(ph[1:]), st['gap']]
    tot = [a + b for a, b in zip(tot, vals)]
    out.append('%3d: %8.3f | %7.3f | %6.3f | %6.3f | %6.3f' % tuple([i + 1] + vals))
out.append('sum: %8.3f | %7.3f | %6.3f | %6.3f | %6.3f' % tuple(tot))
for k in (5, 10, 15):
    if len(steps) > k:
        rest = steps[k:]
        out.append('steps after %d: %.3f ms in %d steps (cast %.3f)' % (k, sum(s['cast'] + sum(s['phys']) + s['retry'] + s['gap'] for s in rest), len(rest), sum(s['cast'] for s in rest)))
out.append('# other kernels of the call: name, ms, start (ms after the call began)')
for name, dur, at in others:
    out.append('%s, %.3f, %.3f' % (name[:48], dur, at))
text = '\n'.join(out) + '\n'
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(text)
print(text)
