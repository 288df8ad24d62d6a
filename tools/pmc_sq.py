"""Issue-side counters of the ray cast and of k_physics from the SQ passes of tools/pmc.sh (groups 1 and 2) into
profiles/pmc_traffic.json, next to the traffic figures and under the same source hash: bench.py quotes them as
roofline.valu_issue_frac / valu_lane_utilisation when the hash matches the build it runs.

  valu_issue_frac       = SQ_ACTIVE_INST_VALU * 4 / (SQ_BUSY_CYCLES * 32)      (a wave64 VALU instruction holds its SIMD for 4
                          cycles; SQ_BUSY_CYCLES is summed over 32 shader engines, the chip has 1024 SIMDs: DESIGN.md section 3.1)
  valu_lane_utilisation = SQ_THREAD_CYCLES_VALU / (SQ_INSTS_VALU * 64)
usage: pmc_sq.py PMC_DIR key            (key like c3:100000000:100; writes key + ':sq' and key + ':physics:sq')"""
import csv, glob, json, os, sys, collections, datetime, hashlib
d, key = sys.argv[1:3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(d, 'pass*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        k = 'k_raycast_quad<false>' if 'k_raycast_quad<false>' in name else 'k_physics' if 'k_physics' in name else None
        if k:
            tot[k][r['Counter_Name']] += float(r['Counter_Value'])
sys.path.insert(0, root)
from bench import kernel_source_hash      # (the one definition of what identifies a kernel build)
class _H(object):
    def hexdigest(self): return kernel_source_hash()
h = _H()
path = os.path.join(root, 'profiles', 'pmc_traffic.json')
out = json.load(open(path)) if os.path.exists(path) else {}
for k, suffix in (('k_raycast_quad<false>', ':sq'), ('k_physics', ':physics:sq')):
    c = tot.get(k)
    if not c or not c.get('SQ_BUSY_CYCLES') or not c.get('SQ_INSTS_VALU'):
        print('no SQ counters for', k)
        continue
    out[key + suffix] = {'kernel': k, 'source_hash': h.hexdigest()[:12], 'date': datetime.date.today().isoformat(),
                         'valu_issue_frac': c['SQ_ACTIVE_INST_VALU'] * 4.0 / (c['SQ_BUSY_CYCLES'] * 32.0),
                         'valu_lane_utilisation': c['SQ_THREAD_CYCLES_VALU'] / (c['SQ_INSTS_VALU'] * 64.0),
                         'salu_per_valu': c.get('SQ_INSTS_SALU', 0.0) / c['SQ_INSTS_VALU'],
                         'wait_frac': c.get('SQ_WAIT_ANY', 0.0) / max(c.get('SQ_WAVE_CYCLES', 0.0), 1.0),
                         'counters': {n: v for n, v in sorted(c.items())}}
    print(key + suffix, {n: v for n, v in out[key + suffix].items() if n != 'counters'})
json.dump(out, open(path, 'w'), indent=1)
