import sys
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of the bench command into profiles/pmc_traffic.json.

HBM bytes = 2 * FETCH_SIZE(KB) * 1024 + WRITE_SIZE(KB) * 1024.  The factor 2 on FETCH_SIZE is the gfx950
correction of MI355X_MICROARCH.md (requests of 128 B tallied at 64 B), re-calibrated for THIS access pattern
with tools/calib_fetch.hip: 268 M scattered 16-B reads report 64 B each while running at the 128-B-line rate of
the streaming peak (6.3 TB/s), and a 4 GiB coalesced stream reports 2 GiB.  WRITE_SIZE is taken as is.
usage: pmc_traffic.py FETCH_DIR WRITE_DIR key [kernel-name-substring, default k_raycast_quad]
"""
import csv, glob, json, os, sys
fetch_dir, write_dir, key = sys.argv[1:4]
kernel = sys.argv[4] if len(sys.argv) > 4 else 'k_raycast_quad'
def total(d, counter, name):
    tot, n = 0.0, 0
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter and name in r['Kernel_Name']:
                tot += float(r['Counter_Value']); n += 1
    return tot, n
out = {}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'pmc_traffic.json')
if os.path.exists(path):
    out = json.load(open(path))
f, nf = total(fetch_dir, 'FETCH_SIZE', kernel)
w, nw = total(write_dir, 'WRITE_SIZE', kernel)
import datetime, hashlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import kernel_source_hash      # (the one definition of what identifies a kernel build)
class _H(object):
    def hexdigest(self): return kernel_source_hash()
h = _H()
out[key] = {'kernel': kernel, 'launches': nf, 'fetch_size_kb_sum': f, 'write_size_kb_sum': w,
            'source_hash': h.hexdigest()[:12], 'date': datetime.date.today().isoformat(),
            'hbm_bytes_per_launch': (2.0 * f + w) * 1024.0 / max(nf, 1),
            'correction': 'FETCH_SIZE x2 (gfx950, calibrated with tools/calib_fetch.hip), WRITE_SIZE x1'}
json.dump(out, open(path, 'w'), indent=1)
print(key, out[key])
