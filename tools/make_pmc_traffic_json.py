#!/usr/bin/env python
"""Turn the two PMC passes of tools/collect_pmc.sh (FETCH_SIZE.txt, WRITE_SIZE.txt) into the JSON bench.py reads for
roofline.traffic.  gfx950 correction as in MI355X_MICROARCH.md: FETCH_SIZE counts 128-B requests as 64 B -> x2; both in KB."""
import json
import os
import sys

d = sys.argv[1]


def table(name):
    out = {}
    for line in open(os.path.join(d, name + '.txt')):
        p = line.split()
        if len(p) >= 5 and p[1] == name:
            out[p[0]] = float(p[3])
    return out


fetch, write = table('FETCH_SIZE'), table('WRITE_SIZE')


def short(k):
    k = k.split('ILi')[0].split('ILb')[0]
    return k[k.index('k_'):] if 'k_' in k else k


tot = {short(k): 2 * fetch.get(k, 0.0) * 1024 + write.get(k, 0.0) * 1024 for k in set(fetch) | set(write)}
dom = max((k for k in tot if 'expm' in k), key=lambda k: tot[k])
raw = [k for k in fetch if short(k) == dom][0]
js = {'workload': {'n': 32, 'k': 4, 'steps': 500, 'm': 8, 'taylor': [5, 3], 'seeds_per_gpu': 64, 'chunks': 16},
      'kernel': dom, 'FETCH_SIZE_KB': fetch[raw], 'WRITE_SIZE_KB': write[raw], 'hbm_bytes_per_launch': int(tot[dom]),
      'correction': '2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts 128-B requests as 64 B)',
      'command': 'tools/collect_pmc.sh: rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate passes) '
                 '-- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-single',
      'other_kernels_bytes_per_launch': {k: int(v) for k, v in sorted(tot.items(), key=lambda kv: -kv[1]) if k != dom and v > 1e6}}
json.dump(js, open(os.path.join(d, 'pmc_traffic.json'), 'w'), indent=1)
print(json.dumps(js, indent=1))
