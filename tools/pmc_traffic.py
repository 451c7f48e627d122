#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from two rocprofv3 PMC passes (one `--pmc FETCH_SIZE`, one `--pmc WRITE_SIZE`, each
with --kernel-trace only) -> JSON for profiles/.  Units and the gfx950 correction follow MI355X_MICROARCH.md's
HBM / rocprofv3 section: both counters are in KiB; FETCH_SIZE tallies 128-B requests at 64 B on gfx950, so it is
doubled.  The counters sit at the L2 <-> fabric boundary: Infinity-Cache hits are included.
usage: tools/pmc_traffic.py <fetch_dir> <write_dir> <note> > profiles/rNN_hbm_traffic_pmc.json"""
import collections
import csv
import glob
import json
import sys


def collect(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            a = acc[name]
            a[0] += 1
            a[1] += float(r['Counter_Value'])
    return acc


fetch, write = collect(sys.argv[1], 'FETCH_SIZE'), collect(sys.argv[2], 'WRITE_SIZE')
out = {'note': sys.argv[3] if len(sys.argv) > 3 else '', 'kernels': {}}
for k in sorted(fetch, key=lambda k: -fetch[k][1]):
    n = fetch[k][0]
    f_kb = fetch[k][1] / n
    w_kb = write[k][1] / write[k][0] if k in write and write[k][0] else 0.0
    out['kernels'][k] = {'launches': n, 'FETCH_SIZE_KB_per_launch': round(f_kb, 1),
                         'WRITE_SIZE_KB_per_launch': round(w_kb, 1),
                         'hbm_bytes_per_launch_corrected': int((2 * f_kb + w_kb) * 1024)}
json.dump(out, sys.stdout, indent=1)
