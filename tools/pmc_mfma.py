#!/usr/bin/env python3
"""Per-kernel matrix-core utilisation from one rocprofv3 PMC pass (`--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
GRBM_GUI_ACTIVE`, --kernel-trace only) -> JSON for profiles/.
  mfma_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 4 / SQ_BUSY_CU_CYCLES   (busy cycles of the matrix pipes, summed over the
                     4 SIMDs of every CU, against the cycles the CUs had work: MI355X_MICROARCH.md, counter units)
  clock_ghz        = GRBM_GUI_ACTIVE / 8 / kernel duration (sum over the 8 XCDs).  GRBM_GUI_ACTIVE keeps counting while the
                     dispatch is set up and drained, so on short dispatches the quotient exceeds what the part can clock
                     (max 2.4 GHz): values above it are not written — the row gets `clock_ghz: null` and
                     `clock_note: "non-physical (<value>): dispatch too short for GRBM_GUI_ACTIVE / duration"` instead
usage: tools/pmc_mfma.py <pmc_dir> <note> > profiles/rNN_mfma_util_pmc.json"""
import collections
import csv
import glob
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
dur = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        acc[name][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_BUSY_CU_CYCLES':
            n[name] += 1
            if 'Start_Timestamp' in r and 'End_Timestamp' in r:
                dur[name] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
out = {'note': sys.argv[2] if len(sys.argv) > 2 else '', 'kernels': {}}
for k in sorted(acc, key=lambda k: -acc[k].get('SQ_BUSY_CU_CYCLES', 0.0)):
    a = acc[k]
    if not n[k] or not a.get('SQ_BUSY_CU_CYCLES'):
        continue
    e = {'launches': n[k], 'SQ_VALU_MFMA_BUSY_CYCLES_per_launch': round(a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / n[k]),
         'SQ_BUSY_CU_CYCLES_per_launch': round(a['SQ_BUSY_CU_CYCLES'] / n[k]),
         'mfma_utilisation': round(a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / 4.0 / a['SQ_BUSY_CU_CYCLES'], 4)}
    if dur[k] and a.get('GRBM_GUI_ACTIVE'):
        e['avg_us_under_the_profiler'] = round(dur[k] / n[k] / 1e3, 1)
        ghz = a['GRBM_GUI_ACTIVE'] / 8.0 / dur[k]
        if ghz <= 2.4:  # (the part's maximum clock)
            e['clock_ghz'] = round(ghz, 2)
        else:
            e['clock_ghz'] = None
            e['clock_note'] = f'non-physical ({ghz:.2f}): dispatch too short for GRBM_GUI_ACTIVE / duration'
    out['kernels'][k] = e
json.dump(out, sys.stdout, indent=1)
