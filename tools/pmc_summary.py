"""Summarise rocprofv3 --pmc CSV output for step_kernel (mean per dispatch, and per wave)."""
import collections
import csv
import glob
import os
import sys

KERNEL = os.environ.get('PMC_KERNEL', 'step_kernel')       # substring of the kernel name to summarise

for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        meta = None
        for r in csv.DictReader(open(f)):
            if KERNEL in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
                meta = r
        waves = float(meta['Grid_Size']) / 64 if meta else 1
        print(f, 'VGPR', meta and meta['VGPR_Count'], 'scratch', meta and meta['Scratch_Size'])
        for k in sorted(acc):
            v = acc[k]
            m = sum(v) / len(v)
            print('  %-26s mean/dispatch %.4g   per wave %.1f' % (k, m, m / waves))
