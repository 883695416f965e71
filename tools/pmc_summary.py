#!/usr/bin/env python3
"""tools/pmc_summary.py -- averages rocprofv3 --pmc counter_collection.csv files per kernel (developer tool)."""
import csv, glob, sys, collections, re
root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out'
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'{root}/pmc_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        name = name.split('(')[0].replace('void ', '')
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    if not any(x in k for x in ('k_grid_encode', 'k_ngp_mlp', 'k_render', 'k_composite')):
        continue
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f'    {c:40s} n={len(v):4d} mean={sum(v)/len(v):16.1f}')
