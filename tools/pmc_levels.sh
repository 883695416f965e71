#!/bin/bash
# tools/pmc_levels.sh -- on the GPU box: L1 (TCP) cache-line lookups of k_grid_encode per RANGE OF LEVELS (NRC_ENC_LEVELS switch), to show where the
# 42 lookups per sample come from (round-2 review: "cut L1 tag lookups on levels 0-6 with scalar loads / LDS staging").
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_levels
mkdir -p $O
for range in 0-16 0-5 5-8 8-10 10-12 12-16; do
  export NRC_ENC_LEVELS=$range
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/r_$range -- python3 $R/tools/bench_query.py 2 > $O/r_$range.log 2>&1
done
unset NRC_ENC_LEVELS
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc_levels'
print('| levels | L1 lookups per live sample | L1->L2 requests per sample | VALU instructions per sample | kernel cycles |')
print('|---|---|---|---|---|')
for d in sorted(glob.glob(O + '/r_*/')):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + '*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'k_grid_encode' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    if not acc:
        continue
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    n = 7.69e6   # live samples per 8 Mi-slot launch on the bench poses (bench.py: samples_per_launch)
    print(f"| {os.path.basename(d.rstrip('/'))[2:]} | {m['TCP_TOTAL_CACHE_ACCESSES_sum'] / n:.1f} | {m['TCP_TCC_READ_REQ_sum'] / n:.1f} | {m['SQ_INSTS_VALU'] * 64 / n / 64:.0f} per wave-of-64 / 64 | {m['GRBM_GUI_ACTIVE'] / 8:.0f} |")
PY
