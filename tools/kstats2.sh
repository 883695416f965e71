#!/bin/bash
# tools/kstats2.sh DIVISOR SCRIPT ARGS... : rocprofv3 kernel stats of a tool run (default library); per-iteration launches and microseconds
cd /tmp && export TMPDIR=/tmp
div=$1; shift
rm -rf /tmp/pks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pks -- python3 $GRAFT_REPO_ROOT/$@ 2>&1 | grep -v "rocprofv3\|amdgpu.ids" | tail -2
f=$(find /tmp/pks -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$div" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); div = float(sys.argv[2])
tot = 0
for r in rows:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    c = int(r['Calls']) / div
    t = float(r['TotalDurationNs']) / div / 1e3
    tot += t
    if t > 2.0:
        print('%5.2f x %7.1f = %6.1f  %s' % (c, float(r['AverageNs']) / 1e3, t, n[:110]))
print('total us per iteration', round(tot, 1), 'launches', round(sum(int(r['Calls']) for r in rows) / div, 1))
P
