#!/bin/bash
# tools/kstats.sh VARIANT SCRIPT ARGS... : rocprofv3 kernel stats of a tool run with _ab/VARIANT.so, prints name,calls,avg_us for the top kernels
cd /tmp && export TMPDIR=/tmp
v=$1; shift
rm -rf /tmp/p_$v
NRC_LIB_PATH=$GRAFT_REPO_ROOT/_ab/$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$v -- python3 $GRAFT_REPO_ROOT/$@ > /dev/null 2>&1
f=$(find /tmp/p_$v -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    print(f"{n[:40]:40s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
P
