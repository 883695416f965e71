#!/bin/bash
# tools/kseq.sh N_LAST SCRIPT ARGS... : rocprofv3 kernel trace of a tool run; prints the last N_LAST kernel launches in order (name, grid, us, gap to the previous end)
cd /tmp && export TMPDIR=/tmp
n=$1; shift
rm -rf /tmp/pkq
rocprofv3 --kernel-trace --output-format csv -d /tmp/pkq -- python3 $GRAFT_REPO_ROOT/$@ 2>&1 | grep -v "rocprofv3\|amdgpu.ids\|Opened result" | tail -2
f=$(find /tmp/pkq -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$n" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); n = int(sys.argv[2])
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-n:]
prev = None
for r in rows:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0.0
    prev = e
    print('%7.1f us  gap %6.1f  grid %9s  %s' % ((e - s) / 1e3, gap, r.get('Grid_Size', r.get('Grid_Size_X', '?')), name[:120]))
print('span us', (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3)
P
