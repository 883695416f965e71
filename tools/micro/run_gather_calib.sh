#!/bin/bash
# tools/micro/run_gather_calib.sh -- on the GPU box: the FETCH_SIZE calibration of tools/micro/gather_fetch_calib.hip (plain run + two counter passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=$R/tools/micro/gather_fetch_calib
O=$R/gpurun_out/calib
mkdir -p $O
[ -x $B ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $B $B.hip
$B > $O/run.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $O/tcc -- $B > $O/tcc.log 2>&1
cat $O/run.log
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/calib'
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(k, {n: round(sum(x) / len(x), 1) for n, x in v.items()})
PY
