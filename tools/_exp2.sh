cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ "$1" == "test" ]; then shift; python3 -m pytest tests/test_gpu_gs_parity.py tests/test_gpu_baseline_size_parity.py -x -q -k "not ingp and not raymarching" 2>&1 | tail -5; fi
sizes=${SIZES:-"1000000 6000000"}
for v in "$@"; do
  if [ "$v" == "head" ]; then unset NRC_LIB_PATH; else export NRC_LIB_PATH=$GRAFT_REPO_ROOT/_ab/$v.so; fi
  for n in $sizes; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/exp_${v}_$n -o x -- python3 tools/exp_gs.py $n 10 > gpurun_out/exp_${v}_$n.log 2>&1
  echo "== $v $n"; grep "^n=" gpurun_out/exp_${v}_$n.log | cut -c1-60
  f=$(find gpurun_out/exp_${v}_$n -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
out=[]
for r in rows:
    n=r['Name']
    if 'GsCam' in n or any(k in n for k in ('k_bin','k_radix','k_tile','k_scan','k_depth','k_span','k_item','k_rects')):
        short=n.split('::')[-1].split('(')[0]
        if 'GsCam' in n and 'k_' not in short:
            na=n.count(',')
            short={11:'k_render',15:'k_render_bw',16:'k_preprocess',17:'k_preprocess_bw'}.get(na,'k?%d'%na)
        out.append(f"{short} {float(r['AverageNs'])/1e3:.1f}")
print('   '+' | '.join(out))
PY
  done
done
