cd $GRAFT_REPO_ROOT
echo "== default"; python tools/bench_query.py 5 2>&1 | grep "encode kernel" | sed 's/per launch.*whole/whole/'
for a in 1 2 16 17 18; do
  echo "== fine levels >= 12 aux=$a"; NRC_LIB_PATH=_ab/enc_aux$a.so python tools/bench_query.py 5 2>&1 | grep "encode kernel" | sed 's/per launch.*whole/whole/'
done
