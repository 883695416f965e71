cd $GRAFT_REPO_ROOT
for u in 0 1 2 3; do
  echo "== NRC_ENC_STORE=$u"
  NRC_ENC_STORE=$u python tools/bench_query.py 5 2>&1 | grep "encode kernel"
done
