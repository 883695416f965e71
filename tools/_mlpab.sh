cd $GRAFT_REPO_ROOT
python tools/bench_query.py 5 2>&1 | grep "encode kernel" | sed 's/per launch.*whole/whole/'
python tools/bench_query.py 5 2>&1 | grep "encode kernel" | sed 's/per launch.*whole/whole/'
python -m pytest tests/test_gpu_render_parity.py tests/test_gpu_garden_parity.py -q -m gpu -x 2>&1 | tail -2
