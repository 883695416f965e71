cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_adam_parity.py tests/test_gpu_graphs.py tests/test_gpu_convergence.py -q -m gpu 2>&1 | tail -3
python tools/bench_train_graph.py 2200 100 2>&1 | grep recorded
python tools/bench_train.py 2200 60 2>&1 | grep "train iteration"
