cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_tcnn_parity.py tests/test_gpu_render_parity.py tests/test_gpu_fused_training_ops.py tests/test_gpu_convergence.py tests/test_gpu_graphs.py -q -m gpu 2>&1 | tail -3
python tools/bench_train_graph.py 2200 100 2>&1 | grep recorded
python - <<'PY'
import sys; sys.path.insert(0,'.')
import torch, bench
dev=torch.device('cuda',0)
model, renderer, cam, poses = bench.build_scene(dev)
r=bench.time_train(model, renderer, cam, poses, iters=40)
print(r['roofline']['_per_kernel_ms'], r['ms_per_iteration'], r['hip_graph']['ms_per_iteration'])
PY
