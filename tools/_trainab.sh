cd $GRAFT_REPO_ROOT
for cfg in "NRC_BWD_BLOCKS=256" "NRC_BWD_BLOCKS=512" "NRC_BWD_BLOCKS=1024" "NRC_BWD_BLOCKS=128"; do
  echo "== $cfg"
  env $cfg python - <<'PY'
import sys; sys.path.insert(0,'.')
import torch, bench
dev=torch.device('cuda',0)
model, renderer, cam, poses = bench.build_scene(dev)
r=bench.time_train(model, renderer, cam, poses, iters=40)
print({k:v for k,v in r['roofline']['_per_kernel_ms'].items() if 'nwie' in k}, r['ms_per_iteration'], r['hip_graph']['ms_per_iteration'])
PY
done
