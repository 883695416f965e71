import sys, torch
sys.path.insert(0, '.')
from tests.test_gpu_graphs import _rays, _train_pair
from nerficg_amd.apex_optimizers import FusedAdam
from nerficg_amd.graphs import instant_ngp_iteration
DEV = 'cuda'
cam, o, d = _rays()
n = 2048
model, renderer, scaler = _train_pair(seed=4)
opt = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False, capturable=True)
target = torch.tensor([0.8, 0.3, 0.1], device=DEV).expand(n, 3).contiguous()
step = instant_ngp_iteration(model, renderer, opt, scaler, cam, n_rays=n, sample_capacity=400_000)
step(origin=o[:n].contiguous(), view_direction=d[:n].contiguous(), rgb=target)
# capture by hand with debug mode
step.calls += 1
step.graph = torch.cuda.CUDAGraph()
step.graph.enable_debug_mode()
with torch.cuda.graph(step.graph, stream=step.stream):
    step.outputs = step.body(**step.inputs)
step.graph.debug_dump('/tmp/ingp_graph.dot')
print('dumped')
