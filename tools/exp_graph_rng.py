import sys, torch
sys.path.insert(0, '.')
from nerficg_amd.graphs import GraphedIteration
dev = 'cuda'
w = torch.zeros(1000, device=dev, requires_grad=True)
def body(x):
    r = torch.rand(1000, device=dev)
    loss = ((w * x - r) ** 2).sum()
    loss.backward()
    with torch.no_grad():
        w.sub_(0.01 * w.grad); w.grad = None
    return {'loss': loss.detach(), 'r0': r[:3].clone()}
st = GraphedIteration(body, {'x': torch.ones(1000, device=dev)})
for i in range(50):
    out = st(x=torch.full((1000,), 1.0 + i, device=dev))
    if i % 10 == 0 or i < 4:
        print(i, float(out['loss']), out['r0'].tolist(), st.recorded)
torch.cuda.synchronize(); print('ok')
