import sys, numpy as np, torch
sys.path.insert(0, '.')

from nerficg_amd import _lib
import oracle
lib = _lib.load()
PLS = float(np.exp(np.log(2048 / 16) / 15))
grid = dict(n_levels=16, log2_hashmap_size=19, base_resolution=16, per_level_scale=PLS)
total, offsets, _, _ = oracle.grid_layout(**grid)
rng = np.random.default_rng(0)
n_rays, per = 400, 100
o = rng.random((n_rays, 1, 3)) * 0.3 + 0.1
d = rng.normal(size=(n_rays, 1, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
t = (np.arange(per)[None, :, None] * (1.7 / 1024))
x = np.clip(o + np.abs(d) * t, 0, 1).reshape(-1, 3).astype(np.float32)
m = x.shape[0]
g = rng.normal(size=(16, m, 2)).astype(np.float32)
g[:, rng.random(m) < 0.2] = 0
tx, tg = torch.tensor(x).cuda(), torch.tensor(g).cuda()
ws = torch.empty(int(lib.nrc_grid_backward_ws_bytes(m, 16, 19, 16, PLS)), dtype=torch.uint8, device='cuda')
def run(w):
    out = torch.zeros(total, 2, device='cuda')
    _lib.check(lib.nrc_grid_backward(_lib.ptr(tx), m, _lib.ptr(tg), 1, 16, 19, 16, PLS, _lib.ptr(out), _lib.ptr(w), _lib.stream_of(out)), 'gb')
    return out.cpu().numpy()
want = oracle.grid_encode_bw(x, np.ascontiguousarray(g.transpose(1, 0, 2).reshape(m, 32)), total, **grid)
for name, w in (('bucket', ws), ('plain', None)):
    a = run(w)
    for l in range(16):
        sl = slice(offsets[l], offsets[l + 1])
        print(name, l, float(np.abs(a[sl] - want[sl]).max() / np.abs(want[sl]).max()))
