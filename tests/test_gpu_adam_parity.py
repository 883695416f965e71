"""GPU: nerficg_amd.apex_optimizers.FusedAdam (HIP, C ABI group 8) against oracle/adam_oracle.c, with and without torch.amp.GradScaler,
and under the reference's optimizer-state surgery (src/Optim/adam_utils.py)."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


@pytest.mark.parametrize('n', [1, 7, 4096, 100003])
def test_fused_adam_matches_oracle(n):
    from nerficg_amd.apex_optimizers import FusedAdam
    rng = np.random.default_rng(n)
    p0 = rng.normal(size=n).astype(np.float32)
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()).to(DEV))
    opt = FusedAdam([tp], lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False)  # Trainer.py:35
    p, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for step in range(1, 5):
        g = (rng.normal(size=n) * 10.0 ** rng.integers(-6, 1, size=n)).astype(np.float32)
        tp.grad = torch.from_numpy(g.copy()).to(DEV)
        opt.step()
        p, m, v = oracle.adam_step(p, g, m, v, step, 1e-2, (0.9, 0.99), 1e-15)
        np.testing.assert_allclose(tp.detach().cpu().numpy(), p, rtol=1e-6, atol=1e-7)  # same f32 operation order; sqrt/div correctly rounded
    st = opt.state[tp]
    np.testing.assert_allclose(st['exp_avg'].cpu().numpy(), m, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(st['exp_avg_sq'].cpu().numpy(), v, rtol=1e-6, atol=1e-20)
    assert opt.param_groups[0]['step'] == 4
    opt.zero_grad()
    assert tp.grad is None


def test_fused_adam_on_a_misaligned_view():
    from nerficg_amd.apex_optimizers import FusedAdam
    rng = np.random.default_rng(3)
    base = torch.from_numpy(rng.normal(size=1001).astype(np.float32)).to(DEV)
    tp = torch.nn.Parameter(base[1:])  # data pointer 4 bytes past a 16-byte boundary
    p0 = tp.detach().cpu().numpy().copy()
    opt = FusedAdam([tp], lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False)
    g = rng.normal(size=1000).astype(np.float32)
    tp.grad = torch.from_numpy(g).to(DEV)
    opt.step()
    p, _, _ = oracle.adam_step(p0, g, np.zeros_like(p0), np.zeros_like(p0), 1, 1e-2, (0.9, 0.99), 1e-15)
    np.testing.assert_allclose(tp.detach().cpu().numpy(), p, rtol=1e-6, atol=1e-7)


def test_fused_adam_with_grad_scaler_and_inf_skip():
    from nerficg_amd.apex_optimizers import FusedAdam
    rng = np.random.default_rng(1)
    p0 = rng.normal(size=5000).astype(np.float32)
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()).to(DEV))
    opt = FusedAdam([tp], lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False)
    scaler = torch.amp.GradScaler(init_scale=128.0, growth_interval=10 ** 9)  # Trainer.py:44
    x = torch.from_numpy(rng.normal(size=5000).astype(np.float32)).to(DEV)
    loss = (tp * x).sum()
    scaler.scale(loss).backward()
    scaler.step(opt); scaler.update(); opt.zero_grad()
    p, m, v = oracle.adam_step(p0, x.cpu().numpy() * 128.0, np.zeros_like(p0), np.zeros_like(p0), 1, 1e-2, (0.9, 0.99), 1e-15, grad_scale=128.0)
    np.testing.assert_allclose(tp.detach().cpu().numpy(), p, rtol=1e-6, atol=1e-7)
    before = tp.detach().clone()
    loss = (tp * x).sum() * float('inf')
    scaler.scale(loss).backward()
    scaler.step(opt); scaler.update(); opt.zero_grad()
    assert torch.equal(tp.detach(), before) and scaler.get_scale() == 64.0  # step skipped on the device, scale backed off
    # the skipped call must not count (apex: the scaler never calls step() on an overflow): the next step is step 2 of the oracle
    assert opt.param_groups[0]['step'] == 2 and opt.effective_step(opt.param_groups[0]) == 1
    loss = (tp * x).sum()
    scaler.scale(loss).backward()
    scaler.step(opt); scaler.update(); opt.zero_grad()
    p2, _, _ = oracle.adam_step(p, x.cpu().numpy() * 64.0, m, v, 2, 1e-2, (0.9, 0.99), 1e-15, grad_scale=64.0)
    np.testing.assert_allclose(tp.detach().cpu().numpy(), p2, rtol=2e-6, atol=2e-7)
    assert opt.effective_step(opt.param_groups[0]) == 2


def test_fused_adam_bumps_the_version_counter_and_feeds_the_tinycudann_half_copy():
    """The step kernel writes through a raw pointer: caches keyed on the parameter version (nerficg_amd.tinycudann) must still see it."""
    from nerficg_amd.apex_optimizers import FusedAdam
    from nerficg_amd.tinycudann import NetworkWithInputEncoding
    net = NetworkWithInputEncoding(3, 16, {'otype': 'Grid', 'type': 'Hash', 'n_levels': 16, 'n_features_per_level': 2, 'log2_hashmap_size': 15,
                                           'base_resolution': 16, 'per_level_scale': 1.5},
                                   {'otype': 'FullyFusedMLP', 'activation': 'ReLU', 'output_activation': 'None', 'n_neurons': 64, 'n_hidden_layers': 1},
                                   seed=3).to(DEV)
    opt = FusedAdam(net.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False)
    x = torch.rand(1000, 3, device=DEV)
    y0 = net(x).float()
    v0 = net.params._version
    y0.square().sum().backward()
    opt.step()
    assert net.params._version > v0
    assert torch.equal(net._half_params(), net.params.detach().half())  # written by the step kernel, same rounding as the conversion pass
    assert not torch.equal(net(x).float(), y0.detach())
    # a parameter without a module behind it: version bump only
    q = torch.nn.Parameter(torch.rand(10, device=DEV)); q.grad = torch.ones_like(q)
    o2 = FusedAdam([q], lr=1e-2); v = q._version; o2.step()
    assert q._version > v


def test_param_group_surgery_like_adam_utils():
    """prune_param_groups / extend_param_groups (adam_utils.py:21-61) rebuild the parameter and move the state entry."""
    from nerficg_amd.apex_optimizers import FusedAdam
    a = torch.nn.Parameter(torch.rand(100, 3, device=DEV))
    opt = FusedAdam([{'params': [a], 'lr': 1e-3, 'name': 'positions'}], lr=0.0, eps=1e-15, adam_w_mode=False)  # Model.py:133
    a.grad = torch.rand_like(a); opt.step()
    mask = torch.arange(100, device=DEV) % 2 == 0
    group = opt.param_groups[0]
    state = opt.state[a]
    new = torch.nn.Parameter(torch.cat((a[mask], torch.rand(10, 3, device=DEV))))
    for k in ('exp_avg', 'exp_avg_sq'):
        state[k] = torch.cat((state[k][mask], torch.zeros(10, 3, device=DEV)))
    opt.state.pop(a); opt.state[new] = state; group['params'][0] = new
    new.grad = torch.rand_like(new); opt.step()
    assert group['step'] == 2 and torch.isfinite(new).all() and opt.state[new]['exp_avg'].shape == (60, 3)


# ------------------------------------------------------------------------------------------------ GradScaler's inf check (nerficg_amd.amp)
@pytest.mark.parametrize('n,offset', [(1, 0), (3, 0), (4099, 0), (1 << 20, 0), (70001, 1), (12_196_240, 0)])
def test_streaming_nonfinite_check_equals_torch(n, offset):
    """nrc_nonfinite_check against torch.isfinite: clean tensors leave the flag at 0; one inf / NaN anywhere -- first element, last element, the
    unaligned tail, a view that starts 4 bytes into an allocation -- sets it."""
    import ctypes
    from nerficg_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device='cuda').manual_seed(n)
    base = torch.randn(n + offset, device='cuda', generator=g) * 1e30   # large finite values: 0x7f7fffff-ish exponents must not trip the test
    base.clamp_(-3e38, 3e38)
    t = base[offset:]
    flag = torch.zeros((), device='cuda')
    _lib.check(lib.nrc_nonfinite_check(_lib.ptr(t), n, _lib.ptr(flag), _lib.stream_of(t)), 'nonfinite_check')
    assert float(flag) == 0.0 and bool(torch.isfinite(t).all())
    for pos, val in ((0, float('inf')), (n - 1, float('nan')), (n // 2, -float('inf')), (max(0, n - 2), float('nan'))):
        u = t.clone() if offset == 0 else base.clone()[offset:]
        u[pos] = val
        flag.zero_()
        _lib.check(lib.nrc_nonfinite_check(_lib.ptr(u), n, _lib.ptr(flag), _lib.stream_of(u)), 'nonfinite_check')
        assert float(flag) == 1.0, (pos, val)


def test_fast_grad_scaler_follows_torch_grad_scaler():
    """nerficg_amd.amp.GradScaler and torch.amp.GradScaler through five FusedAdam steps, the third with an overflowing gradient: same parameters,
    same scale, same skipped step."""
    from nerficg_amd.amp import GradScaler
    from nerficg_amd.apex_optimizers import FusedAdam
    results = []
    for cls in (torch.amp.GradScaler, GradScaler):
        torch.manual_seed(0)
        p = torch.nn.Parameter(torch.randn(100_003, device='cuda'))
        q = torch.nn.Parameter(torch.randn(7, 5, device='cuda'))
        opt = FusedAdam([{'params': [p]}, {'params': [q]}], lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False)
        scaler = cls('cuda', init_scale=128.0, growth_interval=2)
        trace = []
        for it in range(5):
            x = torch.full_like(p, 1.0 + it)
            if it == 2:
                x[77] = float('inf')
            loss = (p * x).sum() + (q * q).sum()
            scaler.scale(loss).backward()
            scaler.step(opt); scaler.update(); opt.zero_grad()
            trace.append((p.detach().clone(), q.detach().clone(), float(scaler.get_scale())))
        results.append(trace)
    for (pa, qa, sa), (pb, qb, sb) in zip(*results):
        assert torch.equal(pa, pb) and torch.equal(qa, qb) and sa == sb
    assert torch.equal(results[1][2][0], results[1][1][0])   # the overflowing iteration changed nothing
    assert results[1][2][2] == 0.5 * results[1][1][2]         # and halved the scale


def test_load_state_dict_keeps_the_live_moment_tensors():
    """A recorded iteration holds raw pointers to exp_avg / exp_avg_sq: load_state_dict must fill the LIVE tensors instead of replacing them
    (advisor finding of round 4), for the capturable and the plain mode."""
    from nerficg_amd.apex_optimizers import FusedAdam
    for capturable in (False, True):
        p = torch.nn.Parameter(torch.randn(5000, device=DEV))
        opt = FusedAdam([p], lr=1e-2, capturable=capturable)
        for _ in range(3):
            p.grad = torch.randn_like(p)
            opt.step()
        ptrs = (opt.state[p]['exp_avg'].data_ptr(), opt.state[p]['exp_avg_sq'].data_ptr())
        sd = {k: (v if k != 'state' else {i: {n: t.clone() for n, t in st.items()} for i, st in v.items()}) for k, v in opt.state_dict().items()}
        want = sd['state'][0]['exp_avg'].clone()
        p.grad = torch.randn_like(p)
        opt.step()                                      # the live moments move on ...
        assert not torch.equal(opt.state[p]['exp_avg'], want)
        opt.load_state_dict(sd)                         # ... and come back through the load, in the same storage
        assert (opt.state[p]['exp_avg'].data_ptr(), opt.state[p]['exp_avg_sq'].data_ptr()) == ptrs
        assert torch.equal(opt.state[p]['exp_avg'], want)
        assert opt.effective_step(opt.param_groups[0]) == 3


def test_scaler_shared_by_two_optimizers_follows_torch_and_the_fused_step_refuses_a_second_one():
    """nerficg_amd.amp.GradScaler(single_optimizer=False) around TWO FusedAdam optimizers: a clean step, an overflow in the second optimizer's gradients (torch
    skips per optimizer and backs the scale off once), then growth -- the same parameters, scale and step counts as
    torch.amp.GradScaler.  The default scaler takes the fused step + update for the first optimizer of an iteration and raises when a second one follows."""
    from nerficg_amd.amp import GradScaler
    from nerficg_amd.apex_optimizers import FusedAdam
    rng = np.random.default_rng(11)
    a0, b0 = rng.normal(size=3000).astype(np.float32), rng.normal(size=777).astype(np.float32)
    xs = [(rng.normal(size=3000).astype(np.float32), rng.normal(size=777).astype(np.float32)) for _ in range(4)]

    def run(make_scaler):
        pa, pb = torch.nn.Parameter(torch.from_numpy(a0.copy()).to(DEV)), torch.nn.Parameter(torch.from_numpy(b0.copy()).to(DEV))
        oa = FusedAdam([pa], lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False)
        ob = FusedAdam([pb], lr=3e-3, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False)
        scaler = make_scaler()
        scales = []
        for it, (xa, xb) in enumerate(xs):
            loss = (pa * torch.from_numpy(xa).to(DEV)).sum() + (pb * torch.from_numpy(xb).to(DEV)).sum()
            scaler.scale(loss).backward()
            if it == 1:
                pb.grad[5] = float('inf')
            scaler.step(oa); scaler.step(ob); scaler.update()
            oa.zero_grad(); ob.zero_grad()
            scales.append(float(scaler.get_scale()))
        return pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), scales, oa.effective_step(oa.param_groups[0]), ob.effective_step(ob.param_groups[0])

    ref = run(lambda: torch.amp.GradScaler(init_scale=128.0, growth_interval=2))
    got = run(lambda: GradScaler(init_scale=128.0, growth_interval=2, single_optimizer=False))
    assert got[2] == ref[2] == [128.0, 64.0, 64.0, 128.0]
    assert got[3:] == ref[3:] == (4, 3)
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])
    with pytest.raises(RuntimeError, match='single_optimizer=False'):
        run(lambda: GradScaler(init_scale=128.0, growth_interval=2))


@pytest.mark.parametrize('n_t', [40_000, 40_002])      # 40 002: the second shard starts 4 bytes off a 16-byte boundary (8 ranks cut the shipped table at odd multiples of 1 524 530)
@pytest.mark.parametrize('overflow', [False, True])
def test_settle_plus_adam_slices_equal_the_one_call_step_bit_for_bit(overflow, n_t):
    """C-ABI group 14 against group 8 on the SAME gradients: nrc_amp_adam_step on the mean of two ranks' gradients (what a replica does behind
    parallel.allreduce_flat(average=True)) and nrc_amp_settle(divisor 2) + nrc_amp_adam_slices on their SUM, MLP weights and the two table shards as separate
    launches (what the sharded ranks do between them): parameters, moments, fp16 copies, step counter, scale and growth tracker bit for bit -- also on an
    overflow, where nothing but the scale may change."""
    from nerficg_amd import _lib
    lib, p = _lib.load(), _lib.ptr
    gen = torch.Generator(device=DEV).manual_seed(5)
    n_mlp, n_c = 3072, 7168
    n_a = n_mlp + n_t
    l2 = 2.0 * 0.5 / (n_mlp + n_c)

    def state():
        g2 = torch.Generator(device=DEV).manual_seed(9)
        S = dict(pa=torch.randn(n_a, device=DEV, generator=g2), pb=torch.randn(n_c, device=DEV, generator=g2))
        for k, n in (('a', n_a), ('b', n_c)):
            S['m' + k], S['v' + k] = torch.rand(n, device=DEV, generator=g2) * 1e-3, torch.rand(n, device=DEV, generator=g2) * 1e-6
            S['h' + k] = S['p' + k].half()
        S.update(step=torch.full((1,), 3, dtype=torch.int32, device=DEV), bc=torch.zeros(2, device=DEV), scale=torch.full((1,), 128.0, device=DEV),
                 tracker=torch.full((1,), 1, dtype=torch.int32, device=DEV), state4=torch.zeros(4, device=DEV), ticket=torch.zeros(17 * 16, dtype=torch.int32, device=DEV))
        return S
    g0a, g1a = torch.randn(n_a, device=DEV, generator=gen) * 128, torch.randn(n_a, device=DEV, generator=gen) * 128
    g0b, g1b = torch.randn(n_c, device=DEV, generator=gen) * 128, torch.randn(n_c, device=DEV, generator=gen) * 128
    if overflow:
        g1a[n_mlp + 11] = float('inf')
    hyper = (1e-2, None, 0.9, 0.99, 1e-15, 0.0, 0)
    A = state()
    mean_a, mean_b = ((g0a + g1a) / 2).contiguous(), ((g0b + g1b) / 2).contiguous()
    _lib.check(lib.nrc_amp_adam_step(p(A['pa']), p(mean_a), p(A['ma']), p(A['va']), p(A['ha']), n_a, l2, n_mlp, p(A['pb']), p(mean_b), p(A['mb']), p(A['vb']), p(A['hb']), n_c, l2, n_c,
                                     *hyper, p(A['step']), p(A['bc']), p(A['scale']), p(A['tracker']), 2.0, 0.5, 2, p(A['state4']), p(A['ticket']), None, _lib.stream_of(mean_a)), 'amp_adam_step')
    B = state()
    sum_a, sum_b = (g0a + g1a).contiguous(), (g0b + g1b).contiguous()
    flag = torch.tensor([1.0 if overflow else 0.0], device=DEV)      # the sum of the ranks' producer flags
    s = _lib.stream_of(sum_a)
    _lib.check(lib.nrc_amp_settle(p(flag), 2.0, 0.9, 0.99, p(B['step']), p(B['bc']), p(B['scale']), p(B['tracker']), 2.0, 0.5, 2, p(B['state4']), None, s), 'amp_settle')
    tail = (*hyper, p(B['bc']), p(B['state4']), s)
    _lib.check(lib.nrc_amp_adam_slices(p(B['pa']), p(sum_a), p(B['ma']), p(B['va']), p(B['ha']), n_mlp, l2, n_mlp, p(B['pb']), p(sum_b), p(B['mb']), p(B['vb']), p(B['hb']), n_c, l2, n_c, *tail), 'slices')
    at = lambda t, off: __import__('ctypes').c_void_p(t.data_ptr() + off * t.element_size())
    half = n_t // 2
    for shard in (1, 0):          # "rank 1" first: the order must not matter
        b = n_mlp + shard * half
        _lib.check(lib.nrc_amp_adam_slices(at(B['pa'], b), at(sum_a, b), at(B['ma'], b), at(B['va'], b), at(B['ha'], b), half, 0.0, 0, None, None, None, None, None, 0, 0.0, 0, *tail), 'slices')
    torch.cuda.synchronize()
    for k in ('pa', 'pb', 'ma', 'mb', 'va', 'vb', 'ha', 'hb', 'step', 'scale', 'tracker'):
        assert torch.equal(A[k], B[k]), k
    assert int(A['step']) == (3 if overflow else 4) and float(A['scale']) == (64.0 if overflow else 256.0)
    fresh = state()
    assert torch.equal(A['pa'], fresh['pa']) == overflow


def test_multi_tensor_step_equals_the_per_tensor_launches_bit_for_bit(monkeypatch):
    """Six single-tensor groups with their own learning rates (src/Methods/GaussianSplatting/Model.py:121-138) through nrc_adam_step_multi (one launch) and
    through six nrc_adam_step launches: the same parameters and moments bit for bit over three steps, one group without a gradient in step 2 (its step counter
    must not advance), odd sizes and a misaligned view among them."""
    from nerficg_amd.apex_optimizers import FusedAdam
    shapes = [(1000, 3), (1000, 1, 3), (1000, 15, 3), (1000, 1), (1000, 3), (1001,)]
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 5e-2, 5e-3, 1e-3]

    def run(multi):
        gen = torch.Generator(device=DEV).manual_seed(4)
        base = torch.randn(1002, device=DEV, generator=gen)
        params = [torch.nn.Parameter(torch.randn(s, device=DEV, generator=gen)) for s in shapes[:-1]] + [torch.nn.Parameter(base[1:])]   # the last one 4 bytes off a 16-byte boundary
        opt = FusedAdam([{'params': [p], 'lr': lr, 'name': str(k)} for k, (p, lr) in enumerate(zip(params, lrs))], lr=0.0, eps=1e-15)
        if not multi:
            monkeypatch.setattr(opt, '_step_multi', lambda lib: False)
        calls = []
        real = opt._step_multi
        if multi:
            monkeypatch.setattr(opt, '_step_multi', lambda lib: (calls.append(1), real(lib))[1])
        for it in range(3):
            for k, p in enumerate(params):
                p.grad = None if (it == 1 and k == 3) else torch.randn(p.shape, device=DEV, generator=gen) * 10.0 ** (k - 3)
            opt.step()
        assert (len(calls) == 3) == multi
        return [p.detach().clone() for p in params], [opt.state[p][n].clone() for p in params for n in ('exp_avg', 'exp_avg_sq')], [g['step'] for g in opt.param_groups]

    a, b = run(True), run(False)
    assert a[2] == b[2] == [3, 3, 3, 2, 3, 3]
    for x, y in zip(a[0] + a[1], b[0] + b[1]):
        assert torch.equal(x, y)


def test_wire_pack_saturates_and_unpack_round_trips():
    """nrc_wire_pack_f16 / nrc_wire_unpack_f16 (the optional 16-bit wire of the sharded step): round-to-nearest fp16 of the clamped value, the clamped count on the
    device, NaN passed on, odd lengths and a misaligned start."""
    from nerficg_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device=DEV).manual_seed(2)
    base = torch.randn(100_004, device=DEV, generator=gen) * 10.0 ** torch.randint(-6, 6, (100_004,), device=DEV, generator=gen).float()
    for off, n in ((0, 100_004), (1, 100_003), (0, 5)):
        src = base[off:off + n]
        src_bad = src.clone()
        if n > 100:
            src_bad[7], src_bad[11], src_bad[13] = 1e9, -3e7, float('nan')
        dst = torch.zeros(n, dtype=torch.float16, device=DEV)
        sat = torch.zeros(1, dtype=torch.int64, device=DEV)
        _lib.check(lib.nrc_wire_pack_f16(_lib.ptr(src_bad), _lib.ptr(dst), n, _lib.ptr(sat), _lib.stream_of(dst)), 'wire_pack_f16')
        ref = src_bad.clamp(-65504.0, 65504.0).half()
        assert torch.equal(dst[~torch.isnan(src_bad)], ref[~torch.isnan(src_bad)]) and bool(torch.isnan(dst[torch.isnan(src_bad)]).all())
        assert int(sat) == int((src_bad.abs() > 65504.0).sum()) and bool(torch.isfinite(dst[~torch.isnan(src_bad)]).all())
        back = torch.zeros(n, device=DEV)
        _lib.check(lib.nrc_wire_unpack_f16(_lib.ptr(dst), _lib.ptr(back), n, _lib.stream_of(back)), 'wire_unpack_f16')
        assert torch.equal(back[~torch.isnan(src_bad)], dst.float()[~torch.isnan(src_bad)])


def test_adam_launch_leaves_out_a_nonfinite_gradient_element_and_says_so():
    """The fused step trusts its producers' overflow flags and runs no pass over the summed gradients; should an inf / NaN reach the Adam launch all the same (a sum of
    finite values overflowing behind the flags), that element keeps its parameter and moments, every other element is updated, and state4[3] is raised."""
    from nerficg_amd import _lib
    lib, p = _lib.load(), _lib.ptr
    n = 10_001
    gen = torch.Generator(device=DEV).manual_seed(6)
    P0, G = torch.randn(n, device=DEV, generator=gen), torch.randn(n, device=DEV, generator=gen) * 128
    G[17], G[4096], G[n - 1] = float('inf'), float('nan'), float('-inf')
    Pm, M, V, H = P0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), P0.half()
    step, bc, scale, tracker, state4 = (torch.zeros(1, dtype=torch.int32, device=DEV), torch.zeros(2, device=DEV), torch.full((1,), 128.0, device=DEV),
                                        torch.zeros(1, dtype=torch.int32, device=DEV), torch.zeros(4, device=DEV))
    flag = torch.zeros(1, device=DEV)       # the producers saw nothing
    s = _lib.stream_of(Pm)
    _lib.check(lib.nrc_amp_settle(p(flag), 1.0, 0.9, 0.99, p(step), p(bc), p(scale), p(tracker), 2.0, 0.5, 1000, p(state4), None, s), 'amp_settle')
    _lib.check(lib.nrc_amp_adam_slices(p(Pm), p(G), p(M), p(V), p(H), n, 0.0, 0, None, None, None, None, None, 0, 0.0, 0, 1e-2, None, 0.9, 0.99, 1e-15, 0.0, 0,
                                       p(bc), p(state4), s), 'amp_adam_slices')
    bad = ~torch.isfinite(G)
    assert float(state4[3]) == 1.0 and int(step) == 1
    assert torch.equal(Pm[bad], P0[bad]) and bool((M[bad] == 0).all()) and bool((V[bad] == 0).all()) and torch.equal(H[bad], P0.half()[bad])
    assert bool(torch.isfinite(Pm).all()) and bool((Pm[~bad] != P0[~bad]).all()) and torch.equal(H[~bad], Pm.half()[~bad])
