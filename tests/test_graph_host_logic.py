"""CPU: host-side logic around the recorded iterations (nerficg_amd.graphs) -- argument checks and the capacity context; no kernel runs."""
import pytest
import torch


def test_graphed_iteration_wants_device_buffers_and_an_eager_call():
    from nerficg_amd.graphs import GraphedIteration
    with pytest.raises(RuntimeError, match='must live on the GPU'):
        GraphedIteration(lambda x: {'y': x}, {'x': torch.zeros(3)})
    with pytest.raises(ValueError, match='eager call'):
        GraphedIteration(lambda x: {'y': x}, {'x': torch.zeros(3)}, eager_calls=0)


def test_fixed_capacity_context_nests_and_restores():
    from nerficg_amd import diff_gaussian_rasterization as dgr
    assert dgr._FIXED_CAPACITY is None
    with dgr.fixed_capacity(1000, 50):
        assert dgr._FIXED_CAPACITY == (1000, 50)
        with dgr.fixed_capacity(7):
            assert dgr._FIXED_CAPACITY == (7, 0)
        assert dgr._FIXED_CAPACITY == (1000, 50)
    assert dgr._FIXED_CAPACITY is None
    with pytest.raises(ValueError):
        with dgr.fixed_capacity(0):
            pass
    with pytest.raises(RuntimeError):  # an exception inside the block still restores the previous setting
        with dgr.fixed_capacity(5):
            raise RuntimeError('boom')
    assert dgr._FIXED_CAPACITY is None


def test_fused_adam_flags():
    from nerficg_amd.apex_optimizers import FusedAdam
    p = torch.nn.Parameter(torch.zeros(4))
    opt = FusedAdam([p], lr=1e-3, capturable=True)
    assert opt.capturable and opt._step_supports_amp_scaling
    opt.set_l2_slice(p, 3, 0.5)
    assert opt._l2_slice_of(p) == (3, 0.5)
    opt.set_l2_slice(p, 0, 0.0)
    assert id(p) not in opt._l2_slices and opt._l2_slice_of(p) == (0, 0.0)
    # the slice belongs to the parameter OBJECT: another tensor never inherits it, a dropped iteration object removes it (round-2 advisor finding)
    opt.set_l2_slice(p, 3, 0.5)
    q = torch.nn.Parameter(torch.zeros(4))
    opt._l2_slices[id(q)] = opt._l2_slices[id(p)]          # what a reused id would look like
    assert opt._l2_slice_of(q) == (0, 0.0) and id(q) not in opt._l2_slices
    opt.clear_l2_slices()
    assert opt._l2_slice_of(p) == (0, 0.0)
    # a slice is removed by the token of whoever installed it: an owner that goes away AFTER its successor re-installed the slice on the same
    # optimizer (rebuild of a recorded iteration with a new n_rays) must not take the successor's entry with it (round-3 advisor finding)
    tok_a = opt.set_l2_slice(p, 3, 0.5)
    tok_b = opt.set_l2_slice(p, 3, 0.25)
    assert not opt.remove_l2_slice(p, tok_a) and opt._l2_slice_of(p) == (3, 0.25)
    assert opt.remove_l2_slice(p, tok_b) and opt._l2_slice_of(p) == (0, 0.0)
    assert not opt.remove_l2_slice(p, tok_b)
    with pytest.raises(RuntimeError, match='master_weights'):
        FusedAdam([p], master_weights=True)
    with pytest.raises(RuntimeError, match='AMSGrad'):
        FusedAdam([p], amsgrad=True)


def test_fused_adam_group_lookup_and_saved_skips():
    """effective_step() finds a group by identity (list.index would compare the groups' 'params' lists with ==: tensor == tensor, ambiguous
    truth value for the reference's one-tensor-per-group layout), and the overflow-skipped steps travel in state_dict()."""
    from nerficg_amd.apex_optimizers import FusedAdam
    a, b = torch.nn.Parameter(torch.zeros(4)), torch.nn.Parameter(torch.zeros(6))
    opt = FusedAdam([{'params': [a], 'name': 'a'}, {'params': [b], 'name': 'b'}], lr=1e-3)
    opt.param_groups[1]['step'] = 5
    assert opt.effective_step(opt.param_groups[1]) == 5 and opt.effective_step(opt.param_groups[0]) == 0
    opt._amp[1] = (torch.full((1,), 2, dtype=torch.int32), torch.ones(2))
    assert opt.effective_step(opt.param_groups[1]) == 3
    sd = opt.state_dict()
    assert [g['skipped_steps'] for g in sd['param_groups']] == [0, 2]
    opt2 = FusedAdam([{'params': [a], 'name': 'a'}, {'params': [b], 'name': 'b'}], lr=1e-3)
    opt2.load_state_dict(sd)
    assert opt2.effective_step(opt2.param_groups[1]) == 3 and 'skipped_steps' not in opt2.param_groups[1]


def test_fused_adam_load_state_dict_updates_device_scalars_in_place():
    """A recorded iteration holds raw pointers to the optimizer's device scalars: a load must write into them, not replace them, and a
    non-capturable checkpoint (host step incl. skipped steps) becomes the effective step of the capturable mode (round-3 advisor finding)."""
    from nerficg_amd.apex_optimizers import FusedAdam
    a = torch.nn.Parameter(torch.zeros(4))
    src = FusedAdam([a], lr=2e-3)
    src.param_groups[0]['step'] = 9
    src._amp[0] = (torch.full((1,), 2, dtype=torch.int32), torch.ones(2))
    sd = src.state_dict()
    assert sd['param_groups'][0]['step'] == 9 and sd['param_groups'][0]['skipped_steps'] == 2
    cap = FusedAdam([a], lr=1e-3, capturable=True)
    step_dev, skipped_dev, lr_dev = torch.zeros(1, dtype=torch.int32), torch.full((1,), 5, dtype=torch.int32), torch.full((1,), 1e-3)
    cap.param_groups[0]['step'] = step_dev
    cap._amp[0] = (skipped_dev, torch.ones(2))
    cap._lr_dev[0] = [lr_dev, 1e-3]
    cap.load_state_dict(sd)
    assert cap.param_groups[0]['step'] is step_dev and int(step_dev) == 7          # 9 host steps, 2 of them skipped
    assert cap._amp[0][0] is skipped_dev and int(skipped_dev) == 0
    assert cap._lr_dev[0][0] is lr_dev and abs(float(lr_dev) - 2e-3) < 1e-9 and cap._lr_dev[0][1] == 2e-3
    sd2 = cap.state_dict()
    assert sd2['param_groups'][0]['step'] == 7 and sd2['param_groups'][0]['skipped_steps'] == 0


def test_graphed_iteration_close_runs_its_hooks():
    from nerficg_amd.graphs import GraphedIteration
    if not torch.cuda.is_available():
        it = GraphedIteration.__new__(GraphedIteration)
        it.graph = it.outputs = None
        seen = []
        it.on_close = (lambda: seen.append(1),)
        it.close(); it.close()
        assert seen == [1]


def test_recorded_iterations_need_a_capturable_optimizer():
    from nerficg_amd.apex_optimizers import FusedAdam
    from nerficg_amd.graphs import gaussian_splatting_step, instant_ngp_iteration

    class Stub:
        pass

    p = torch.nn.Parameter(torch.zeros(4))
    opt = FusedAdam([p], lr=1e-3)
    model = Stub(); model.center = torch.zeros(3)
    with pytest.raises(RuntimeError, match='capturable=True'):
        instant_ngp_iteration(model, None, opt, None, None, n_rays=8, sample_capacity=64)
    g = Stub(); g.optimizer = opt
    with pytest.raises(RuntimeError, match='capturable=True'):
        gaussian_splatting_step(g, None, instance_capacity=10)


def test_bench_refuses_a_world_size_that_contradicts_gpus(tmp_path):
    """`bench.py --gpus N` must never silently benchmark another number of ranks (round-2 review: it printed n_gpus 1 for --gpus 8)."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, str(root / 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and '--gpus 2 but WORLD_SIZE=1' in (r.stderr + r.stdout)


def test_bench_starts_its_own_ranks_when_no_launcher_is_present(monkeypatch):
    import subprocess
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import bench
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen['cmd'], seen['env'] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3'])
    assert bench.spawn_ranks(4) == 7
    cmd = seen['cmd']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and '--nproc-per-node=4' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-4:] == ['--gpus', '4', '--steps', '3']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_fast_grad_scaler_falls_back_to_torch_on_host_tensors():
    """nerficg_amd.amp.GradScaler only replaces the check for contiguous f32 device gradients; anything else takes torch's own path."""
    from nerficg_amd.amp import GradScaler
    p = torch.nn.Parameter(torch.ones(5))
    opt = torch.optim.SGD([p], lr=0.1)
    scaler = GradScaler('cpu', init_scale=4.0, growth_interval=10 ** 6)
    for x in (1.0, float('inf'), 2.0):
        loss = (p * x).sum()
        scaler.scale(loss).backward()
        scaler.step(opt); scaler.update(); opt.zero_grad()
    assert torch.allclose(p.detach(), torch.full((5,), 1.0 - 0.1 * 1.0 - 0.1 * 2.0)) and scaler.get_scale() == 2.0
