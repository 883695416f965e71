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
    assert opt._l2_slices[id(p)] == (3, 0.5)
    opt.set_l2_slice(p, 0, 0.0)
    assert id(p) not in opt._l2_slices
    with pytest.raises(RuntimeError, match='master_weights'):
        FusedAdam([p], master_weights=True)
    with pytest.raises(RuntimeError, match='AMSGrad'):
        FusedAdam([p], amsgrad=True)


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
