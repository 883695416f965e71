"""GPU: the 3DGS model life cycle through the MI355X-native pieces only -- point-cloud initialisation (HIP kNN), optimisation steps
(rasterizer fwd/bwd, SSIM loss, fused Adam), densification statistics + densify_and_prune (HIP plan/gather), activation baking (HIP Morton
codes + gather), reference-format checkpoint and PLY, and inference from the baked covariances (Renderer.py:129-139)."""
import numpy as np
import pytest
import torch

from tests import scenes

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def test_train_densify_bake_save_load_render(tmp_path):
    from nerficg_amd import formats
    from nerficg_amd.gaussian_splatting import Gaussians, PerspectiveCamera, render_image_inference, render_image_training, training_loss
    rng = np.random.default_rng(0)
    W, H = 160, 120
    cam = PerspectiveCamera(W, H, 1.2 * W, 1.2 * W, background_color=torch.tensor([0.1, 0.2, 0.3], device=DEV))
    poses = [scenes.orbit_pose(0.4 + 1.3 * k, 0.3, 3.0) for k in range(4)]
    # target: a denser, coloured cloud of the same shape rendered by the same rasterizer
    truth = scenes.gs_random_scene(3000, seed=1, extent=1.0, log_scale_mean=np.log(0.05))
    t = {k: torch.from_numpy(v).to(DEV) if isinstance(v, np.ndarray) else v for k, v in truth.items()}
    gt = Gaussians(t['means3D'], torch.log(t['scales']), t['rotations'], torch.logit(t['opacities'].clamp(1e-4, 1 - 1e-4))[:, None],
                   t['shs'][:, :1].contiguous(), t['shs'][:, 1:].contiguous())
    targets = [render_image_inference(gt, cam, p, to_chw=True)['rgb'] for p in poses]

    pts = torch.from_numpy(truth['means3D'][::4] + rng.normal(size=(750, 3)).astype(np.float32) * 0.02).to(DEV)
    g = Gaussians.from_point_cloud(pts, colors=torch.full((750, 3), 0.5, device=DEV))
    assert g.active_sh_degree == 0 and g._scales.shape == (750, 3) and bool(torch.isfinite(g._scales).all())
    g.training_setup(training_cameras_extent=3.0)
    first = last = None
    for it in range(200):
        g.update_learning_rate(it)
        out = render_image_training(g, cam, poses[it % 4])
        loss = training_loss(out['rgb'], targets[it % 4])
        loss.backward()
        g.add_densification_stats(out['viewspace_points'], out['visibility_mask'])
        g.optimizer.step()
        g.optimizer.zero_grad()
        first = loss.item() if first is None else first
        last = loss.item()
        if it == 80:
            assert int(g.n_observations.max()) > 0 and float(g.densification_gradient_accum.max()) > 0
            n_before = g.get_positions.shape[0]
            info = g.densify_and_prune(grad_threshold=1e-5, min_opacity=0.005, prune_large_gaussians=False)
            assert info['n_out'] == g.get_positions.shape[0] == g.n_observations.shape[0] and info['n_out'] > n_before  # the sparse init is under-reconstructed
            assert g.optimizer.state[g._positions]['exp_avg'].shape == g._positions.shape
            g.increase_used_sh_degree()
    assert np.isfinite(last) and last < 0.9 * first, (first, last)

    unbaked = render_image_inference(g, cam, poses[1], to_chw=True)['rgb']
    g.bake_activations()
    assert g.baked and g.get_baked_covariances.shape == (g.get_positions.shape[0], 6)
    formats.gaussians_to_checkpoint(g, tmp_path / 'final.pt', model_name='lifecycle', num_iterations_trained=200)
    h, meta = formats.gaussians_from_checkpoint(tmp_path / 'final.pt', device=DEV)
    assert meta['num_iterations_trained'] == 200 and h.baked and h.active_sh_degree == 3
    h.active_sh_degree = g.active_sh_degree  # the training above activated one degree only
    baked = render_image_inference(h, cam, poses[1], to_chw=True)['rgb']
    fused = render_image_inference(h, cam, poses[1], to_chw=True, use_baked_covariance=False)['rgb']
    # baked covariances (torch R S S^T R^T) vs the in-kernel computation, and baked (1/255-pruned, Morton-ordered) vs unbaked model
    assert float((baked - fused).abs().max()) < 2e-3 and float((baked - unbaked).abs().max()) < 2e-2
    assert float((baked - unbaked).abs().mean()) < 5e-4
    formats.write_ply(tmp_path / 'final.ply', formats.gaussians_ply_dict(h))
    back = formats.read_ply(tmp_path / 'final.ply')
    np.testing.assert_array_equal(back['vertex']['x'], h.get_positions[:, 0].detach().cpu().numpy())


def test_precomputed_rays_form_a_collection():
    from nerficg_amd.instant_ngp import Camera
    from nerficg_amd.rays import compute_all_rays, rays_of_view
    cam_a, cam_b = Camera(32, 24, 40.0, 40.0), Camera(20, 10, 30.0, 31.0)
    views = [dict(camera=cam_a, c2w=scenes.orbit_pose(0.2, 0.1, 1.3), rgb=torch.rand(3, 24, 32), alpha=torch.rand(1, 24, 32)),
             dict(camera=cam_b, c2w=scenes.orbit_pose(1.2, -0.3, 1.1), rgb=torch.rand(3, 10, 20), alpha=torch.rand(1, 10, 20))]
    col = compute_all_rays(views, as_ray_collection=True)
    assert len(col) == 2 and len(col.all_rays) == 32 * 24 + 20 * 10 and col.all_rays.origin.is_cuda
    second = rays_of_view(cam_b, views[1]['c2w'], views[1]['rgb'], views[1]['alpha'])
    assert torch.equal(col[1].direction, second.direction) and torch.equal(col[1].rgb, second.rgb)
    host = compute_all_rays(views, store_on_cpu=True)
    assert not host.origin.is_cuda and torch.equal(host.origin, col.all_rays.origin.cpu())


def test_rest_step_inside_the_backward_pass_equals_the_optimizers_own_launch():
    """render_image_training(fuse_rest_step=True): the Adam step of the 45 higher SH coefficients runs inside the preprocessing backward (C ABI
    nrc_gs_backward_rest_step) -- three steps leave the same parameters, moments and step counters as the optimizer's own launch, to within the run-to-run spread of
    the unfused loop (the per-Gaussian sums across tiles are float atomics); the gradient of that tensor is never materialised."""
    import numpy as np
    from nerficg_amd.gaussian_splatting import Gaussians, PerspectiveCamera, render_image_training, training_loss
    from tests import scenes
    from tests.noise import assert_within_run_to_run_noise
    dev = torch.device('cuda', 0)
    sc = scenes.gs_random_scene(30_000, seed=5, extent=1.0, log_scale_mean=np.log(0.03))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    cam = PerspectiveCamera(320, 200, 380.0, 380.0, background_color=torch.zeros(3, device=dev))
    target = torch.rand(3, 200, 320, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    poses = [torch.from_numpy(np.asarray(scenes.orbit_pose(0.5 + 0.9 * i, 0.3, 3.0), dtype=np.float32)).to(dev) for i in range(3)]

    def run(fuse):
        g = Gaussians(T(sc['means3D']), torch.log(T(sc['scales'])), T(sc['rotations']), torch.logit(T(sc['opacities']).clamp(1e-4, 1 - 1e-4))[:, None].contiguous(),
                      T(sc['shs'][:, :1]), T(sc['shs'][:, 1:]))
        g.training_setup(training_cameras_extent=3.0)
        g.fuse_rest_step = fuse
        for i in range(3):
            out = render_image_training(g, cam, poses[i])
            training_loss(out['rgb'], target).backward()
            assert (g._features_rest.grad is None) == fuse and g._features_dc.grad is not None
            g.optimizer.step(); g.optimizer.zero_grad()
        params = [grp['params'][0].detach().clone() for grp in g.optimizer.param_groups]
        moments = [g.optimizer.state[grp['params'][0]][k].clone() for grp in g.optimizer.param_groups for k in ('exp_avg', 'exp_avg_sq')]
        return params + moments, [grp['step'] for grp in g.optimizer.param_groups]

    ref, ref2, got = run(False), run(False), run(True)
    assert got[1] == ref[1] == [3] * 6
    assert_within_run_to_run_noise(got[0], ref[0], ref2[0], atol=1e-6, rtol=1e-3, what='three steps, rest step inside the backward pass')
    moved = float((got[0][2] - T(sc['shs'][:, 1:])).abs().max())
    assert moved > 1e-5     # the tensor really was stepped


def test_rest_step_schedule_keeps_the_reference_order_where_an_update_is_dropped():
    """Gaussians.fuse_rest_schedule (rest_step_schedule): the trainer's own update_learning_rate(iteration + 1) call switches the in-backward f_rest step per iteration.
    The loop below is the reference's order -- backward, [on densify iterations: the parameters lose their gradients, here by zero_grad(), as prune_points' fresh
    nn.Parameters do], optimizer.step() -- with start 1, end 6, interval 2: iterations 3 and 5 drop their update.  With the schedule the fused loop lands where the
    plain loop does (run-to-run spread of the float atomics); without it the f_rest tensor has taken two steps the reference drops."""
    import numpy as np
    from nerficg_amd.gaussian_splatting import Gaussians, PerspectiveCamera, render_image_training, training_loss, rest_step_schedule
    from tests.noise import assert_within_run_to_run_noise
    sc = scenes.gs_random_scene(20_000, seed=6, extent=1.0, log_scale_mean=np.log(0.03))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)   # noqa: E731
    cam = PerspectiveCamera(256, 160, 300.0, 300.0, background_color=torch.zeros(3, device=DEV))
    target = torch.rand(3, 160, 256, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    poses = [torch.from_numpy(np.asarray(scenes.orbit_pose(0.4 + 0.8 * i, 0.3, 3.0), dtype=np.float32)).to(DEV) for i in range(7)]
    clean = rest_step_schedule(1, 6, 2)
    assert [it for it in range(7) if not clean(it)] == [3, 5]

    def run(mode):
        g = Gaussians(T(sc['means3D']), torch.log(T(sc['scales'])), T(sc['rotations']), torch.logit(T(sc['opacities']).clamp(1e-4, 1 - 1e-4))[:, None].contiguous(),
                      T(sc['shs'][:, :1]), T(sc['shs'][:, 1:]))
        g.training_setup(training_cameras_extent=3.0)
        if mode == 'scheduled':
            g.fuse_rest_schedule = clean
        elif mode == 'always':
            g.fuse_rest_step = True
        fused_in = []
        for it in range(7):
            g.update_learning_rate(it + 1)                      # Trainer.py:84
            out = render_image_training(g, cam, poses[it])
            training_loss(out['rgb'], target).backward()
            fused_in.append(g._features_rest.grad is None)
            if not clean(it):
                g.optimizer.zero_grad()                         # what densify_and_prune leaves for optimizer.step(): parameters without gradients
            g.optimizer.step(); g.optimizer.zero_grad()
        return [grp['params'][0].detach().clone() for grp in g.optimizer.param_groups], fused_in

    plain, plain2, sched, always = run('plain'), run('plain'), run('scheduled'), run('always')
    assert plain[1] == [False] * 7 and always[1] == [True] * 7 and sched[1] == [True, True, True, False, True, False, True]
    assert_within_run_to_run_noise(sched[0], plain[0], plain2[0], atol=1e-6, rtol=1e-3, what='seven iterations, scheduled rest step')
    rest = 2     # param group order of training_setup: positions, f_dc, f_rest, ...
    spread = float((plain[0][rest] - plain2[0][rest]).abs().max())
    assert float((always[0][rest] - plain[0][rest]).abs().max()) > 100 * max(spread, 1e-7)      # two extra steps of ~lr each: not noise
