"""GPU: the host mailbox (include/nerficg_hip.h: nrc_host_mailbox_alloc; nerficg_amd._lib.HostMailbox) and the paths behind it that never run while it
works -- a ticket that is never posted times out, a retired mailbox sends the rasterizer and the image renderer back to their device-to-host reads,
and both forms give the same lists / pictures."""
import numpy as np
import pytest
import torch

from tests import scenes

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def _gs_frame(n=30_000, seed=11):
    from nerficg_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    sc = scenes.gs_random_scene(n, seed=seed)
    cam = scenes.gs_camera(200, 136, scenes.orbit_pose(0.7, 0.3, 4.0))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    rs = GaussianRasterizationSettings(image_height=136, image_width=200, tanfovx=cam['tanfovx'], tanfovy=cam['tanfovy'], bg=torch.zeros(3, device=DEV),
                                       scale_modifier=1.0, viewmatrix=T(cam['viewmatrix']), projmatrix=T(cam['projmatrix']), sh_degree=3,
                                       campos=T(cam['campos']), prefiltered=False, debug=False)
    t = {k: T(v) for k, v in sc.items() if k != 'sh_degree'}
    rast = GaussianRasterizer(rs)

    def frame():
        m = t['means3D'].detach().requires_grad_(True)
        color, radii = rast(means3D=m, means2D=torch.zeros_like(m), opacities=t['opacities'][:, None].contiguous(), shs=t['shs'], scales=t['scales'],
                            rotations=t['rotations'])
        fn = color.grad_fn
        saved = dict(zip(('point_list', 'ranges'), (fn.saved_tensors[12], fn.saved_tensors[13])))
        return color.detach().clone(), fn.num_rendered, saved['point_list'][:fn.num_rendered].clone(), saved['ranges'].clone()
    return frame


def test_a_ticket_that_is_never_posted_times_out_and_a_posted_one_is_seen():
    from nerficg_amd import _lib
    box = _lib.HostMailbox(DEV)
    old = _lib.HostMailbox.TIMEOUT_S
    _lib.HostMailbox.TIMEOUT_S = 0.05
    try:
        assert box.wait(box.next_ticket()) is None
    finally:
        _lib.HostMailbox.TIMEOUT_S = old
    # a spin that gives up too early (as behind seconds of queued work) is not a broken mailbox: the stream is synchronised, the counts are there
    frame0 = _gs_frame(seed=3)
    frame0()
    ref = frame0()
    spin = _lib.HostMailbox.wait
    _lib.HostMailbox.wait = lambda self, ticket: None          # every spin "times out"
    try:
        a = frame0()
    finally:
        _lib.HostMailbox.wait = spin
    assert _lib.HostMailbox.for_device(DEV) is not None and a[1] == ref[1] and torch.equal(a[0], ref[0])
    # a kernel that posts: the rasterizer's counting pass (through the wrapper, which uses the per-device mailbox)
    frame = _gs_frame()
    frame(); frame()                      # the second frame is sized speculatively and takes its counts from the mailbox
    mb = _lib.HostMailbox.for_device(DEV)
    assert mb is not None and mb._ticket >= 1 and int(mb._seen[2]) == mb._ticket and int(mb._seen[0]) > 0


def test_retired_mailbox_falls_back_to_device_reads_with_the_same_results():
    from nerficg_amd import _lib
    import nerficg_amd.diff_gaussian_rasterization as dgr
    from nerficg_amd.instant_ngp import InstantNGPRenderer
    from tests.test_gpu_render_parity import make_camera, make_model
    frame = _gs_frame(seed=12)
    frame()
    with_box = frame()
    model = make_model()
    cam = make_camera(96, 72)
    pose = scenes.orbit_pose(0.5, 0.2, scenes.LEGO_RADIUS)
    r = InstantNGPRenderer(model)
    r.render_image_fused(cam, pose)
    img_box = {k: v.clone() for k, v in r.render_image_fused(cam, pose, early_termination=False).items()}
    saved = dict(_lib.HostMailbox._per_device)
    try:
        _lib.HostMailbox.retire(DEV)
        assert _lib.HostMailbox.for_device(DEV) is None
        without = frame()                 # side-stream copy of the counts
        assert with_box[1] == without[1] and torch.equal(with_box[0], without[0]) and torch.equal(with_box[2], without[2]) and torch.equal(with_box[3], without[3])
        img = r.render_image_fused(cam, pose, early_termination=False)     # counter.tolist()
        for k in ('rgb', 'alpha', 'depth'):
            assert torch.equal(img[k], img_box[k]), k
        dgr.COUNT_MAILBOX = False
        assert torch.equal(frame()[0], with_box[0])
    finally:
        dgr.COUNT_MAILBOX = True
        _lib.HostMailbox._per_device.clear()
        _lib.HostMailbox._per_device.update(saved)
    again = frame()
    assert torch.equal(again[0], with_box[0]) and again[1] == with_box[1]
