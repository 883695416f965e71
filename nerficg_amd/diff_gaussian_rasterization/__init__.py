"""nerficg_amd.diff_gaussian_rasterization -- drop-in for the `diff_gaussian_rasterization` package the reference imports
through src/Thirdparty/DiffGaussianRasterization.py:17 (pinned upstream commit 59f5f77e, :9) and drives from
src/Methods/GaussianSplatting/Renderer.py:60-81,94-153,163-183:

    GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix, projmatrix,
                                  sh_degree, campos, prefiltered, debug)
    GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                                        rotations=None, cov3D_precomp=None) -> (color (3,H,W) f32, radii (P,) i32)
    GaussianRasterizer.markVisible(positions) -> bool mask

Gradients flow to every provided input and to `means2D` (the screen-space gradient that densification reads,
src/Methods/GaussianSplatting/Model.py:258).  Backed by libnerficg_hip.so (nrc_gs_*); no CPU fallback.
"""
from __future__ import annotations

import contextlib
import warnings
from typing import NamedTuple

import torch

from .. import _lib

__all__ = ['GaussianRasterizationSettings', 'GaussianRasterizer', 'rasterize_gaussians', 'fixed_capacity', 'last_counts']


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


_CAMERA_BLOCKS: dict = {}
_ADOPTED_BLOCK: list = [None]   # the latest camera block written by nrc_gs_camera_block (gaussian_splatting.make_raster_settings)


def adopt_camera_block(block: torch.Tensor) -> None:
    """Registers a float[38] device block whose slices a GaussianRasterizationSettings is about to carry (viewmatrix = block[:16], projmatrix = block[16:32],
    campos = block[32:35], bg = block[35:38]): _camera_block() then uses it as it is."""
    _ADOPTED_BLOCK[0] = block


def _camera_block(rs, dev) -> torch.Tensor:
    """viewmatrix | projmatrix | campos | bg of the settings as ONE device float[38] (the `camera_dev` block of the C ABI): the camera never
    crosses to the host.  Built with one concatenation when the settings carry tensors this process has not seen yet (same storage, same
    version counter -> same values), so a renderer that draws several passes from one camera pays it once.  During a stream capture it is
    always rebuilt, so that the recorded graph re-reads the (then static) camera tensors on every replay."""
    tensors = (rs.viewmatrix, rs.projmatrix, rs.campos, rs.bg)
    blk = _ADOPTED_BLOCK[0]
    if blk is not None and blk.device == torch.device(dev) and all(t.is_cuda and t.dtype == torch.float32 for t in tensors):
        base = blk.data_ptr()
        if (rs.viewmatrix.data_ptr() == base and rs.projmatrix.data_ptr() == base + 64 and rs.campos.data_ptr() == base + 128 and rs.bg.data_ptr() == base + 140
                and rs.viewmatrix.is_contiguous() and rs.projmatrix.is_contiguous()):
            return blk     # (also inside a stream capture: the launch that fills the block is part of the recording)
    capturing = torch.cuda.is_current_stream_capturing()
    key = tuple((t.data_ptr(), t._version, t.device) for t in tensors)
    hit = None if capturing else _CAMERA_BLOCKS.get(key)
    if hit is None:
        block = torch.cat([t.detach().reshape(-1).to(device=dev, dtype=torch.float32) for t in tensors])
        if block.numel() != 38:
            raise RuntimeError(f'raster settings: viewmatrix (4,4), projmatrix (4,4), campos (3), bg (3) expected, got {block.numel()} values')
        if capturing:
            return block
        if len(_CAMERA_BLOCKS) >= 256:
            _CAMERA_BLOCKS.pop(next(iter(_CAMERA_BLOCKS)))
        hit = _CAMERA_BLOCKS[key] = (block, tensors)  # the tensors are kept alive: their addresses are the key
    return hit[0]


_FIXED_CAPACITY: tuple[int, int] | None = None
_LAST_COUNTS: torch.Tensor | None = None


@contextlib.contextmanager
def fixed_capacity(instances: int, spans: int = 0):
    """Inside the block the rasterizer sizes nothing from device counts: the per-tile lists hold `instances` entries, the binning workspace
    `spans` row-span records (0: the default 4 P + 65536), and the forward reads nothing back -- which is what a HIP-graph capture needs
    (nerficg_amd.graphs.gaussian_splatting_step).  Instances beyond the capacity are dropped; `last_counts()` tells."""
    global _FIXED_CAPACITY
    if instances < 1 or spans < 0:
        raise ValueError('fixed_capacity: instances >= 1, spans >= 0')
    previous, _FIXED_CAPACITY = _FIXED_CAPACITY, (int(instances), int(spans))
    try:
        yield
    finally:
        _FIXED_CAPACITY = previous


def last_counts() -> torch.Tensor | None:
    """DEVICE int64[2] of the most recent forward: (tile, Gaussian) instances and (tile row, Gaussian) spans the frame needed."""
    return _LAST_COUNTS


_GRAD_RECORDS: dict = {}   # (device, stream) -> (rows, (rows, 16) f32 accumulator records known to be all zero): ONE buffer per device and stream (see backward)
_SPAN_CAPACITY: dict = {}  # (device, P, W, H) -> row-span capacity of the binning workspace once the default proved too small
# Speculative sizing of the tile lists (round 4).  The forward used to STOP at the instance count: D2H copy, host wait, allocation, and only then
# the list scatter and the blend -- 33-37 us of idle GPU per frame at 1 M Gaussians (rocprofv3 kernel trace).  Now the lists are sized from the
# counts of the recent frames of the same (device, P, W, H) (largest of the last 16 x 1.3 + 64 K entries), everything is enqueued at once with the
# capacity-checked scatter, and the count travels to pinned host memory on a side stream while the blend kernel runs; the host looks at it
# before returning and repeats the frame with exact sizes in the rare case that it did not fit.  Same lists, same image.
_INSTANCE_HISTORY: dict = {}   # (device, P, W, H) -> recent instance counts
_READBACK: dict = {}           # device -> (side stream, pinned int64[2])
SPECULATIVE_SIZING = True
# How the count reaches the host (round 4, later).  The side-stream copy needs an event on the caller's stream between the counting kernels and the list
# scatter -- 6-7 us of idle GPU in the kernel trace -- plus ~20 us of host calls, and the copy itself lands ~15 us after the count exists.  With a
# MAILBOX (include/nerficg_hip.h, nrc_host_mailbox_alloc: pinned, device-mapped, coherent host memory) the last workgroup of the counting kernel
# stores the two counts and this call's ticket straight into host memory and the host polls the ticket: no event, no copy, no second stream.
# A spin that times out (_lib.HostMailbox.TIMEOUT_S) synchronises the stream and looks again; a mailbox still silent then (never observed) is not used again.
COUNT_MAILBOX = True


def _readback(dev):
    hit = _READBACK.get(dev)
    if hit is None:
        hit = _READBACK[dev] = (torch.cuda.Stream(device=dev), torch.empty(2, dtype=torch.int64).pin_memory())
    return hit


def _opt(t):
    """Absent optional inputs arrive as torch.Tensor([]) (1-D, empty), like in the upstream package; a provided tensor of an
    empty scene has more dimensions."""
    return None if t is None or (t.dim() <= 1 and t.numel() == 0) else t


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, sh_rest=None, raw=False, rest_step=None):
        rs = raster_settings
        lib = _lib.load()
        dev = means3D.device
        f32 = torch.float32
        prep = lambda t: None if _opt(t) is None else t.detach().to(f32).contiguous()
        means3D_c, sh_c, col_c, op_c = prep(means3D), prep(sh), prep(colors_precomp), prep(opacities)
        sc_c, rot_c, cov_c = prep(scales), prep(rotations), prep(cov3Ds_precomp)
        rest_c = prep(sh_rest)
        raw = bool(raw)
        if not means3D.is_cuda:
            raise RuntimeError('means3D must be a CUDA tensor')
        P = means3D_c.shape[0]
        W, H = int(rs.image_width), int(rs.image_height)
        D = int(rs.sh_degree)
        M = 0 if sh_c is None else int(sh_c.shape[1]) + (0 if rest_c is None else int(rest_c.shape[1]))
        gx, gy = (W + 15) // 16, (H + 15) // 16
        nt = gx * gy
        cam_block = _camera_block(rs, dev)
        i32, u8 = torch.int32, torch.uint8
        n1 = max(P, 1)
        radii = torch.empty(n1, dtype=i32, device=dev)
        depths = torch.empty(n1, dtype=f32, device=dev)
        points_xy = torch.empty(n1, 2, dtype=f32, device=dev)
        conic_opacity = torch.empty(n1, 4, dtype=f32, device=dev)
        rgb = torch.empty(n1, 3, dtype=f32, device=dev)
        clamped = torch.empty(n1, dtype=u8, device=dev)
        cov3D = torch.empty(n1, 6, dtype=f32, device=dev)
        tiles_touched = torch.empty(n1, dtype=i32, device=dev)
        splat = torch.empty(n1, 16, dtype=f32, device=dev)
        tile_counts = torch.empty(nt, dtype=i32, device=dev)
        tile_fill = torch.empty(nt, dtype=i32, device=dev)
        ranges = torch.empty(nt, 2, dtype=i32, device=dev)
        num_rendered = torch.empty(2, dtype=torch.int64, device=dev)
        st = _lib.stream_of(radii)
        fixed = _FIXED_CAPACITY
        if fixed is None and torch.cuda.is_current_stream_capturing():
            raise RuntimeError('rasterizer forward: sizing the tile lists reads num_rendered on the host, which a stream capture cannot do -- '
                               'wrap the call in diff_gaussian_rasterization.fixed_capacity(instances, spans)')
        # binning workspace: kept per (device, P, W, H) and regrown when a frame needs more row-span records than it holds (the count comes
        # back with the instance count in the one host read of the forward)
        ws_key = (dev, P, W, H)
        span_cap = _SPAN_CAPACITY.get(ws_key, 0) if fixed is None else fixed[1]
        inst_cap = 0 if fixed is None else fixed[0]
        history = _INSTANCE_HISTORY.get(ws_key) if fixed is None else None
        speculative = bool(SPECULATIVE_SIZING and history and P > 0 and gx <= 256 and gy <= 256)   # the span binning path (fixed capacities exist there only)
        if speculative:
            inst_cap = min(int(max(history) * 1.3) + 65536, 0xfffffff0)
        color = torch.empty(3, H, W, dtype=f32, device=dev)
        n_contrib = torch.empty(H * W, dtype=i32, device=dev)
        final_T = torch.empty(H * W, dtype=f32, device=dev)
        global _LAST_COUNTS

        def preprocess(cap_spans, cap_inst, box=None, ticket=0):
            hist_bytes = int(lib.nrc_gs_bin_hist_bytes(P, W, H, cap_spans))
            hist = torch.empty(hist_bytes // 4, dtype=i32, device=dev) if hist_bytes > 0 else None
            _lib.check(lib.nrc_gs_preprocess(
                P, D, M, W, H, _lib.ptr(means3D_c), _lib.ptr(sh_c), _lib.ptr(rest_c), int(raw), _lib.ptr(col_c), _lib.ptr(op_c), _lib.ptr(sc_c), float(rs.scale_modifier),
                _lib.ptr(rot_c), _lib.ptr(cov_c), None, None, None, _lib.ptr(cam_block), float(rs.tanfovx), float(rs.tanfovy), _lib.ptr(radii), _lib.ptr(depths),
                _lib.ptr(points_xy), _lib.ptr(conic_opacity), _lib.ptr(rgb), _lib.ptr(clamped), _lib.ptr(cov3D), _lib.ptr(tiles_touched),
                _lib.ptr(tile_counts), _lib.ptr(ranges), _lib.ptr(tile_fill), _lib.ptr(hist), cap_spans, cap_inst, _lib.ptr(splat), _lib.ptr(num_rendered),
                box if hist is not None else None, ticket, st), 'gs_preprocess')
            return hist

        def bin_render(hist, cap_spans, cap_inst, n_list):
            keys_ = torch.empty(max(n_list, 1) if hist is None else 1, dtype=torch.int64, device=dev)  # only the per-tile key sort fallback uses them
            plist = torch.empty(max(n_list, 1), dtype=i32, device=dev)
            _lib.check(lib.nrc_gs_bin_render(P, W, H, None, _lib.ptr(cam_block), _lib.ptr(radii), _lib.ptr(depths), _lib.ptr(points_xy), _lib.ptr(conic_opacity),
                                             _lib.ptr(rgb), _lib.ptr(ranges), _lib.ptr(tile_fill), _lib.ptr(hist), cap_spans, cap_inst, _lib.ptr(keys_), _lib.ptr(plist),
                                             _lib.ptr(splat), _lib.ptr(color), _lib.ptr(n_contrib), _lib.ptr(final_T), st), 'gs_bin_render')
            return keys_, plist

        done = False
        if speculative:
            mb = _lib.HostMailbox.for_device(dev) if COUNT_MAILBOX else None
            if mb is not None:
                mb.lock.acquire()
            try:
                ticket = mb.next_ticket() if mb is not None else 0
                bin_hist = preprocess(span_cap, inst_cap, mb.ptr if mb is not None else None, ticket)
                have = span_cap if span_cap > 0 else 4 * max(P, 1) + 65536
                counts = None
                if bin_hist is not None and mb is not None:
                    keys, point_list = bin_render(bin_hist, span_cap, inst_cap, inst_cap)
                    counts = mb.counts(ticket, dev)
            finally:
                if mb is not None:
                    mb.lock.release()
            if bin_hist is not None and mb is not None:
                if counts is None:   # (the mailbox retired itself: silent even behind a stream synchronisation)
                    warnings.warn('diff_gaussian_rasterization: the count mailbox did not answer; using the device counters from now on')
                    counts = num_rendered.tolist()
                n_inst, n_spans = counts
                done = n_inst <= inst_cap and n_spans <= have
                if done:
                    point_list = point_list[:max(n_inst, 1)]
                elif n_spans > have:
                    span_cap = _SPAN_CAPACITY[ws_key] = int(n_spans * 1.25) + 65536
            elif bin_hist is not None:
                side, pinned = _readback(dev)
                main = torch.cuda.current_stream(dev)
                counted = torch.cuda.Event()
                counted.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(counted)
                    pinned.copy_(num_rendered, non_blocking=True)
                    copied = torch.cuda.Event()
                    copied.record(side)
                num_rendered.record_stream(side)
                keys, point_list = bin_render(bin_hist, span_cap, inst_cap, inst_cap)
                copied.synchronize()
                n_inst, n_spans = pinned.tolist()
                done = n_inst <= inst_cap and n_spans <= have
                if done:
                    point_list = point_list[:max(n_inst, 1)]
                elif n_spans > have:
                    span_cap = _SPAN_CAPACITY[ws_key] = int(n_spans * 1.25) + 65536
        if not done:
            inst_cap = 0 if fixed is None else fixed[0]
            while True:
                bin_hist = preprocess(span_cap, inst_cap)
                if fixed is not None:
                    n_inst = inst_cap
                    break
                n_inst, n_spans = num_rendered.tolist()
                have = span_cap if span_cap > 0 else 4 * max(P, 1) + 65536
                if bin_hist is None or n_spans <= have:
                    break
                span_cap = _SPAN_CAPACITY[ws_key] = int(n_spans * 1.25) + 65536
                if len(_SPAN_CAPACITY) > 64:
                    _SPAN_CAPACITY.pop(next(iter(_SPAN_CAPACITY)))
            keys, point_list = bin_render(bin_hist, span_cap, inst_cap, n_inst)
        if fixed is None and P > 0:
            recent = _INSTANCE_HISTORY.setdefault(ws_key, [])
            recent.append(int(n_inst))
            del recent[:-16]
            if len(_INSTANCE_HISTORY) > 64:
                _INSTANCE_HISTORY.pop(next(iter(_INSTANCE_HISTORY)))
        _LAST_COUNTS = num_rendered
        ctx.raster_settings = rs
        ctx.dims = (P, D, M, W, H)
        ctx.num_rendered = n_inst if fixed is None else -1
        ctx.has = (sh_c is not None, col_c is not None, sc_c is not None, cov_c is not None)
        ctx.opacity_shape = tuple(opacities.shape)
        ctx.save_for_backward(means3D_c, sh_c if sh_c is not None else torch.empty(0), col_c if col_c is not None else torch.empty(0),
                              sc_c if sc_c is not None else torch.empty(0), rot_c if rot_c is not None else torch.empty(0),
                              cov_c if cov_c is not None else torch.empty(0), radii, points_xy, conic_opacity, rgb, clamped, cov3D,
                              point_list, ranges, n_contrib, final_T, splat, tile_fill, rest_c if rest_c is not None else torch.empty(0), op_c, cam_block)
        ctx.raw, ctx.has_rest = raw, rest_c is not None
        ctx.rest_step = rest_step
        if rest_step is not None and (rest_c is None or rest_c.data_ptr() != sh_rest.data_ptr() or sc_c is None or col_c is not None or cov_c is not None):
            raise RuntimeError('rest_step: needs shs_rest as a contiguous f32 CUDA tensor (updated in place), scales / rotations, no colors_precomp / cov3D_precomp')
        ctx.debug_state = dict(depths=depths, tiles_touched=tiles_touched, keys=keys)
        radii_out = radii[:P]
        ctx.mark_non_differentiable(radii_out)
        # radii carry no gradient: without this autograd hands the backward a zero-filled (P,) int tensor for them -- a 4 MB fill launch per step
        ctx.set_materialize_grads(False)
        return color, radii_out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out_color, _grad_radii):
        (means3D, sh, col, sc, rot, cov, radii, points_xy, conic_opacity, rgb, clamped, cov3D, point_list, ranges, n_contrib,
         final_T, splat, tile_order, sh_rest, opac, cam_block) = ctx.saved_tensors
        raw, has_rest = ctx.raw, ctx.has_rest
        has_sh, has_col, has_sr, has_cov = ctx.has
        P, D, M, W, H = ctx.dims
        rs = ctx.raster_settings
        lib = _lib.load()
        dev = means3D.device
        f32 = torch.float32
        if grad_out_color is None:   # (grads are not materialised: the image did not take part in the loss)
            grad_out_color = torch.zeros(3, H, W, dtype=f32, device=dev)
        g = grad_out_color.to(f32).contiguous()
        n1 = max(P, 1)
        dmean2D = torch.empty(n1, 3, dtype=f32, device=dev)
        # workspace: one 64-byte accumulator record per Gaussian.  The per-Gaussian backward clears every record it reads, so the buffer of the
        # previous backward of the same size is all zero again and is reused without a clearing launch (one buffer per device, size and stream).
        # A recording (HIP-graph capture) takes a buffer of its own and keeps the clearing launch: the recorded pointer must not be shared.
        capturing = torch.cuda.is_current_stream_capturing()
        rec_key = (dev, torch.cuda.current_stream(dev).cuda_stream)
        kept = None if capturing else _GRAD_RECORDS.pop(rec_key, None)
        # the kept buffer serves the same number of Gaussians only (densification changes P every few hundred steps: the old buffer is dropped then,
        # not parked -- eight parked sizes were ~3 GB at 6 M Gaussians, advisor finding of round 4)
        grad_records = kept[1] if kept is not None and kept[0] == n1 else None
        records_clear = grad_records is not None
        if grad_records is None:
            grad_records = torch.empty(n1, 16, dtype=f32, device=dev)
        dcolor = torch.empty(n1, 3, dtype=f32, device=dev) if has_col else None   # only a caller with colors_precomp asks for it
        dopacity = torch.empty(n1, dtype=f32, device=dev)
        dmean3D = torch.empty(n1, 3, dtype=f32, device=dev)
        dcov3D = torch.empty(n1, 6, dtype=f32, device=dev)
        dsh = torch.empty(n1, 1 if has_rest else max(M, 1), 3, dtype=f32, device=dev) if has_sh else None
        rest_step = ctx.rest_step
        dsh_rest = torch.empty(n1, M - 1, 3, dtype=f32, device=dev) if has_rest and rest_step is None else None
        dscale = torch.empty(n1, 3, dtype=f32, device=dev) if has_sr else None
        drot = torch.empty(n1, 4, dtype=f32, device=dev) if has_sr else None
        if rest_step is not None:
            # the optimizer step of the `rest` SH tensor inside the preprocessing backward (nrc_gs_backward_rest_step): no gradient tensor for it
            param, m_rest, v_rest, lr, beta1, beta2, eps, bc1, bc2 = rest_step.take()
            if param.data_ptr() != sh_rest.data_ptr():
                raise RuntimeError('rest_step: the tensor to update is not the one the forward pass read')
            _lib.check(lib.nrc_gs_backward_rest_step(
                P, D, M, W, H, None, _lib.ptr(means3D), _lib.ptr(sh), _lib.ptr(sh_rest), int(raw), _lib.ptr(opac if raw else None), _lib.ptr(sc), float(rs.scale_modifier),
                _lib.ptr(rot), _lib.ptr(cam_block), float(rs.tanfovx), float(rs.tanfovy), _lib.ptr(radii), _lib.ptr(points_xy), _lib.ptr(conic_opacity), _lib.ptr(rgb),
                _lib.ptr(clamped), _lib.ptr(cov3D), _lib.ptr(point_list), _lib.ptr(ranges), _lib.ptr(splat), _lib.ptr(tile_order), _lib.ptr(n_contrib), _lib.ptr(final_T),
                _lib.ptr(g), _lib.ptr(dmean2D), _lib.ptr(dopacity), _lib.ptr(dmean3D), _lib.ptr(dcov3D), _lib.ptr(dsh), _lib.ptr(dscale), _lib.ptr(drot),
                _lib.ptr(grad_records), int(records_clear), _lib.ptr(m_rest), _lib.ptr(v_rest), float(lr), float(beta1), float(beta2), float(eps), float(bc1), float(bc2),
                _lib.stream_of(g)), 'gs_backward_rest_step')
            torch.autograd.graph.increment_version(param)      # written through a raw pointer
        else:
            _lib.check(lib.nrc_gs_backward(
                P, D, M, W, H, None, _lib.ptr(means3D), _lib.ptr(sh if has_sh else None), _lib.ptr(sh_rest if has_rest else None), int(raw),
                _lib.ptr(opac if raw else None), _lib.ptr(col if has_col else None),
                _lib.ptr(sc if has_sr else None), float(rs.scale_modifier), _lib.ptr(rot if has_sr else None), _lib.ptr(cov if has_cov else None),
                None, None, None, _lib.ptr(cam_block), float(rs.tanfovx), float(rs.tanfovy), _lib.ptr(radii), _lib.ptr(points_xy), _lib.ptr(conic_opacity),
                _lib.ptr(rgb), _lib.ptr(clamped), _lib.ptr(cov3D), _lib.ptr(point_list), _lib.ptr(ranges), _lib.ptr(splat), _lib.ptr(tile_order), _lib.ptr(n_contrib), _lib.ptr(final_T),
                _lib.ptr(g), _lib.ptr(dmean2D), None, _lib.ptr(dopacity), _lib.ptr(dcolor), _lib.ptr(dmean3D), _lib.ptr(dcov3D),
                _lib.ptr(dsh), _lib.ptr(dsh_rest), _lib.ptr(dscale), _lib.ptr(drot), _lib.ptr(grad_records), int(records_clear), _lib.stream_of(g)), 'gs_backward')
        if not capturing and P > 0:   # (P == 0: the library returned before touching the uninitialised buffer -- it is NOT known to be zero)
            _GRAD_RECORDS[rec_key] = (n1, grad_records)   # only after a call that went through: a failed one leaves the entry popped (contents unknown)
        cut = (lambda t: t) if P == n1 else (lambda t: t[:P])     # (whole buffers when nothing is cut: a gradient that is not a view can be adopted as .grad without a copy)
        return (cut(dmean3D), cut(dmean2D), cut(dsh) if has_sh else None, cut(dcolor) if has_col else None,
                cut(dopacity).reshape(ctx.opacity_shape), cut(dscale) if has_sr else None, cut(drot) if has_sr else None,
                cut(dcov3D) if has_cov else None, None, cut(dsh_rest) if dsh_rest is not None else None, None, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, sh_rest=None, raw=False, rest_step=None):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, sh_rest, raw, rest_step)


class GaussianRasterizer(torch.nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings) -> None:
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Frustum test of the upstream rasterizer: view-space z > 0.2."""
        with torch.no_grad():
            vm = self.raster_settings.viewmatrix.to(positions.device, torch.float32)  # (4,4) = w2c.T
            z = positions.float() @ vm[:3, 2] + vm[3, 2]
            return z > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, shs_rest=None,
                raw_parameters=False, rest_step=None):
        """The reference's call (Renderer.py:75-81) plus two keyword extensions for callers that own the model tensors: `shs_rest` -- `shs` is then
        (`rest_step`, round 6: an object whose take() returns (parameter, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, bc1, bc2) -- the backward pass then applies
        the optimizer's Adam step to shs_rest itself and returns no gradient for it: nerficg_amd.gaussian_splatting.RestStep)
        the DC part (P,1,3) and shs_rest the (P,M-1,3) remainder, the concatenation of Gaussians.get_features is never built; `raw_parameters` --
        opacities are logits, scales log-scales, rotations unnormalised: the activations of Model.py:45-87 run inside the preprocess kernel and the
        gradients come back w.r.t. the raw tensors."""
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])
        return rasterize_gaussians(means3D, means2D, empty if shs is None else shs, empty if colors_precomp is None else colors_precomp, opacities,
                                   empty if scales is None else scales, empty if rotations is None else rotations,
                                   empty if cov3D_precomp is None else cov3D_precomp, self.raster_settings, shs_rest, raw_parameters, rest_step)
