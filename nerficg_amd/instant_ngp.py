"""nerficg_amd.instant_ngp -- host side of the InstantNGP hot path on top of libnerficg_hip.so.

The reference's method package (src/Methods/InstantNGP/{Model,Renderer}.py) keeps running unchanged over nerficg_amd/shims;
this module is NOT a copy of it.  It is the host code this repository needs for itself -- tests, bench.py, the data-parallel tools -- with
the reference's configuration keys and method names where a caller would look for them, and its own structure underneath:

  InstantNGPModel        the parameters and buffers of Model.py:14-123 (same yaml keys, same flat parameter vectors, same buffers).
  InstantNGPRenderer     render_rays(train)  : box test -> deterministic march (optional explicit jitter, so that data-parallel ranks can slice
                                               one seeded vector) -> ONE autograd node for both networks -> wave-per-ray compositing.
                         render_rays(eval)   : an arbitrary ray list in chunks: march everything, query everything, composite with the
                                               early-out (the reference's alive-ray loop, Renderer.py:86-138, produces the same pixels; that
                                               loop's two ops raymarching_test / composite_test_fw stay in the C ABI for the reference's own
                                               Renderer.py and are tested against the oracle directly).
                         render_image[_fused]: the tile-interleaved device pipeline (include/nerficg_hip.h group 6), one host sync.
                         update_occupancy_grid / carve_occupancy_grid: cell choice, query positions, EMA, threshold, packing and carving
                                               are device ops (group 11); nothing is read back.
"""
from __future__ import annotations

import ctypes
import math
import warnings
from dataclasses import dataclass, field

import numpy as np
import torch

from . import VolumeRenderingV2 as VolumeRenderingCuda
from . import _lib
from . import tinycudann as tcnn
from .ngp import composite_over_background, query_fused, query_train


def next_multiple(value, multiple: int) -> int:
    """src/Methods/InstantNGP/utils.py:10-11"""
    return int(((value + multiple - 1) // multiple) * multiple)


@dataclass
class Camera:
    """The fields of PerspectiveCamera / SharedCameraSettings the hot path reads (src/Cameras/Perspective.py:16-37)."""
    width: int
    height: int
    focal_x: float
    focal_y: float
    center_x: float | None = None
    center_y: float | None = None
    near_plane: float = 0.2
    far_plane: float = 1000.0
    background_color: torch.Tensor = field(default_factory=lambda: torch.ones(3))

    def __post_init__(self) -> None:
        if self.center_x is None:
            self.center_x = self.width / 2
        if self.center_y is None:
            self.center_y = self.height / 2


class InstantNGPModel(torch.nn.Module):
    """src/Methods/InstantNGP/Model.py:15-123 (defaults = the @Framework.Configurable.configure block :14-30)."""

    def __init__(self, SCALE: float = 0.5, RESOLUTION: int = 128, CENTER=(0.0, 0.0, 0.0), HASHGRID_N_LEVELS: int = 16,
                 HASHGRID_N_FEATURES_PER_LEVEL: int = 2, HASHGRID_LOG2_SIZE: int = 19, HASHGRID_BASE_RESOLUTION: int = 16,
                 HASHGRID_TARGET_RESOLUTION: int = 2048, N_DENSITY_OUTPUT_FEATURES: int = 16, N_DENSITY_NEURONS: int = 64,
                 N_DENSITY_LAYERS: int = 1, DIR_SH_ENCODING_DEGREE: int = 4, N_COLOR_NEURONS: int = 64, N_COLOR_LAYERS: int = 2,
                 RANDOM_SEED: int = 1618033989, device: str | torch.device = 'cuda') -> None:
        super().__init__()
        self.SCALE, self.RESOLUTION, self.CENTER = SCALE, RESOLUTION, list(CENTER)
        # the configurable parameters travel with checkpoints (Base/Model.py:103-111; nerficg_amd.formats)
        self.HASHGRID_N_LEVELS, self.HASHGRID_N_FEATURES_PER_LEVEL, self.HASHGRID_LOG2_SIZE = HASHGRID_N_LEVELS, HASHGRID_N_FEATURES_PER_LEVEL, HASHGRID_LOG2_SIZE
        self.HASHGRID_BASE_RESOLUTION, self.HASHGRID_TARGET_RESOLUTION = HASHGRID_BASE_RESOLUTION, HASHGRID_TARGET_RESOLUTION
        self.N_DENSITY_OUTPUT_FEATURES, self.N_DENSITY_NEURONS, self.N_DENSITY_LAYERS = N_DENSITY_OUTPUT_FEATURES, N_DENSITY_NEURONS, N_DENSITY_LAYERS
        self.DIR_SH_ENCODING_DEGREE, self.N_COLOR_NEURONS, self.N_COLOR_LAYERS = DIR_SH_ENCODING_DEGREE, N_COLOR_NEURONS, N_COLOR_LAYERS
        dev = torch.device(device)
        extent = float(SCALE)
        # the scene box [-SCALE, SCALE]^3 around CENTER and the cascade count of the multi-resolution occupancy grid (Model.py:46-52)
        self.center = torch.tensor(self.CENTER, dtype=torch.float32, device=dev).reshape(1, 3)
        self.xyz_max = torch.full((1, 3), extent, device=dev)
        self.xyz_min = -self.xyz_max
        self.xyz_size = self.xyz_max * 2
        self.half_size = self.xyz_max.clone()
        self.cascades = max(1, 1 + math.ceil(math.log2(2 * extent)))
        # cell coordinates in the enumeration order of a meshgrid(indexing='xy') over three aranges: entry i*G^2 + j*G + k is (x=j, y=i, z=k)
        flat = torch.arange(RESOLUTION ** 3, dtype=torch.int32, device=dev)
        self.grid_coords = torch.stack(((flat // RESOLUTION) % RESOLUTION, flat // RESOLUTION ** 2, flat % RESOLUTION), dim=1).contiguous()
        n_cells = RESOLUTION ** 3
        self.register_buffer('occupancy_grid', torch.zeros(self.cascades, n_cells, device=dev))
        self.register_buffer('occupancy_bitfield', torch.zeros(self.cascades * n_cells // 8, dtype=torch.uint8, device=dev))
        # geometric growth from the base to the target resolution over the levels (Model.py:61-70)
        growth = math.exp(math.log(HASHGRID_TARGET_RESOLUTION * (2 * SCALE) / HASHGRID_BASE_RESOLUTION) / (HASHGRID_N_LEVELS - 1))
        grid = {'otype': 'Grid', 'type': 'Hash', 'interpolation': 'Linear', 'n_levels': HASHGRID_N_LEVELS,
                'n_features_per_level': HASHGRID_N_FEATURES_PER_LEVEL, 'log2_hashmap_size': HASHGRID_LOG2_SIZE,
                'base_resolution': HASHGRID_BASE_RESOLUTION, 'per_level_scale': growth}
        mlp = lambda width, depth, out: {'otype': 'FullyFusedMLP', 'activation': 'ReLU', 'output_activation': out, 'n_neurons': width, 'n_hidden_layers': depth}
        self.encoding_xyz = tcnn.NetworkWithInputEncoding(3, N_DENSITY_OUTPUT_FEATURES, grid, mlp(N_DENSITY_NEURONS, N_DENSITY_LAYERS, 'None'),
                                                          seed=RANDOM_SEED).to(dev)
        # the density MLP's weights sit at the front of encoding_xyz.params (layer widths padded to 16): the slice the weight decay acts on
        widths = [next_multiple(HASHGRID_N_FEATURES_PER_LEVEL * HASHGRID_N_LEVELS, 16)] + [N_DENSITY_NEURONS] * N_DENSITY_LAYERS
        hidden = sum(next_multiple(fan_in * N_DENSITY_NEURONS, 16) for fan_in in widths[:-1])
        self.n_params_encoding_mlp = hidden + next_multiple(N_DENSITY_OUTPUT_FEATURES, 16) * widths[-1]
        direction_and_features = {'otype': 'Composite', 'nested': [{'otype': 'SphericalHarmonics', 'degree': DIR_SH_ENCODING_DEGREE, 'n_dims_to_encode': 3},
                                                                   {'otype': 'Identity'}]}
        self.color_mlp_with_encoding = tcnn.NetworkWithInputEncoding(3 + N_DENSITY_OUTPUT_FEATURES, 3, direction_and_features,
                                                                     mlp(N_COLOR_NEURONS, N_COLOR_LAYERS, 'Sigmoid'), seed=RANDOM_SEED).to(dev)
        self.n_mlp_params = self.n_params_encoding_mlp + self.color_mlp_with_encoding.params.numel()

    def weight_decay_mlp(self) -> torch.Tensor:
        """Mean squared MLP weight over both networks, hash table excluded (Model.py:38-44): one launch, and a backward that costs no launch and no
        dense gradient (nerficg_amd.ngp.weight_decay_mlp)."""
        from .ngp import weight_decay_mlp
        return weight_decay_mlp(self.encoding_xyz, self.color_mlp_with_encoding, self.n_params_encoding_mlp, self.n_mlp_params)


class InstantNGPLoss(torch.nn.Module):
    """src/Methods/InstantNGP/Loss.py:11-26: 'MSE_Color' (weight 1) + 'Weight_Decay_MLP' (weight 0.5e-6) on a training batch, same call signature
    forward(outputs, rays, bg_color) with `rays` anything that has `.rgb` and `.alpha` (RayBatch) or a dict with those keys.  One node / one launch
    each way (nerficg_amd.ngp.instant_ngp_loss); `last` holds (loss, mse, weight decay) of the latest call on the device, `psnr()` the quality
    metric the reference logs (computed when asked, not every iteration)."""

    def __init__(self, model: InstantNGPModel, weight_decay_weight: float = 1.0e-6 / 2.0) -> None:
        super().__init__()
        self.model, self.weight_decay_weight = model, float(weight_decay_weight)
        self.last: torch.Tensor | None = None

    def forward(self, outputs: dict, rays, bg_color: torch.Tensor) -> torch.Tensor:
        from .ngp import instant_ngp_loss
        field = (lambda k: rays.get(k)) if isinstance(rays, dict) else (lambda k: getattr(rays, k, None))
        rgb, alpha = field('rgb'), field('alpha')
        if alpha is not None:   # apply_background_color (Datasets/utils.py:185-189)
            rgb = torch.lerp(bg_color.to(rgb.device), rgb, alpha.reshape(-1, 1)).clamp(0, 1)
        m = self.model
        loss, self.last = instant_ngp_loss(outputs['rgb'], rgb, m.encoding_xyz, m.color_mlp_with_encoding, m.n_params_encoding_mlp, m.n_mlp_params,
                                           self.weight_decay_weight)
        return loss

    def psnr(self) -> torch.Tensor:
        """10 log10(1 / MSE) of the latest batch (torchmetrics' peak_signal_noise_ratio with data_range 1, Loss.py:17)"""
        return -10.0 * torch.log10(self.last[1])


class InstantNGPRenderer:
    """Rendering + occupancy maintenance for an InstantNGPModel.  Configuration keys as in src/Methods/InstantNGP/Renderer.py:141-145."""

    RAY_CHUNK = 1 << 16        # rays per pass of the ray-list inference path
    T_THRESHOLD = 1e-4         # transmittance below which a ray is finished (Renderer.py:79,127)
    COUNT_MAILBOX = True       # fused image path: the row count reaches the host through a mapped host mailbox (False: device-to-host copy)
    POSE_ENCODER_SHAPE = True  # fused image path: the encoder's brick of samples per wave follows the camera's axes (False: the fixed 8 x 2 x 4 default)
    ARENA_IN_PLACE = True      # fused image path, single pass: the query / compositing kernels read the samples from the count pass's arena (False: copied to compact rows first)

    def __init__(self, model: InstantNGPModel, MAX_SAMPLES: int = 1024, EXPONENTIAL_STEPS: bool = False, DENSITY_THRESHOLD: float = 0.01) -> None:
        self.model = model
        self.MAX_SAMPLES, self.EXPONENTIAL_STEPS, self.DENSITY_THRESHOLD = MAX_SAMPLES, EXPONENTIAL_STEPS, DENSITY_THRESHOLD
        self.density_threshold = DENSITY_THRESHOLD * MAX_SAMPLES / 3 ** 0.5
        self._fused_ws: dict = {}
        self._box_host = None
        self._scene_box_host = None
        self._zero_center = torch.zeros(1, 3, device=model.center.device)
        # fused image path: the count pass parks the samples in a max_samples-row arena per tile (2.6 GB at 800x800) and the write pass
        # copies them instead of marching every ray twice; False = second march, no arena
        self.provisional_march = True
        # training batches: None = sample buffers sized from the marched count (one host read per iteration, like the reference); an int = fixed
        # number of sample rows and no host read (nerficg_amd.graphs captures the iteration in a HIP graph with this set)
        self.sample_capacity: int | None = None

    # ---------------------------------------------------------------- the two networks
    @staticmethod
    def _host_copies(cache, tensors):
        """host copies of small device tensors, re-read only when one of them is another object or was written to (identity + version counter): a
        checkpoint loaded into the model, or a box set by hand, reaches the next frame (advisor finding of round 4: the copies were read once per renderer)"""
        key = tuple((id(t), t._version) for t in tensors)
        if cache is None or cache[0] != key:
            cache = (key, tuple(t.detach().float().reshape(-1).cpu().contiguous() for t in tensors))
        return cache

    def _box(self):
        m = self.model
        self._box_host = self._host_copies(self._box_host, (m.xyz_min, m.xyz_size))
        return self._box_host[1]

    def _scene_box(self):
        """(centre, half size) of the scene box as host tensors; read again when the model's buffers change."""
        m = self.model
        self._scene_box_host = self._host_copies(self._scene_box_host, (m.center, m.half_size))
        return self._scene_box_host[1]

    def query(self, xyzs: torch.Tensor, dirs: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """(density, colour) of box-centred sample positions seen along unit directions.  With autograd on: one node that reaches both
        parameter vectors (nerficg_amd.ngp.query_train); otherwise the two-kernel inference query."""
        m = self.model
        if torch.is_grad_enabled() and m.encoding_xyz.params.requires_grad:
            return query_train(m.encoding_xyz, m.color_mlp_with_encoding, xyzs, dirs, *self._box())
        unit = ((xyzs.float() - m.xyz_min) / m.xyz_size).contiguous()
        return query_fused(m.encoding_xyz, m.color_mlp_with_encoding, unit, dirs.float().contiguous())

    @torch.no_grad()
    def density(self, xyzs: torch.Tensor) -> torch.Tensor:
        """exp (in f32) of the first output of the density network at box-centred positions."""
        m = self.model
        return m.encoding_xyz((xyzs - m.xyz_min) / m.xyz_size)[:, 0].float().exp()

    # ---------------------------------------------------------------- rays
    def clip_rays(self, origin: torch.Tensor, view_direction: torch.Tensor, camera: Camera):
        """Box-centred origins, contiguous directions and the [t_in, t_out] interval of every ray inside the scene box, clipped to the
        camera's depth range (a miss is (near, -1): an empty interval) -- one launch (nrc_ngp_clip_rays)."""
        origin, d = origin.contiguous(), view_direction.contiguous()
        _lib.check_input(origin, 'origin', torch.float32)
        _lib.check_input(d, 'view_direction', torch.float32)
        n = origin.shape[0]
        o = torch.empty_like(origin)
        span = torch.empty(n, 2, dtype=torch.float32, device=origin.device)
        center, half = self._scene_box()
        _lib.check(_lib.load().nrc_ngp_clip_rays(n, _lib.ptr(origin), _lib.ptr(d), _lib.ptr(center), _lib.ptr(half), float(camera.near_plane),
                                                 float(camera.far_plane), _lib.ptr(o), _lib.ptr(span), _lib.stream_of(origin)), 'ngp_clip_rays')
        return o, d, span

    def render_rays(self, origin: torch.Tensor, view_direction: torch.Tensor, camera: Camera, train_mode: bool = False,
                    custom_bg_color: torch.Tensor | None = None, noise: torch.Tensor | None = None) -> dict[str, torch.Tensor]:
        """rgb / alpha / depth (+ 'rm_samples' when training) for a list of rays.  `noise` (one value in [0,1) per ray) replaces the
        training jitter draw."""
        bg = (custom_bg_color if custom_bg_color is not None else camera.background_color).to(origin.device)
        step_growth = 1 / 256 if self.EXPONENTIAL_STEPS else 0.0
        o, d, span = self.clip_rays(origin.float(), view_direction.float(), camera)
        if train_mode:
            return self._render_training_batch(o, d, span, bg, step_growth, noise)
        return self._render_ray_list(o, d, span, bg, step_growth)

    def _march(self, o, d, span, step_growth, jitter):
        m = self.model
        return VolumeRenderingCuda.raymarching_train(o, d, span, m.occupancy_bitfield, m.cascades, m.SCALE, step_growth, jitter, m.RESOLUTION,
                                                     self.MAX_SAMPLES, sample_capacity=self.sample_capacity, return_overflow=True)

    def _render_training_batch(self, o, d, span, bg, step_growth, noise):
        jitter = torch.rand(o.shape[0], device=o.device) if noise is None else noise.to(torch.float32).contiguous()
        rays_a, xyzs, dirs, deltas, ts, counter, *cut = self._march(o, d, span, step_growth, jitter)
        sigmas, rgbs = self.query(xyzs, dirs)
        # compositing, background and the training depth (weighted mean with a guarded denominator, Renderer.py:78-84) as one autograd node
        rgb, alpha, depth = composite_over_background(sigmas, rgbs, deltas, ts, rays_a, bg, self.T_THRESHOLD)
        # 'rm_samples' (Trainer.py:94 reads it with .item() every iteration): without a fixed capacity the host sized the sample buffers from that
        # very count a moment ago -- it comes back as a HOST scalar tensor, and the trainer's .item() is not a second stream synchronisation
        marched = counter[0] if self.sample_capacity is not None else torch.tensor(xyzs.shape[0], dtype=torch.int32)
        out = {'rgb': rgb, 'alpha': alpha, 'depth': depth, 'rm_samples': marched}
        if cut:   # fixed sample capacity: how many samples did not fit (device int64)
            out['sample_overflow'] = cut[0]
        return out

    @torch.no_grad()
    def _render_ray_list(self, o, d, span, bg, step_growth):
        n = o.shape[0]
        rgb, alpha, depth = (torch.empty(n, 3, device=o.device), torch.empty(n, device=o.device), torch.empty(n, device=o.device))
        for lo in range(0, n, self.RAY_CHUNK):
            hi = min(n, lo + self.RAY_CHUNK)
            jitter = torch.zeros(hi - lo, device=o.device)
            rays_a, xyzs, dirs, deltas, ts, *_ = self._march(o[lo:hi].contiguous(), d[lo:hi].contiguous(), span[lo:hi].contiguous(), step_growth, jitter)
            if xyzs.shape[0]:
                sigmas, rgbs = self.query(xyzs, dirs)
            else:
                sigmas, rgbs = torch.empty(0, device=o.device), torch.empty(0, 3, device=o.device)
            _, a, dsum, rad, _ = VolumeRenderingCuda.composite_train_fw(sigmas, rgbs, deltas, ts, rays_a, self.T_THRESHOLD)
            a.clamp_(0, 1)
            rgb[lo:hi] = (rad + (1 - a)[:, None] * bg).clamp_(0, 1)
            alpha[lo:hi] = a
            depth[lo:hi] = torch.where(a > 0, dsum / a, torch.zeros_like(a))  # inference depth: plain weighted mean, 0 where nothing was hit (Renderer.py:137)
        return {'rgb': rgb, 'alpha': alpha, 'depth': depth}

    def render_image(self, camera: Camera, c2w: np.ndarray, to_chw: bool = False) -> dict[str, torch.Tensor]:
        """(H, W, C) images (or (C, H, W)) of one view through the fused pipeline."""
        flat = self.render_image_fused(camera, c2w)
        out = {}
        for key in ('rgb', 'alpha', 'depth'):
            img = flat[key].reshape(camera.height, camera.width, -1).clone()
            out[key] = img.permute(2, 0, 1) if to_chw else img
        return out

    @torch.no_grad()
    def _render_image_general(self, camera: Camera, c2w) -> dict[str, torch.Tensor]:
        """One frame as a list of rays through render_rays (march -> query through the tcnn modules -> composite): flat (H*W, C) buffers like the fused frame."""
        from .raygen import generate_rays
        dev = self.model.center.device
        c2w = c2w.detach().cpu().numpy() if isinstance(c2w, torch.Tensor) else c2w
        rays = generate_rays(camera.width, camera.height, camera.focal_x, camera.focal_y, camera.center_x, camera.center_y, c2w, device=dev)
        out = self.render_rays(rays['origin'], rays['view_direction'], camera, train_mode=False)
        return {'rgb': out['rgb'], 'alpha': out['alpha'], 'depth': out['depth']}

    # ---------------------------------------------------------------- MI355X-native image pipeline
    @staticmethod
    def n_image_tiles(camera: Camera) -> int:
        lib = _lib.load()
        tw, th = int(lib.nrc_ngp_tile_width()), int(lib.nrc_ngp_tile_height())
        return ((camera.width + tw - 1) // tw) * ((camera.height + th - 1) // th)

    # -- the stages of the fused image pipeline (one frame = frame constants + a workspace per rendered tile range) --------------------------
    def _frame_constants(self, camera: Camera, c2w) -> dict:
        m = self.model
        c2w = np.ascontiguousarray(np.asarray(c2w, dtype=np.float64))
        if c2w.shape == (3, 4):
            c2w = np.vstack([c2w, [0.0, 0.0, 0.0, 1.0]])
        f3 = lambda t: (ctypes.c_float * 3)(*[float(v) for v in t.reshape(-1).tolist()])
        g = m.encoding_xyz.grid_cfg
        # the model's box lives in device buffers: read ONCE per renderer (host copies, _box / _scene_box).  Until round 4 every frame read the four
        # vectors back (`.tolist()` of a device tensor = a stream synchronisation each): four host round trips in front of every frame's first launch,
        # and a read-back inside a stream capture.
        center, half = self._scene_box()
        mn, sz = self._box()
        bg = camera.background_color
        if bg.is_cuda:   # a colour that lives on the device is read once per tensor object / version
            hit = self.__dict__.get('_bg_host')
            if hit is None or hit[0] is not bg or hit[1] != bg._version:
                hit = self.__dict__['_bg_host'] = (bg, bg._version, bg.detach().float().cpu())
            bg = hit[2]
        # The brick of samples one wave of the encoder gathers for (nrc_ngp_set_encoder_shape).  Hash-table entries are contiguous along WORLD x only, so
        # the brick that shares the most cache lines depends on which of the camera's axes runs along x: image rows (the default 8 x 2 pixels x 4 steps),
        # image columns (4 x 4 x 1), or the viewing direction (4 x 2 x 8).  Measured per pose on the bench orbit (tools/exp_pose_shapes.py): the choice
        # below loses to the best of twelve shapes by < 1 % on every pose and saves 2 % of the encoder over the fixed default.
        right_x, down_x = abs(float(c2w[0, 0])), abs(float(c2w[0, 1]))
        shape = (3, 1) if right_x >= 0.6 else ((2, 2) if down_x >= 0.35 else (2, 1))
        return dict(enc_shape=shape, intr=(ctypes.c_double * 4)(camera.focal_x, camera.focal_y, camera.center_x, camera.center_y),
                    mat=(ctypes.c_double * 16)(*c2w.reshape(-1).tolist()), center=f3(center), half=f3(half), mn=f3(mn), sz=f3(sz),
                    bg=f3(bg.float()), esf=1 / 256 if self.EXPONENTIAL_STEPS else 0.0, grid=g, camera=camera,
                    hw=camera.width * camera.height)

    def _fused_workspace(self, store: dict, key, nt: int, hw: int, dev, images: bool) -> dict:
        ws = store.get(key)
        if ws is None:
            lib = _lib.load()
            n = nt * 64
            ws = dict(ray_od=torch.empty(max(n, 1), 6, device=dev), ray_t=torch.empty(max(n, 1), 2, device=dev),
                      ray_cnt=torch.empty(max(n, 1), dtype=torch.int32, device=dev), tile_rows=torch.empty(2 * max(nt, 1), dtype=torch.int32, device=dev),
                      tile_off=torch.empty(nt + 1, dtype=torch.int32, device=dev), counter=torch.empty(2, dtype=torch.int32, device=dev), cap=0,
                      skipped=torch.zeros(1, dtype=torch.int32, device=dev))
            if images:
                ws.update(rgb=torch.zeros(hw, 3, device=dev), alpha=torch.zeros(hw, device=dev), depth=torch.zeros(hw, device=dev))
            if self.provisional_march:  # samples parked by the count pass, copied (not re-marched) by the write pass
                ws['ts_prov'] = torch.empty(int(lib.nrc_ngp_render_provisional_bytes(nt, self.MAX_SAMPLES)), dtype=torch.uint8, device=dev)
            store[key] = ws
        return ws

    def _fused_count(self, fc: dict, ws: dict, tile_begin: int, nt: int, mailbox=None) -> int:
        """the count pass; with a `mailbox` (_lib.HostMailbox) the totals also go straight to host memory: returns the ticket to wait for"""
        m, lib, cam = self.model, _lib.load(), fc['camera']
        vp = ctypes.c_void_p
        ticket = mailbox.next_ticket() if mailbox is not None else 0
        _lib.check(lib.nrc_ngp_render_count(
            cam.width, cam.height, ctypes.cast(fc['intr'], vp), ctypes.cast(fc['mat'], vp), ctypes.cast(fc['center'], vp), ctypes.cast(fc['half'], vp),
            float(cam.near_plane), float(cam.far_plane), int(tile_begin), nt, _lib.ptr(m.occupancy_bitfield), m.cascades, float(m.SCALE), float(fc['esf']),
            m.RESOLUTION, self.MAX_SAMPLES, _lib.ptr(ws['ray_od']), _lib.ptr(ws['ray_t']), _lib.ptr(ws['ray_cnt']), _lib.ptr(ws['tile_rows']),
            _lib.ptr(ws['tile_off']), _lib.ptr(ws['counter']), _lib.ptr(ws.get('ts_prov')), mailbox.ptr if mailbox is not None else None, ticket,
            _lib.stream_of(ws['ray_od'])), 'ngp_render_count')
        return ticket

    def _fused_size_rows(self, ws: dict, rows: int, nt: int) -> None:
        if rows > ws['cap']:
            lib = _lib.load()
            dev = ws['ray_od'].device
            cap = int(rows * 1.5) + 64  # poses of one scene differ by up to 30 % in rows: grow rarely (a regrowth costs milliseconds of hipMalloc)
            ws.update(ts=torch.empty(cap * 64, device=dev), row_tile=torch.empty(cap, dtype=torch.int32, device=dev),
                      packed=torch.empty(cap * 64, 4, dtype=torch.float16, device=dev),
                      qws=torch.empty(int(lib.nrc_ngp_render_layers_ws_bytes(cap, nt)), dtype=torch.uint8, device=dev),
                      row_of=torch.empty(cap, dtype=torch.int32, device=dev), row_k=torch.empty(cap, dtype=torch.int32, device=dev),
                      layer_off=torch.empty(self.MAX_SAMPLES + 2, dtype=torch.int32, device=dev), cap=cap)

    def _query_args(self, fc: dict, ws: dict, nt: int, fixed: bool, arena: bool) -> list:
        """argument list of nrc_ngp_query_samples with the row count (index 3) still open: marshalled BEFORE the host waits for that count, so that
        nothing but the call itself stands between the count's arrival and the launch"""
        m, vp, g = self.model, ctypes.c_void_p, fc['grid']
        return [_lib.ptr(ws['ts_prov'] if arena else ws['ts']), _lib.ptr(ws['row_tile']), _lib.ptr(ws['ray_od']), 0, nt, ctypes.cast(fc['mn'], vp),
                ctypes.cast(fc['sz'], vp), _lib.ptr(m.encoding_xyz._half_params()), _lib.ptr(m.color_mlp_with_encoding._half_params()),
                _lib.ptr(m.encoding_xyz._table16()), g['n_levels'], g['log2_hashmap_size'], g['base_resolution'], float(g['per_level_scale']),
                _lib.ptr(ws['packed']), _lib.ptr(ws['qws']), _lib.ptr(ws['counter']) if fixed else None,
                _lib.ptr(ws['tile_off']) if arena else None, self.MAX_SAMPLES if arena else 0, _lib.stream_of(ws['ray_od'])]

    def _fused_write_query(self, fc: dict, ws: dict, rows: int, nt: int, fixed: bool = False, arena: bool = False, query_args: list | None = None) -> None:
        """single pass: the parked samples into their final rows, then encode + MLPs over all of them.  fixed: `rows` is a CAPACITY -- rows behind it
        are not written, and the query kernels read the number of rows that exist from the device counter.  arena: the parked samples are queried
        where the count pass left them (include/nerficg_hip.h, "the arena queried in place"): no copy into compact rows, no write pass at all"""
        if rows <= 0:
            return
        m, lib = self.model, _lib.load()
        if not arena:
            _lib.check(lib.nrc_ngp_render_write(nt, _lib.ptr(m.occupancy_bitfield), m.cascades, float(m.SCALE), float(fc['esf']), m.RESOLUTION,
                                                self.MAX_SAMPLES, _lib.ptr(ws['ray_od']), _lib.ptr(ws['ray_t']), _lib.ptr(ws['ray_cnt']),
                                                _lib.ptr(ws['tile_off']), _lib.ptr(ws['ts']), _lib.ptr(ws['row_tile']), _lib.ptr(ws.get('ts_prov')),
                                                int(rows) if fixed else 0, _lib.stream_of(ws['ray_od'])), 'ngp_render_write')
        args = query_args if query_args is not None else self._query_args(fc, ws, nt, fixed, arena)
        args[3] = rows
        if self.POSE_ENCODER_SHAPE:
            lib.nrc_ngp_set_encoder_shape(*fc['enc_shape'])
        _lib.check(lib.nrc_ngp_query_samples(*args), 'ngp_query_samples')

    def _fused_composite(self, fc: dict, ws: dict, out: dict, tile_begin: int, nt: int, row_capacity: int = 0, arena: bool = False) -> None:
        m, lib, cam = self.model, _lib.load(), fc['camera']
        _lib.check(lib.nrc_ngp_composite_image(
            _lib.ptr(ws.get('packed')), _lib.ptr(ws.get('ts_prov') if arena else ws.get('ts')), _lib.ptr(ws['ray_cnt']), _lib.ptr(ws['tile_off']), cam.width,
            cam.height, int(tile_begin), nt, m.cascades, float(fc['esf']), m.RESOLUTION, self.MAX_SAMPLES, 1e-4, ctypes.cast(fc['bg'], ctypes.c_void_p),
            _lib.ptr(out['rgb']), _lib.ptr(out['alpha']), _lib.ptr(out['depth']), int(row_capacity), self.MAX_SAMPLES if arena else 0,
            _lib.stream_of(ws['ray_od'])), 'ngp_composite_image')

    @torch.no_grad()
    def render_image_pipelined(self, camera: Camera, c2w: np.ndarray, shards: int = 4, return_stats: bool = False) -> dict[str, torch.Tensor]:
        """The single-pass image of render_image_fused with the ray march of tile range k + 1 running NEXT TO the encode / MLP kernels of range k
        (round 4).  A frame is count -> [host read of the row count] -> write -> 10 x (encode, MLP) -> composite, strictly in sequence; the two
        march kernels (divergent DDA, latency-bound: 0.36 + 0.14 ms of an 8.35 ms frame at 800x800) and the host read leave the chip almost
        idle.  Here the frame is cut into `shards` contiguous tile ranges (the ranges of parallel.shard_range: they compose bit for bit,
        tests/test_gpu_render_parity.py); the march of a range runs on a side stream, its encode / MLP / composite on the caller's stream behind
        an event, and the host blocks on the NEXT range's row count only after it has enqueued the current range's long kernels.  Same
        per-ray arithmetic, same pixels (tests/test_gpu_render_parity.py).
        MEASURED SLOWER, kept as an experiment (tools/bench_pipeline.py, 800x800 bench frame, ms per frame): one pass 8.67; 2 / 3 / 4 / 6 / 8
        ranges 8.82 / 8.91 / 8.98 / 9.29 / 9.56 -- next to the march the encoder (bound by its L1 / texture-address path) loses more than the
        0.4 ms the overlap hides, and every range ends in a partly filled launch.  The same was seen in the 3DGS forward (a bandwidth-heavy pass
        next to the latency-bound sort: NRC_GS_OVERLAP) -- on this chip concurrent kernels do not add up."""
        from .parallel import shard_range
        m = self.model
        dev = m.center.device
        total_tiles = self.n_image_tiles(camera)
        shards = max(1, min(int(shards), total_tiles))
        fc = self._frame_constants(camera, c2w)
        hw = fc['hw']
        store = self.__dict__.setdefault('_pipe_ws', {})
        if store.get('_shape') != (total_tiles, hw, shards, str(dev)):
            store.clear()
            store['_shape'] = (total_tiles, hw, shards, str(dev))
            store['_side'] = torch.cuda.Stream(device=dev)
            store['_images'] = dict(rgb=torch.zeros(hw, 3, device=dev), alpha=torch.zeros(hw, device=dev), depth=torch.zeros(hw, device=dev))
        side, out = store['_side'], store['_images']
        main = torch.cuda.current_stream(dev)
        ranges = [shard_range(total_tiles, k, shards) for k in range(shards)]
        wss = [self._fused_workspace(store, ('ws', k), e - b, hw, dev, images=False) for k, (b, e) in enumerate(ranges)]
        side.wait_stream(main)          # the march reads the occupancy bitfield / writes buffers the caller's stream may still be using
        n_rows = n_samples = 0
        with torch.cuda.stream(side):
            self._fused_count(fc, wss[0], ranges[0][0], ranges[0][1] - ranges[0][0])
        for k, ((b, e), ws) in enumerate(zip(ranges, wss)):
            nt = e - b
            with torch.cuda.stream(side):
                rows, samples = ws['counter'].tolist()     # waits for the side stream only: the caller's stream keeps computing the previous range
            # (re)allocation in the CALLER's stream context: a freed buffer goes back to that stream's pool, behind the kernels that still use it
            self._fused_size_rows(ws, rows, nt)
            with torch.cuda.stream(side):
                # the write pass stays on the side stream too (it only copies the parked samples)
                if rows > 0:
                    lib = _lib.load()
                    _lib.check(lib.nrc_ngp_render_write(nt, _lib.ptr(m.occupancy_bitfield), m.cascades, float(m.SCALE), float(fc['esf']), m.RESOLUTION,
                                                        self.MAX_SAMPLES, _lib.ptr(ws['ray_od']), _lib.ptr(ws['ray_t']), _lib.ptr(ws['ray_cnt']),
                                                        _lib.ptr(ws['tile_off']), _lib.ptr(ws['ts']), _lib.ptr(ws['row_tile']), _lib.ptr(ws.get('ts_prov')),
                                                        0, _lib.stream_of(ws['ray_od'])), 'ngp_render_write')
                marched = torch.cuda.Event()
                marched.record(side)
            n_rows += rows
            n_samples += samples
            main.wait_event(marched)
            if rows > 0:
                g = fc['grid']
                vp = ctypes.c_void_p
                _lib.check(_lib.load().nrc_ngp_query_samples(
                    _lib.ptr(ws['ts']), _lib.ptr(ws['row_tile']), _lib.ptr(ws['ray_od']), rows, nt, ctypes.cast(fc['mn'], vp), ctypes.cast(fc['sz'], vp),
                    _lib.ptr(m.encoding_xyz._half_params()), _lib.ptr(m.color_mlp_with_encoding._half_params()), _lib.ptr(m.encoding_xyz._table16()),
                    g['n_levels'], g['log2_hashmap_size'], g['base_resolution'], float(g['per_level_scale']), _lib.ptr(ws['packed']), _lib.ptr(ws['qws']),
                    None, None, 0, _lib.stream_of(ws['ray_od'])), 'ngp_query_samples')
            self._fused_composite(fc, ws, out, b, nt)
            if k + 1 < shards:     # the next range's count pass starts now, under the kernels just enqueued; its buffers were last used a frame ago
                nb, ne = ranges[k + 1]
                with torch.cuda.stream(side):
                    self._fused_count(fc, wss[k + 1], nb, ne - nb)
        res = dict(out)
        if return_stats:
            res.update(n_rows=n_rows, n_slots=n_rows * 64, n_samples=n_samples)
        return res

    @torch.no_grad()
    def render_image_fused(self, camera: Camera, c2w: np.ndarray, tile_begin: int = 0, n_tiles: int | None = None,
                           return_stats: bool = False, out: dict | None = None, early_termination: bool | str = 'auto',
                           row_capacity: int | None = None) -> dict[str, torch.Tensor]:
        """Same image as render_image, flat (H*W, C) pixel-major buffers, through the tile-interleaved device pipeline
        (include/nerficg_hip.h group 6): four device stages and ONE host sync (the row count, to size the sample buffers).
        A shard renders the 8x8-pixel tiles [tile_begin, tile_begin + n_tiles) and writes only their pixels (pass `out` to
        accumulate several shards into the same buffers).  early_termination: rows in layer order, depth slabs composited front to
        back, finished tiles skipped in the following slabs (Renderer.py:118-132 at slab granularity); False: one pass over all
        samples.  Both give the same image.  'auto' (default): slabs as long as they pay -- when a slab frame saved less than 10 % of
        its rows (e.g. an untrained model: no ray ever saturates, and the slab bookkeeping costs ~7 %) the next 32 frames of this
        renderer take the single pass, then one slab frame probes again.
        row_capacity (round 4): a frame WITHOUT the host read -- the sample buffers hold `row_capacity` rows of 64 slots, the single pass is
        enqueued for that capacity, the kernels take the number of rows that exist from the device counter, and nothing is read back: the call only
        enqueues (it can sit inside a stream capture).  The result carries 'counter' (DEVICE int32[2]: rows, samples): rows > row_capacity means
        the frame did not fit and its picture is not valid -- look at it when convenient and render again with more."""
        m = self.model
        from .ngp import default_layout
        if not default_layout(m.encoding_xyz, m.color_mlp_with_encoding):
            # the tile pipeline's kernels are built for the shipped 16 x 2 grid / degree-4 SH; any other yaml configuration renders through the
            # drop-in modules, ray list by ray list (what the reference's render_image does, Renderer.py:100-138)
            if tile_begin or n_tiles is not None or row_capacity is not None or out is not None:
                raise RuntimeError('render_image_fused: tile shards / fixed-capacity frames need the default HASHGRID_N_LEVELS = 16, HASHGRID_N_FEATURES_PER_LEVEL = 2, '
                                   'DIR_SH_ENCODING_DEGREE = 4; this model renders whole frames through the general path')
            return self._render_image_general(camera, c2w)
        lib = _lib.load()
        dev = m.center.device
        total_tiles = self.n_image_tiles(camera)
        nt = total_tiles - tile_begin if n_tiles is None else int(n_tiles)
        fc = self._frame_constants(camera, c2w)
        hw = fc['hw']
        key = (nt, hw, str(dev))
        if key not in self._fused_ws:
            self._fused_ws = {}
        ws = self._fused_workspace(self._fused_ws, key, nt, hw, dev, images=True)
        if out is None:
            out = {'rgb': ws['rgb'], 'alpha': ws['alpha'], 'depth': ws['depth']}
        st = _lib.stream_of(ws['ray_od'])
        esf = fc['esf']
        # the frame's one host read: the row count that sizes the sample buffers.  Through a host mailbox (the closing scan of the count pass stores
        # the totals in mapped host memory and this thread polls them) when the runtime offers one -- no device-to-host copy, no stream wait:
        # 42 -> ~10 us of idle GPU between the count pass and the write pass
        if row_capacity is not None:
            self._fused_count(fc, ws, tile_begin, nt)
            cap = int(row_capacity)
            if cap < 1:
                raise ValueError('row_capacity must be positive')
            if ws['cap'] < cap:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError('render_image_fused(row_capacity=...): call it once outside the capture so that the sample buffers exist')
                self._fused_size_rows(ws, cap, nt)        # (grows with the usual slack: 1.5 cap + 64 rows)
                ws['row_tile'].zero_()                     # rows that were never written must name a tile that exists
                ws['ts'].fill_(-1.0)                       # ... and hold no sample
            self._fused_write_query(fc, ws, cap, nt, fixed=True)
            self._fused_composite(fc, ws, out, tile_begin, nt, row_capacity=cap)
            res = dict(out)
            res['counter'] = ws['counter']
            return res
        mailbox = _lib.HostMailbox.for_device(dev) if self.COUNT_MAILBOX else None
        arena = self.ARENA_IN_PLACE and ws.get('ts_prov') is not None   # the parked samples are queried where they are: no copy into compact rows
        if mailbox is not None:
            mailbox.lock.acquire()    # one call in flight per mailbox
        try:
            ticket = self._fused_count(fc, ws, tile_begin, nt, mailbox)
            # what does not depend on the count is prepared before the wait (the sample buffers exist from the previous frame; a frame that needs
            # more regrows them below and marshals again)
            ready = self._query_args(fc, ws, nt, False, arena) if ws['cap'] > 0 else None
            cap_before = ws['cap']
            counts = mailbox.counts(ticket, dev) if mailbox is not None else None
        finally:
            if mailbox is not None:
                mailbox.lock.release()
        if counts is None:
            if mailbox is not None:
                warnings.warn('render_image_fused: the count mailbox did not answer; reading the device counter from now on')
            counts = ws['counter'].tolist()
        rows, n_samples = counts   # THE host read of the frame: sizes the sample buffers; the marched total comes with it
        self._fused_size_rows(ws, rows, nt)
        if early_termination == 'auto':
            pol = ws.setdefault('et_policy', {'prev_rows': 0, 'prev_layered': False, 'skip_frames': 0})
            if pol['prev_layered'] and pol['prev_rows'] > 0:  # the previous frame is complete by now (we just synchronised on this one's row count)
                # (the previous frame's count of skipped rows was copied to pinned memory behind that frame; this frame's count pass ran behind the
                # copy and has been waited for, so the value is there -- no device read, no stream synchronisation here)
                skipped_prev = int(ws['skipped_host'][0]) if 'skipped_host' in ws else int(ws['skipped'].item())
                if skipped_prev < 0.1 * pol['prev_rows']:
                    pol['skip_frames'] = 32
            use_layers = pol['skip_frames'] == 0
            pol['skip_frames'] = max(0, pol['skip_frames'] - 1)
            pol['prev_rows'], pol['prev_layered'] = rows, use_layers
        else:
            use_layers = bool(early_termination)
        if use_layers and self.MAX_SAMPLES <= 1024:
            g = fc['grid']
            vp = ctypes.c_void_p
            if rows > 0:
                _lib.check(lib.nrc_ngp_render_write_layers(
                    nt, _lib.ptr(m.occupancy_bitfield), m.cascades, float(m.SCALE), float(esf), m.RESOLUTION, self.MAX_SAMPLES, _lib.ptr(ws['ray_od']),
                    _lib.ptr(ws['ray_t']), _lib.ptr(ws['ray_cnt']), _lib.ptr(ws['tile_rows']), _lib.ptr(ws['tile_off']), None if arena else _lib.ptr(ws['ts']),
                    _lib.ptr(ws['row_tile']), _lib.ptr(ws['layer_off']), _lib.ptr(ws['row_of']), _lib.ptr(ws.get('ts_prov')),
                    _lib.ptr(ws['row_k']) if arena else None, st), 'ngp_render_write_layers')
            if 'qws' not in ws:  # an image without a single sample: state + background only
                ws['qws'] = torch.empty(int(lib.nrc_ngp_render_layers_ws_bytes(0, nt)), dtype=torch.uint8, device=dev)
            if self.POSE_ENCODER_SHAPE:
                lib.nrc_ngp_set_encoder_shape(*fc['enc_shape'])
            _lib.check(lib.nrc_ngp_render_layers(
                _lib.ptr(ws.get('ts_prov') if arena else ws.get('ts')), _lib.ptr(ws.get('row_tile')), _lib.ptr(ws['ray_od']), rows, nt, ctypes.cast(fc['mn'], vp),
                ctypes.cast(fc['sz'], vp), _lib.ptr(m.encoding_xyz._half_params()), _lib.ptr(m.color_mlp_with_encoding._half_params()),
                _lib.ptr(m.encoding_xyz._table16()), g['n_levels'], g['log2_hashmap_size'], g['base_resolution'], float(g['per_level_scale']),
                _lib.ptr(ws['ray_cnt']), _lib.ptr(ws['tile_rows']), _lib.ptr(ws['tile_off']), _lib.ptr(ws.get('row_of')), camera.width, camera.height,
                int(tile_begin), m.cascades, float(esf), m.RESOLUTION, self.MAX_SAMPLES, 1e-4, ctypes.cast(fc['bg'], vp),
                _lib.ptr(ws.get('packed')), _lib.ptr(out['rgb']), _lib.ptr(out['alpha']), _lib.ptr(out['depth']), _lib.ptr(ws['skipped']),
                _lib.ptr(ws['qws']), _lib.ptr(ws['row_k']) if (arena and rows > 0) else None, self.MAX_SAMPLES if (arena and rows > 0) else 0, st), 'ngp_render_layers')
            if 'skipped_host' not in ws:
                ws['skipped_host'] = torch.zeros(1, dtype=torch.int32).pin_memory()
            ws['skipped_host'].copy_(ws['skipped'], non_blocking=True)   # for the next frame's policy; ordered behind this frame on the stream
        else:
            self._fused_write_query(fc, ws, rows, nt, arena=arena, query_args=ready if ws['cap'] == cap_before else None)
            self._fused_composite(fc, ws, out, tile_begin, nt, arena=arena)
        res = dict(out)
        if return_stats:
            res['n_rows'] = rows
            res['n_slots'] = rows * 64
            res['n_samples'] = n_samples
        return res

    # ---------------------------------------------------------------- occupancy grid
    def _occupied_cells(self):
        """Per cascade: ascending Morton indices of the cells above the density threshold + their count, device side (no read-back)."""
        m = self.model
        lib = _lib.load()
        grid = m.occupancy_grid
        cells = grid.shape[1]
        above = (grid > self.density_threshold).contiguous()
        idx = torch.empty(m.cascades, cells, dtype=torch.int32, device=grid.device)
        cnt = torch.zeros(m.cascades, dtype=torch.int32, device=grid.device)
        ws = torch.empty(int(lib.nrc_compact_mask_ws_bytes(cells)), dtype=torch.uint8, device=grid.device)
        for c in range(m.cascades):
            _lib.check(lib.nrc_compact_mask(_lib.ptr(above[c]), cells, _lib.ptr(idx[c]), _lib.ptr(cnt[c:]), _lib.ptr(ws), _lib.stream_of(grid)), 'compact_mask')
        return idx, cnt

    @torch.no_grad()
    def draw_update_cells(self, warmup: bool, seed: torch.Tensor | None = None) -> tuple[torch.Tensor, torch.Tensor]:
        """The cells that get a fresh density this round and one jittered query position inside each (Renderer.py:183-206, 251-258): every
        cell while warming up, afterwards G^3/4 uniformly drawn cells + G^3/4 drawn from the occupied ones per cascade.  Returns Morton
        indices (cascades, per) i64 (-1 = ignored entry) and box-centred positions (cascades * per, 3).  `seed`: one i64 on the device
        (default: a draw from torch's device generator, so torch.manual_seed reproduces the sequence)."""
        m = self.model
        lib = _lib.load()
        dev = m.occupancy_grid.device
        if seed is None:
            seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, device=dev)
        g = m.RESOLUTION
        n_half = 0 if warmup else g ** 3 // 4
        per = g ** 3 if warmup else 2 * n_half
        indices = torch.empty(m.cascades, per, dtype=torch.int64, device=dev)
        points = torch.empty(m.cascades * per, 3, dtype=torch.float32, device=dev)
        occ_idx, occ_cnt = (None, None) if warmup else self._occupied_cells()
        _lib.check(lib.nrc_occupancy_draw_cells(m.cascades, g, float(m.SCALE), 0 if warmup else 1, n_half, _lib.ptr(seed), _lib.ptr(occ_idx),
                                                _lib.ptr(occ_cnt), g ** 3, _lib.ptr(indices), _lib.ptr(points), _lib.stream_of(points)), 'occupancy_draw_cells')
        return indices, points

    @torch.no_grad()
    def update_occupancy_grid(self, warmup: bool = False, decay: float = 0.95) -> None:
        """One maintenance round (Renderer.py:247-272): draw cells, query their densities in ONE network call, then scatter / EMA / masked
        mean / threshold / bit packing in one C-ABI call that leaves the threshold on the device (`self.occupancy_threshold`: used, mean)."""
        m = self.model
        lib = _lib.load()
        grid = m.occupancy_grid
        dev = grid.device
        indices, points = self.draw_update_cells(warmup)
        sigma = self.density(points).reshape(-1).contiguous()
        _lib.check_input(grid, 'occupancy_grid', torch.float32)
        ws_bytes = int(lib.nrc_occupancy_update_ws_bytes(grid.numel()))
        if getattr(self, '_occ_ws', None) is None or self._occ_ws.numel() < ws_bytes:
            self._occ_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            self.occupancy_threshold = torch.zeros(2, dtype=torch.float32, device=dev)
        _lib.check(lib.nrc_occupancy_update(_lib.ptr(grid), _lib.ptr(indices), _lib.ptr(sigma), 0 if sigma.dtype == torch.float32 else 1, m.cascades,
                                            grid.shape[1], indices.shape[1], decay, self.density_threshold, _lib.ptr(m.occupancy_bitfield),
                                            _lib.ptr(self.occupancy_threshold), _lib.ptr(self._occ_ws), _lib.stream_of(grid)), 'occupancy_update')

    @torch.no_grad()
    def carve_occupancy_grid(self, views, subtractive: bool = False, use_alpha: bool = False) -> None:
        """Freezes the cells no training camera can see at -1 (never sampled again) and resets the others to 0 (Renderer.py:208-245).
        `views`: iterable of (Camera, c2w (4,4), alpha (1,H,W) tensor or None).  Union of the frusta by default, intersection when
        `subtractive`; with `use_alpha` a cell must also project onto the 3x3-dilated foreground mask.  The survivors are dilated by one cell."""
        m = self.model
        lib = _lib.load()
        dev = m.occupancy_grid.device
        g = m.RESOLUTION
        remaining = torch.full((m.cascades, g ** 3), 1 if subtractive else 0, dtype=torch.uint8, device=dev)
        f = lambda a: (ctypes.c_float * len(a))(*[float(v) for v in a])
        center = f(m.center.reshape(-1).tolist())
        st = _lib.stream_of(remaining)
        for camera, c2w, alpha in views:
            pose = np.asarray(c2w, dtype=np.float64)[:4, :4]
            mask = alpha.to(dev, torch.float32).reshape(camera.height, camera.width).contiguous() if (use_alpha and alpha is not None) else None
            _lib.check(lib.nrc_occupancy_carve_view(
                _lib.ptr(remaining), m.cascades, g, float(m.SCALE), ctypes.cast(center, ctypes.c_void_p), ctypes.cast(f(pose[:3, :3].reshape(-1)), ctypes.c_void_p),
                ctypes.cast(f(pose[:3, 3]), ctypes.c_void_p), float(camera.focal_x), float(camera.focal_y), float(camera.center_x), float(camera.center_y),
                int(camera.width), int(camera.height), float(camera.near_plane), float(camera.far_plane), _lib.ptr(mask), int(subtractive), st), 'occupancy_carve_view')
        _lib.check(lib.nrc_occupancy_carve_finish(_lib.ptr(remaining), m.cascades, g, _lib.ptr(m.occupancy_grid), st), 'occupancy_carve_finish')


def project_points(camera: Camera, c2w, xyz_world: torch.Tensor, z_culling: bool = True):
    """View.project_points (src/Datasets/utils.py:1040-1044): world_to_cam (:1027-1031, (x - position) @ R) then
    PerspectiveCamera.cam_to_screen (src/Cameras/Perspective.py:39-52, undistorted)."""
    c2w = torch.as_tensor(np.asarray(c2w, dtype=np.float64)[:4, :4], dtype=xyz_world.dtype, device=xyz_world.device)
    xyz_cam = (xyz_world - c2w[:3, 3]) @ c2w[:3, :3]
    depth = xyz_cam[:, 2]
    focals = torch.tensor((camera.focal_x, camera.focal_y), device=xyz_world.device, dtype=xyz_world.dtype)
    screen_size = torch.tensor((camera.width, camera.height), device=xyz_world.device, dtype=xyz_world.dtype)
    center = torch.tensor((camera.center_x, camera.center_y), device=xyz_world.device, dtype=xyz_world.dtype)
    xy_screen = xyz_cam[:, :2] / depth.clamp_min(1.0e-8)[:, None] * focals + center
    in_frustum = ((xy_screen >= 0) & (xy_screen < screen_size)).all(dim=-1)
    if z_culling:
        in_frustum &= (depth > camera.near_plane) & (depth < camera.far_plane)
    return xy_screen, depth, in_frustum
