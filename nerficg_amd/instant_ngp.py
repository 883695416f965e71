"""nerficg_amd.instant_ngp -- host-side mirror of the reference's InstantNGP method for the hot path
(src/Methods/InstantNGP/Model.py, src/Methods/InstantNGP/Renderer.py), written against nerficg_amd's drop-in modules.

Same class / method / attribute names and argument meaning as the reference, minus its Framework plumbing (config
objects, View/RayBatch dataclasses): rays are plain (N,3) tensors, cameras are the small `Camera` record below.  The
reference's own method package keeps working unchanged when its native imports are redirected; this
mirror exists so that parity tests and bench.py can drive the identical call sequence on a box without the reference.

On top of the mirrored op-by-op path, `InstantNGPRenderer.render_image_fused` is the MI355X-native restructuring of
render_image (no ray tensors, no per-iteration host syncs) -- it must produce the same image (tests/test_gpu_render_parity.py).
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass, field

import numpy as np
import torch

from . import VolumeRenderingV2 as VolumeRenderingCuda
from . import _lib
from . import tinycudann as tcnn
from .raygen import generate_rays


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
        self.center = torch.tensor([self.CENTER], dtype=torch.float32, device=dev)
        self.xyz_min = -torch.ones(1, 3, device=dev) * SCALE
        self.xyz_max = torch.ones(1, 3, device=dev) * SCALE
        self.xyz_size = self.xyz_max - self.xyz_min
        self.half_size = self.xyz_size / 2
        self.cascades = max(1 + int(math.ceil(math.log2(2 * SCALE))), 1)
        g = torch.arange(RESOLUTION, dtype=torch.int32, device=dev)
        self.grid_coords = torch.stack(torch.meshgrid([g, g, g], indexing='xy'), dim=-1).reshape(-1, 3).contiguous()
        self.register_buffer('occupancy_grid', torch.zeros(self.cascades, RESOLUTION ** 3, device=dev))
        self.register_buffer('occupancy_bitfield', torch.zeros(self.cascades * RESOLUTION ** 3 // 8, dtype=torch.uint8, device=dev))
        self.encoding_xyz = tcnn.NetworkWithInputEncoding(
            n_input_dims=3, n_output_dims=N_DENSITY_OUTPUT_FEATURES,
            encoding_config={
                'otype': 'Grid', 'type': 'Hash', 'n_levels': HASHGRID_N_LEVELS, 'n_features_per_level': HASHGRID_N_FEATURES_PER_LEVEL,
                'log2_hashmap_size': HASHGRID_LOG2_SIZE, 'base_resolution': HASHGRID_BASE_RESOLUTION,
                'per_level_scale': math.exp(math.log(HASHGRID_TARGET_RESOLUTION * (2 * SCALE) / HASHGRID_BASE_RESOLUTION) / (HASHGRID_N_LEVELS - 1)),
                'interpolation': 'Linear'},
            network_config={'otype': 'FullyFusedMLP', 'activation': 'ReLU', 'output_activation': 'None',
                            'n_neurons': N_DENSITY_NEURONS, 'n_hidden_layers': N_DENSITY_LAYERS},
            seed=RANDOM_SEED).to(dev)
        n_params_mlp = 0
        n_inputs = next_multiple(HASHGRID_N_FEATURES_PER_LEVEL * HASHGRID_N_LEVELS, 16)
        for _ in range(N_DENSITY_LAYERS):
            n_params_mlp += next_multiple(N_DENSITY_NEURONS * n_inputs, 16)
            n_inputs = N_DENSITY_NEURONS
        n_params_mlp += next_multiple(self.encoding_xyz.n_output_dims, 16) * n_inputs
        self.n_params_encoding_mlp = n_params_mlp
        self.color_mlp_with_encoding = tcnn.NetworkWithInputEncoding(
            n_input_dims=3 + self.encoding_xyz.n_output_dims, n_output_dims=3,
            encoding_config={'otype': 'Composite', 'nested': [
                {'n_dims_to_encode': 3, 'otype': 'SphericalHarmonics', 'degree': DIR_SH_ENCODING_DEGREE}, {'otype': 'Identity'}]},
            network_config={'otype': 'FullyFusedMLP', 'activation': 'ReLU', 'output_activation': 'Sigmoid',
                            'n_neurons': N_COLOR_NEURONS, 'n_hidden_layers': N_COLOR_LAYERS},
            seed=RANDOM_SEED).to(dev)
        self.n_mlp_params = len(self.color_mlp_with_encoding.params) + self.n_params_encoding_mlp

    def weight_decay_mlp(self) -> torch.Tensor:
        """Model.py:38-44"""
        loss = self.encoding_xyz.params[:self.n_params_encoding_mlp].pow(2).sum()
        loss = loss + self.color_mlp_with_encoding.params.pow(2).sum()
        return loss / self.n_mlp_params


class InstantNGPRayRenderingComponent(torch.nn.Module):
    """src/Methods/InstantNGP/Renderer.py:19-138"""

    def __init__(self, model: InstantNGPModel) -> None:
        super().__init__()
        self.model = model
        self.fused_training_query = True
        self._box_host = None

    def forward(self, origin: torch.Tensor, view_direction: torch.Tensor, camera: Camera, max_samples: int, bg_color: torch.Tensor,
                exponential_steps: bool, train_mode: bool) -> dict[str, torch.Tensor]:
        rays_o = origin - self.model.center
        rays_d = view_direction.contiguous()
        hits_t = VolumeRenderingCuda.RayAABBIntersector.apply(rays_o, rays_d, torch.zeros((1, 3), device=rays_o.device), self.model.half_size, 1)[1]
        hits_t[..., 0].clamp_min_(camera.near_plane)
        hits_t[..., 1].clamp_max_(camera.far_plane)
        exp_step_factor = 1 / 256 if exponential_steps else 0.0
        render_fn = self.render_rays_training if train_mode else self.render_rays_inference
        return render_fn(rays_o, rays_d, hits_t, max_samples, bg_color, exp_step_factor)

    def query_model(self, x: torch.Tensor, d: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        m = self.model
        if self.fused_training_query and torch.is_grad_enabled() and m.encoding_xyz.params.requires_grad and x.dtype == torch.float32:
            # one autograd node for the whole statement sequence below (same arithmetic; nerficg_amd.ngp.query_train)
            from .ngp import query_train
            if self._box_host is None:
                self._box_host = (m.xyz_min.detach().float().cpu().contiguous(), m.xyz_size.detach().float().cpu().contiguous())
            return query_train(m.encoding_xyz, m.color_mlp_with_encoding, x, d, *self._box_host)
        h = self.model.encoding_xyz((x - self.model.xyz_min) / self.model.xyz_size)
        sigmas = VolumeRenderingCuda.TruncExp.apply(h[:, 0])
        rgbs = self.model.color_mlp_with_encoding(torch.cat([(d * 0.5 + 0.5).to(h.dtype), h], dim=-1))
        return sigmas, rgbs

    def query_density(self, x: torch.Tensor) -> torch.Tensor:
        h = self.model.encoding_xyz((x - self.model.xyz_min) / self.model.xyz_size)
        return VolumeRenderingCuda.TruncExp.apply(h[:, 0])

    @torch.amp.autocast('cuda')
    def render_rays_training(self, rays_o, rays_d, hits_t, max_samples, bg_color, exp_step_factor) -> dict[str, torch.Tensor]:
        rays_a, xyzs, dirs, deltas, ts, rm_samples = VolumeRenderingCuda.RayMarcher.apply(
            rays_o, rays_d, hits_t[:, 0], self.model.occupancy_bitfield, self.model.cascades, self.model.SCALE, exp_step_factor,
            self.model.RESOLUTION, max_samples)
        sigmas, rgbs = self.query_model(xyzs, dirs)
        vr_samples, alpha, depth, rgb, ws = VolumeRenderingCuda.VolumeRenderer.apply(sigmas, rgbs.contiguous(), deltas, ts, rays_a, 1e-4)
        rgb = rgb + bg_color * (1 - alpha[:, None])
        depth = depth / (alpha + 1e-6)
        return {'rgb': rgb, 'alpha': alpha, 'depth': depth, 'rm_samples': rm_samples}

    @torch.no_grad()
    def render_rays_inference(self, rays_o, rays_d, hits_t, max_samples, bg_color, exp_step_factor, fused_query: bool = False) -> dict[str, torch.Tensor]:
        """Renderer.py:86-138, statement for statement.  `fused_query` swaps query_model for the single-kernel equivalent."""
        n_rays = len(rays_o)
        device = rays_o.device
        rgb = torch.zeros(n_rays, 3, device=device)
        alpha = torch.zeros(n_rays, device=device)
        depth = torch.zeros(n_rays, device=device)
        alive_indices = torch.arange(n_rays, device=device)
        min_samples = 1 if exp_step_factor == 0 else 4
        samples = 0
        hits = hits_t[:, 0]
        while samples < max_samples:
            n_alive = len(alive_indices)
            if n_alive == 0:
                break
            n_samples = max(min(n_rays // n_alive, 64), min_samples)
            samples += n_samples
            xyzs, dirs, deltas, ts, n_eff_samples = VolumeRenderingCuda.raymarching_test(
                rays_o, rays_d, hits, alive_indices, self.model.occupancy_bitfield, self.model.cascades, self.model.SCALE,
                exp_step_factor, self.model.RESOLUTION, max_samples, n_samples)
            xyzs = xyzs.reshape(-1, 3)
            dirs = dirs.reshape(-1, 3)
            valid_mask = (dirs != 0).any(dim=1)
            if valid_mask.sum() == 0:
                break
            sigmas = torch.zeros(len(xyzs), device=device)
            rgbs = torch.zeros(len(xyzs), 3, device=device)
            if fused_query:
                from .ngp import query_fused
                x01 = ((xyzs[valid_mask] - self.model.xyz_min) / self.model.xyz_size).contiguous()
                _sigmas, _rgbs = query_fused(self.model.encoding_xyz, self.model.color_mlp_with_encoding, x01, dirs[valid_mask].contiguous())
            else:
                with torch.amp.autocast('cuda'):
                    _sigmas, _rgbs = self.query_model(xyzs[valid_mask], dirs[valid_mask])
            sigmas[valid_mask], rgbs[valid_mask] = _sigmas.float(), _rgbs.float()
            sigmas = sigmas.reshape(-1, n_samples)
            rgbs = rgbs.reshape(-1, n_samples, 3)
            VolumeRenderingCuda.composite_test_fw(sigmas, rgbs, deltas, ts, hits, alive_indices, 1e-4, n_eff_samples, alpha, depth, rgb)
            alive_indices = alive_indices[alive_indices >= 0]
        alpha.clamp_(0, 1)
        transmittance = 1 - alpha
        rgb += transmittance[:, None] * bg_color
        rgb.clamp_(0, 1)
        depth = torch.where(transmittance < 1.0, depth / alpha, 0.0)
        return {'rgb': rgb, 'alpha': alpha, 'depth': depth}


class InstantNGPRenderer:
    """src/Methods/InstantNGP/Renderer.py:141-272 (MAX_SAMPLES / EXPONENTIAL_STEPS / DENSITY_THRESHOLD defaults :141-145)."""

    def __init__(self, model: InstantNGPModel, MAX_SAMPLES: int = 1024, EXPONENTIAL_STEPS: bool = False, DENSITY_THRESHOLD: float = 0.01) -> None:
        self.model = model
        self.MAX_SAMPLES, self.EXPONENTIAL_STEPS, self.DENSITY_THRESHOLD = MAX_SAMPLES, EXPONENTIAL_STEPS, DENSITY_THRESHOLD
        self.ray_rendering_component = InstantNGPRayRenderingComponent(model)
        self.density_threshold = DENSITY_THRESHOLD * MAX_SAMPLES / 3 ** 0.5
        self._fused_ws: dict = {}
        # fused image path: the count pass parks the samples in a max_samples-row arena per tile (2.6 GB at 800x800) and the write pass
        # copies them instead of marching every ray twice; False = second march, no arena
        self.provisional_march = True

    def render_rays(self, origin, view_direction, camera: Camera, train_mode: bool = False, custom_bg_color: torch.Tensor | None = None):
        bg = custom_bg_color if custom_bg_color is not None else camera.background_color.to(origin.device)
        return self.ray_rendering_component(origin, view_direction, camera, self.MAX_SAMPLES, bg, self.EXPONENTIAL_STEPS, train_mode)

    def render_image(self, camera: Camera, c2w: np.ndarray, to_chw: bool = False) -> dict[str, torch.Tensor]:
        """Renderer.py:172-180: view.get_rays() -> render_rays -> reshape to images."""
        rays = generate_rays(camera.width, camera.height, camera.focal_x, camera.focal_y, camera.center_x, camera.center_y, c2w,
                             device=self.model.center.device, want_direction=False)
        out = self.render_rays(rays['origin'], rays['view_direction'], camera)
        for key in out:
            out[key] = out[key].reshape(camera.height, camera.width, -1)
            if to_chw:
                out[key] = out[key].permute(2, 0, 1)
        return out

    # ---------------------------------------------------------------- MI355X-native image pipeline
    @staticmethod
    def n_image_tiles(camera: Camera) -> int:
        lib = _lib.load()
        tw, th = int(lib.nrc_ngp_tile_width()), int(lib.nrc_ngp_tile_height())
        return ((camera.width + tw - 1) // tw) * ((camera.height + th - 1) // th)

    @torch.no_grad()
    def render_image_fused(self, camera: Camera, c2w: np.ndarray, tile_begin: int = 0, n_tiles: int | None = None,
                           return_stats: bool = False, out: dict | None = None, early_termination: bool | str = 'auto') -> dict[str, torch.Tensor]:
        """Same image as render_image, flat (H*W, C) pixel-major buffers, through the tile-interleaved device pipeline
        (include/nerficg_hip.h group 6): four device stages and ONE host sync (the row count, to size the sample buffers).
        A shard renders the 8x8-pixel tiles [tile_begin, tile_begin + n_tiles) and writes only their pixels (pass `out` to
        accumulate several shards into the same buffers).  early_termination: rows in layer order, depth slabs composited front to
        back, finished tiles skipped in the following slabs (Renderer.py:118-132 at slab granularity); False: one pass over all
        samples.  Both give the same image.  'auto' (default): slabs as long as they pay -- when a slab frame saved less than 10 % of
        its rows (e.g. an untrained model: no ray ever saturates, and the slab bookkeeping costs ~7 %) the next 32 frames of this
        renderer take the single pass, then one slab frame probes again."""
        m = self.model
        lib = _lib.load()
        dev = m.center.device
        total_tiles = self.n_image_tiles(camera)
        nt = total_tiles - tile_begin if n_tiles is None else int(n_tiles)
        n = nt * 64
        c2w = np.ascontiguousarray(np.asarray(c2w, dtype=np.float64))
        if c2w.shape == (3, 4):
            c2w = np.vstack([c2w, [0.0, 0.0, 0.0, 1.0]])
        intr = (ctypes.c_double * 4)(camera.focal_x, camera.focal_y, camera.center_x, camera.center_y)
        mat = (ctypes.c_double * 16)(*c2w.reshape(-1).tolist())
        f3 = lambda t: (ctypes.c_float * 3)(*[float(v) for v in t.reshape(-1).tolist()])
        hw = camera.width * camera.height
        key = (nt, hw, str(dev))
        ws = self._fused_ws.get(key)
        if ws is None:
            ws = dict(ray_od=torch.empty(max(n, 1), 6, device=dev), ray_t=torch.empty(max(n, 1), 2, device=dev),
                      ray_cnt=torch.empty(max(n, 1), dtype=torch.int32, device=dev), tile_rows=torch.empty(max(nt, 1), dtype=torch.int32, device=dev),
                      tile_off=torch.empty(nt + 1, dtype=torch.int32, device=dev), counter=torch.empty(2, dtype=torch.int32, device=dev),
                      rgb=torch.zeros(hw, 3, device=dev), alpha=torch.zeros(hw, device=dev), depth=torch.zeros(hw, device=dev), cap=0,
                      skipped=torch.zeros(1, dtype=torch.int32, device=dev))
            if self.provisional_march:  # samples parked by the count pass, copied (not re-marched) by the write pass
                ws['ts_prov'] = torch.empty(int(lib.nrc_ngp_render_provisional_bytes(nt, self.MAX_SAMPLES)), dtype=torch.uint8, device=dev)
            self._fused_ws = {key: ws}
        if out is None:
            out = {'rgb': ws['rgb'], 'alpha': ws['alpha'], 'depth': ws['depth']}
        st = _lib.stream_of(ws['ray_od'])
        esf = 1 / 256 if self.EXPONENTIAL_STEPS else 0.0
        center, half = f3(m.center), f3(m.half_size)
        _lib.check(lib.nrc_ngp_render_count(
            camera.width, camera.height, ctypes.cast(intr, ctypes.c_void_p), ctypes.cast(mat, ctypes.c_void_p),
            ctypes.cast(center, ctypes.c_void_p), ctypes.cast(half, ctypes.c_void_p), float(camera.near_plane), float(camera.far_plane),
            int(tile_begin), nt, _lib.ptr(m.occupancy_bitfield), m.cascades, float(m.SCALE), float(esf), m.RESOLUTION, self.MAX_SAMPLES,
            _lib.ptr(ws['ray_od']), _lib.ptr(ws['ray_t']), _lib.ptr(ws['ray_cnt']), _lib.ptr(ws['tile_rows']), _lib.ptr(ws['tile_off']),
            _lib.ptr(ws['counter']), _lib.ptr(ws.get('ts_prov')), st), 'ngp_render_count')
        rows = int(ws['counter'][0].item())
        if rows > ws['cap']:
            cap = int(rows * 1.5) + 64  # poses of one scene differ by up to 30 % in rows: grow rarely (a regrowth costs milliseconds of hipMalloc)
            ws.update(ts=torch.empty(cap * 64, device=dev), row_tile=torch.empty(cap, dtype=torch.int32, device=dev),
                      packed=torch.empty(cap * 64, 4, dtype=torch.float16, device=dev),
                      qws=torch.empty(int(lib.nrc_ngp_render_layers_ws_bytes(cap, nt)), dtype=torch.uint8, device=dev),
                      row_of=torch.empty(cap, dtype=torch.int32, device=dev),
                      layer_off=torch.empty(self.MAX_SAMPLES + 2, dtype=torch.int32, device=dev), cap=cap)
        bg = f3(camera.background_color.float().cpu())
        if early_termination == 'auto':
            pol = ws.setdefault('et_policy', {'prev_rows': 0, 'prev_layered': False, 'skip_frames': 0})
            if pol['prev_layered'] and pol['prev_rows'] > 0:  # the previous frame is complete by now (we just synchronised on this one's row count)
                if int(ws['skipped'].item()) < 0.1 * pol['prev_rows']:
                    pol['skip_frames'] = 32
            use_layers = pol['skip_frames'] == 0
            pol['skip_frames'] = max(0, pol['skip_frames'] - 1)
            pol['prev_rows'], pol['prev_layered'] = rows, use_layers
        else:
            use_layers = bool(early_termination)
        if use_layers and self.MAX_SAMPLES <= 1024:
            g = m.encoding_xyz.grid_cfg
            mn, sz = f3(m.xyz_min), f3(m.xyz_size)
            if rows > 0:
                _lib.check(lib.nrc_ngp_render_write_layers(
                    nt, _lib.ptr(m.occupancy_bitfield), m.cascades, float(m.SCALE), float(esf), m.RESOLUTION, self.MAX_SAMPLES, _lib.ptr(ws['ray_od']),
                    _lib.ptr(ws['ray_t']), _lib.ptr(ws['ray_cnt']), _lib.ptr(ws['tile_rows']), _lib.ptr(ws['tile_off']), _lib.ptr(ws['ts']),
                    _lib.ptr(ws['row_tile']), _lib.ptr(ws['layer_off']), _lib.ptr(ws['row_of']), _lib.ptr(ws.get('ts_prov')), st), 'ngp_render_write_layers')
            if 'qws' not in ws:  # an image without a single sample: state + background only
                ws['qws'] = torch.empty(int(lib.nrc_ngp_render_layers_ws_bytes(0, nt)), dtype=torch.uint8, device=dev)
            _lib.check(lib.nrc_ngp_render_layers(
                _lib.ptr(ws.get('ts')), _lib.ptr(ws.get('row_tile')), _lib.ptr(ws['ray_od']), rows, nt, ctypes.cast(mn, ctypes.c_void_p),
                ctypes.cast(sz, ctypes.c_void_p), _lib.ptr(m.encoding_xyz._half_params()), _lib.ptr(m.color_mlp_with_encoding._half_params()),
                _lib.ptr(m.encoding_xyz._table16()), g['n_levels'], g['log2_hashmap_size'], g['base_resolution'], float(g['per_level_scale']),
                _lib.ptr(ws['ray_cnt']), _lib.ptr(ws['tile_rows']), _lib.ptr(ws['tile_off']), _lib.ptr(ws.get('row_of')), camera.width, camera.height,
                int(tile_begin), m.cascades, float(esf), m.RESOLUTION, self.MAX_SAMPLES, 1e-4, ctypes.cast(bg, ctypes.c_void_p),
                _lib.ptr(ws.get('packed')), _lib.ptr(out['rgb']), _lib.ptr(out['alpha']), _lib.ptr(out['depth']), _lib.ptr(ws['skipped']),
                _lib.ptr(ws['qws']), st), 'ngp_render_layers')
            res = dict(out)
            if return_stats:
                res['n_rows'] = rows
                res['n_slots'] = rows * 64
                res['n_samples'] = int(ws['ray_cnt'][:n].sum().item()) if n > 0 else 0
            return res
        if rows > 0:
            _lib.check(lib.nrc_ngp_render_write(nt, _lib.ptr(m.occupancy_bitfield), m.cascades, float(m.SCALE), float(esf), m.RESOLUTION,
                                                self.MAX_SAMPLES, _lib.ptr(ws['ray_od']), _lib.ptr(ws['ray_t']), _lib.ptr(ws['ray_cnt']),
                                                _lib.ptr(ws['tile_off']), _lib.ptr(ws['ts']), _lib.ptr(ws['row_tile']), _lib.ptr(ws.get('ts_prov')), st), 'ngp_render_write')
            g = m.encoding_xyz.grid_cfg
            mn, sz = f3(m.xyz_min), f3(m.xyz_size)
            _lib.check(lib.nrc_ngp_query_samples(
                _lib.ptr(ws['ts']), _lib.ptr(ws['row_tile']), _lib.ptr(ws['ray_od']), rows, nt, ctypes.cast(mn, ctypes.c_void_p),
                ctypes.cast(sz, ctypes.c_void_p), _lib.ptr(m.encoding_xyz._half_params()), _lib.ptr(m.color_mlp_with_encoding._half_params()),
                _lib.ptr(m.encoding_xyz._table16()), g['n_levels'], g['log2_hashmap_size'], g['base_resolution'], float(g['per_level_scale']),
                _lib.ptr(ws['packed']), _lib.ptr(ws['qws']), st), 'ngp_query_samples')
        _lib.check(lib.nrc_ngp_composite_image(
            _lib.ptr(ws.get('packed')), _lib.ptr(ws.get('ts')), _lib.ptr(ws['ray_cnt']), _lib.ptr(ws['tile_off']), camera.width, camera.height,
            int(tile_begin), nt, m.cascades, float(esf), m.RESOLUTION, self.MAX_SAMPLES, 1e-4, ctypes.cast(bg, ctypes.c_void_p),
            _lib.ptr(out['rgb']), _lib.ptr(out['alpha']), _lib.ptr(out['depth']), st), 'ngp_composite_image')
        res = dict(out)
        if return_stats:
            res['n_rows'] = rows
            res['n_slots'] = rows * 64
            res['n_samples'] = int(ws['ray_cnt'][:n].sum().item()) if n > 0 else 0
        return res

    # ---------------------------------------------------------------- occupancy grid (Renderer.py:183-206, 247-272)
    @torch.no_grad()
    def get_occupancy_grid_cells(self):
        indices = VolumeRenderingCuda.morton3D(self.model.grid_coords).long()
        return [(indices, self.model.grid_coords)] * self.model.cascades

    @torch.no_grad()
    def sample_occupancy_grid(self, n_samples: int, density_threshold: float):
        cells = []
        dev = self.model.occupancy_grid.device
        for c in range(self.model.cascades):
            coords1 = torch.randint(self.model.RESOLUTION, (n_samples, 3), dtype=torch.int32, device=dev)
            indices1 = VolumeRenderingCuda.morton3D(coords1).long()
            indices2 = torch.nonzero(self.model.occupancy_grid[c] > density_threshold)[:, 0]
            if len(indices2) > 0:
                rand_idx = torch.randint(len(indices2), (n_samples,), device=dev)
                indices2 = indices2[rand_idx]
            coords2 = VolumeRenderingCuda.morton3D_invert(indices2.int())
            cells += [(torch.cat([indices1, indices2]), torch.cat([coords1, coords2]))]
        return cells

    @torch.no_grad()
    @torch.amp.autocast('cuda')
    def update_occupancy_grid(self, warmup: bool = False, decay: float = 0.95) -> None:
        """Renderer.py:247-272.  Cell sampling and jitter draw from torch's generator in the reference's order; the densities of ALL
        cascades come from one network query, and scratch-grid scatter, EMA, masked mean, threshold and bit packing are one C-ABI call
        (include/nerficg_hip.h group 11) that leaves the threshold on the device (`self.occupancy_threshold`, 2 floats: used, mean) --
        the reference's `.item()` host sync is gone."""
        m = self.model
        cells = self.get_occupancy_grid_cells() if warmup else self.sample_occupancy_grid(m.RESOLUTION ** 3 // 4, self.density_threshold)
        points = []
        for c in range(m.cascades):
            _, coords = cells[c]
            s = min(2 ** (c - 1), m.SCALE)
            half_grid_size = s / m.RESOLUTION
            xyzs_w = (coords / (m.RESOLUTION - 1) * 2 - 1) * (s - half_grid_size)
            xyzs_w += (torch.rand_like(xyzs_w) * 2 - 1) * half_grid_size
            points.append(xyzs_w)
        per_cascade = max(p.shape[0] for p in points)
        density = self.ray_rendering_component.query_density(torch.cat(points) if len(points) > 1 else points[0]).reshape(-1)
        if density.dtype not in (torch.float32, torch.float16):
            density = density.float()
        dev = m.occupancy_grid.device
        if all(p.shape[0] == per_cascade for p in points):
            indices = torch.stack([cells[c][0] for c in range(m.cascades)]).contiguous()
            density = density.contiguous()
        else:  # a cascade without occupied cells draws fewer samples: pad with ignored entries (index -1)
            indices = torch.full((m.cascades, per_cascade), -1, dtype=torch.int64, device=dev)
            padded = torch.zeros((m.cascades, per_cascade), dtype=density.dtype, device=dev)
            o = 0
            for c, p in enumerate(points):
                indices[c, :p.shape[0]] = cells[c][0]
                padded[c, :p.shape[0]] = density[o:o + p.shape[0]]
                o += p.shape[0]
            density = padded
        lib = _lib.load()
        grid = m.occupancy_grid
        _lib.check_input(grid, 'occupancy_grid', torch.float32)
        n_cells = grid.numel()
        ws_bytes = int(lib.nrc_occupancy_update_ws_bytes(n_cells))
        if getattr(self, '_occ_ws', None) is None or self._occ_ws.numel() < ws_bytes:
            self._occ_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            self.occupancy_threshold = torch.zeros(2, dtype=torch.float32, device=dev)
        _lib.check(lib.nrc_occupancy_update(_lib.ptr(grid), _lib.ptr(indices), _lib.ptr(density), 0 if density.dtype == torch.float32 else 1, m.cascades,
                                            grid.shape[1], per_cascade, decay, self.density_threshold, _lib.ptr(m.occupancy_bitfield),
                                            _lib.ptr(self.occupancy_threshold), _lib.ptr(self._occ_ws), _lib.stream_of(grid)), 'occupancy_update')

    @torch.no_grad()
    def carve_occupancy_grid(self, views, subtractive: bool = False, use_alpha: bool = False) -> None:
        """Renderer.py:208-245.  `views`: iterable of (Camera, c2w (4,4), alpha (1,H,W) tensor or None) -- the fields of a dataset View the
        reference reads.  Cells outside every frustum (subtractive=False) / outside any frustum (True) get -1 (never sampled again by
        update_occupancy_grid), the others 0; the kept set is dilated by one cell (3x3x3)."""
        m = self.model
        dev = m.occupancy_grid.device
        cells = self.get_occupancy_grid_cells()
        cell_positions_world = []
        for c in range(m.cascades):
            _, coords = cells[c]
            s = min(2 ** (c - 1), m.SCALE)
            half_grid_size = s / m.RESOLUTION
            cell_positions_world.append((coords / (m.RESOLUTION - 1) * 2 - 1) * (s - half_grid_size) + m.center)
        remaining_cells = torch.full_like(m.occupancy_grid, fill_value=subtractive, dtype=torch.bool, device=dev)
        dilation_kernel_2d = torch.ones(1, 1, 3, 3, device=dev)
        for camera, c2w, alpha_gt in views:
            if use_alpha and alpha_gt is not None:
                alpha_gt = torch.nn.functional.conv2d(alpha_gt.to(dev)[None], dilation_kernel_2d, padding=1)[0] > 0.0
            for c in range(m.cascades):
                xy_screen, _, in_frustum = project_points(camera, c2w, cell_positions_world[c])
                if use_alpha and alpha_gt is not None:
                    xy_screen = torch.floor(xy_screen[in_frustum]).long()
                    alpha_values = alpha_gt[:, xy_screen[:, 1], xy_screen[:, 0]] > 0.0
                    in_frustum[in_frustum.clone()] = alpha_values[0]
                remaining_cells[c] = remaining_cells[c] & in_frustum if subtractive else remaining_cells[c] | in_frustum
        dilation_kernel_3d = torch.ones(1, 1, 3, 3, 3, device=dev)
        for c in range(m.cascades):
            # `remaining_cells[c]` is indexed like grid_coords (x-major meshgrid order), exactly as in the reference
            dilated = torch.nn.functional.conv3d(remaining_cells[c].reshape(1, 1, m.RESOLUTION, m.RESOLUTION, m.RESOLUTION).float(),
                                                 dilation_kernel_3d, padding=1)
            values = torch.where(dilated.flatten() > 0.0, 0.0, -1.0)
            m.occupancy_grid[c, cells[c][0]] = values


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
