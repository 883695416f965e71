"""nerficg_amd.nerf -- host-side mirror of the reference's vanilla NeRF hot path (config 1: nerf_lego.yaml, pure PyTorch, runs
on CPU or on the GPU through torch): FrequencyEncoding, NeRFBlock, stratified + PDF sampling, sample integration and the
hierarchical (coarse -> fine) ray renderer.

Reference: src/Methods/NeRF/utils.py:12-136, src/Methods/NeRF/Model.py:10-83, src/Methods/NeRF/Renderer.py:20-95.
Pinned against golden vectors generated from the reference (tests/test_host_golden.py).  `integrate_samples` is also the
independent statement of the compositing math that pins oracle/ngp_oracle.c (tests/test_oracle_ngp.py).
"""
from __future__ import annotations

import torch

__all__ = ['FrequencyEncoding', 'NeRFBlock', 'generate_samples', 'generate_samples_from_pdf', 'integrate_samples', 'render_rays']


class FrequencyEncoding(torch.nn.Module):
    """[x, cos(x 2^k), sin(x 2^k)] for k < K, cos block before sin block per input dimension, no pi factor (utils.py:12-36)."""

    def __init__(self, n_frequencies: int, append_input: bool) -> None:
        super().__init__()
        self.register_buffer('frequency_factors', torch.linspace(0.0, n_frequencies - 1.0, n_frequencies).exp2()[None, None, :])
        self.append_input = append_input

    def get_n_outputs(self, n_inputs: int) -> int:
        return n_inputs * 2 * self.frequency_factors.numel() + (n_inputs if self.append_input else 0)

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        f = inputs[..., None] * self.frequency_factors
        enc = torch.cat((torch.cos(f), torch.sin(f)), dim=-1).flatten(start_dim=1)
        return torch.cat([inputs, enc], dim=-1) if self.append_input else enc


class NeRFBlock(torch.nn.Module):
    """8 x (Linear 256 + ReLU) with the encoded position re-appended after layer 5, density head (Linear + ReLU), feature layer,
    colour head on [features, encoded direction] (Model.py:10-83).  Module / parameter names match the reference's state_dict."""

    def __init__(self, n_layers: int = 8, n_color_layers: int = 1, n_features: int = 256, n_frequencies_position: int = 10,
                 n_frequencies_direction: int = 4, encoding_append_input: bool = True, input_skips=(5,)) -> None:
        super().__init__()
        self.input_skips = list(input_skips)
        self.encoding_position = FrequencyEncoding(n_frequencies_position, encoding_append_input)
        self.encoding_direction = FrequencyEncoding(n_frequencies_direction, encoding_append_input)
        n_pos, n_dir = self.encoding_position.get_n_outputs(3), self.encoding_direction.get_n_outputs(3)
        layers = []
        for index in range(n_layers):
            n_in = n_pos if index == 0 else (n_features + n_pos if index in self.input_skips else n_features)
            layers.append(torch.nn.Sequential(torch.nn.Linear(n_in, n_features), torch.nn.ReLU(True)))
        self.initial_layers = torch.nn.ModuleList(layers)
        self.feature_layer = torch.nn.Linear(n_features, n_features)
        self.density_layer = torch.nn.Linear(n_features, 1)
        self.density_activation = torch.nn.ReLU(True)
        half = n_features // 2
        color = [torch.nn.Linear(n_features + n_dir, half), torch.nn.ReLU(True)]
        color += [torch.nn.Sequential(torch.nn.Linear(half, half), torch.nn.ReLU(True)) for _ in range(n_color_layers - 1)]
        color += [torch.nn.Linear(half, 3), torch.nn.Sigmoid()]
        self.color_layers = torch.nn.Sequential(*color)

    def forward(self, positions: torch.Tensor, directions: torch.Tensor, random_noise_density: float = 0.0):
        pos_enc = self.encoding_position(positions)
        x = pos_enc
        for index, layer in enumerate(self.initial_layers):
            x = layer(x)
            if index + 1 in self.input_skips:
                x = torch.cat((x, pos_enc), dim=-1)
        density = self.density_layer(x)
        if random_noise_density > 0.0:
            density = density + random_noise_density * torch.randn_like(density)
        density = self.density_activation(density)
        features = torch.cat((self.feature_layer(x), self.encoding_direction(directions)), dim=-1)
        return density, self.color_layers(features)


def generate_samples(n_rays: int, n_samples: int, near_plane: float, far_plane: float, randomize_samples: bool, dtype=torch.float32,
                     device=None) -> torch.Tensor:
    """utils.py:57-75: linspace(near, far) per ray, optionally jittered inside the bins spanned by the mid points."""
    depth = torch.linspace(near_plane, far_plane, n_samples, dtype=dtype, device=device).expand(n_rays, n_samples)
    if randomize_samples:
        mid = 0.5 * (depth[..., 1:] + depth[..., :-1])
        upper = torch.cat((mid, depth[..., -1:]), dim=-1)
        lower = torch.cat((depth[..., :1], mid), dim=-1)
        depth = lower + (upper - lower) * torch.rand(depth.shape, dtype=dtype, device=device)
    return depth


def generate_samples_from_pdf(bins: torch.Tensor, values: torch.Tensor, n_samples: int, randomize_samples: bool) -> torch.Tensor:
    """utils.py:78-109: inverse-CDF sampling of the piecewise-constant pdf given by the inner blending weights."""
    bins = 0.5 * (bins[..., :-1] + bins[..., 1:])
    values = values[..., 1:-1] + 1e-5
    pdf = values / torch.sum(values, dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, dim=-1)
    cdf = torch.cat((torch.zeros_like(cdf[..., :1]), cdf), dim=-1)
    if randomize_samples:
        u = torch.rand(*cdf.shape[:-1], n_samples, device=bins.device)
    else:
        u = torch.linspace(0.0, 1.0, steps=n_samples, device=bins.device).expand(*cdf.shape[:-1], n_samples)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below, above = (inds - 1).clamp_min(0), inds.clamp_max(cdf.shape[-1] - 1)
    cdf_lo, cdf_hi = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_lo, bin_hi = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, 1.0, denom)
    return (bin_lo + (u - cdf_lo) / denom * (bin_hi - bin_lo)).detach()


def integrate_samples(depth_samples, ray_directions, densities, colors, background_color, final_delta: float = 1.0e10):
    """utils.py:112-136: alpha = 1 - exp(-sigma delta), T = exclusive cumprod(1 - alpha), w = alpha T; rgb = sum w c (+ T_N bg),
    depth = sum w t / alpha_final where T_N < 1 else 0."""
    deltas = depth_samples[:, 1:] - depth_samples[..., :-1]
    last = torch.full_like(deltas[..., :1], final_delta)
    deltas = torch.cat((deltas, last), dim=-1) * ray_directions.norm(dim=-1, keepdim=True)
    alphas = 1.0 - torch.exp(-densities * deltas)
    transmittance = torch.cumprod(torch.cat((torch.ones_like(alphas[..., :1]), 1.0 - alphas), dim=-1), dim=-1)
    weights = alphas * transmittance[..., :-1]
    t_final = transmittance[..., -1:]
    alpha_final = 1.0 - t_final
    depth = torch.where(t_final < 1.0, torch.sum(weights * depth_samples, dim=-1, keepdim=True) / alpha_final, 0.0)
    rgb = torch.sum(weights[..., None] * colors, dim=-2)
    if background_color is not None:
        rgb = rgb + t_final * background_color
    return rgb, depth, alpha_final, weights


def render_rays(coarse_nerf: NeRFBlock | None, nerf: NeRFBlock, origin, direction, view_direction, near_plane: float, far_plane: float,
                background_color, ray_batch_size: int = 8192, n_samples_coarse_nerf: int = 64, n_samples_nerf: int = 192,
                randomize_samples: bool = False, random_noise_density: float = 0.0) -> dict[str, torch.Tensor]:
    """NeRFRayRenderingComponent.forward (Renderer.py:29-95): chunked coarse pass -> PDF samples -> sorted union -> fine pass."""
    use_coarse = coarse_nerf is not None and n_samples_coarse_nerf > 0
    keys = ['rgb', 'alpha', 'depth'] + (['rgb_coarse', 'alpha_coarse', 'depth_coarse'] if use_coarse else [])
    chunks: dict[str, list] = {k: [] for k in keys}
    bg = background_color.to(origin.device)
    for a in range(0, origin.shape[0], ray_batch_size):
        o, d, vd = origin[a:a + ray_batch_size], direction[a:a + ray_batch_size], view_direction[a:a + ray_batch_size]
        n = o.shape[0]
        if use_coarse:
            t_c = generate_samples(n, n_samples_coarse_nerf, near_plane, far_plane, randomize_samples, o.dtype, o.device)
            pos = o[:, None, :] + d[:, None, :] * t_c[:, :, None]
            dens, col = coarse_nerf(pos.reshape(-1, 3), vd[:, None, :].expand_as(pos).reshape(-1, 3), random_noise_density)
            rgb_c, depth_c, alpha_c, w_c = integrate_samples(t_c, d, dens.reshape(n, -1), col.reshape(n, -1, 3), bg)
            t_f = generate_samples_from_pdf(t_c, w_c, n_samples_nerf, randomize_samples)
            t, _ = torch.sort(torch.cat((t_c, t_f), dim=-1), dim=-1)
            chunks['rgb_coarse'].append(rgb_c); chunks['depth_coarse'].append(depth_c); chunks['alpha_coarse'].append(alpha_c)
        else:
            t = generate_samples(n, n_samples_nerf, near_plane, far_plane, randomize_samples, o.dtype, o.device)
        pos = o[:, None, :] + d[:, None, :] * t[:, :, None]
        dens, col = nerf(pos.reshape(-1, 3), vd[:, None, :].expand_as(pos).reshape(-1, 3), random_noise_density)
        rgb, depth, alpha, _ = integrate_samples(t, d, dens.reshape(n, -1), col.reshape(n, -1, 3), bg)
        chunks['rgb'].append(rgb); chunks['depth'].append(depth); chunks['alpha'].append(alpha)
    return {k: torch.cat(v, dim=0) for k, v in chunks.items()}
