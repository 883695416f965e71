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
        nn = torch.nn
        widths_in = [n_pos] + [n_features + (n_pos if k in self.input_skips else 0) for k in range(1, n_layers)]
        self.initial_layers = nn.ModuleList(nn.Sequential(nn.Linear(w, n_features), nn.ReLU(True)) for w in widths_in)
        self.feature_layer = torch.nn.Linear(n_features, n_features)
        self.density_layer = torch.nn.Linear(n_features, 1)
        self.density_activation = torch.nn.ReLU(True)
        half = n_features // 2
        color = [torch.nn.Linear(n_features + n_dir, half), torch.nn.ReLU(True)]
        color += [torch.nn.Sequential(torch.nn.Linear(half, half), torch.nn.ReLU(True)) for _ in range(n_color_layers - 1)]
        color += [torch.nn.Linear(half, 3), torch.nn.Sigmoid()]
        self.color_layers = torch.nn.Sequential(*color)

    def forward(self, positions: torch.Tensor, directions: torch.Tensor, random_noise_density: float = 0.0):
        encoded = self.encoding_position(positions)
        hidden = encoded
        for depth_index, block in enumerate(self.initial_layers, start=1):
            hidden = block(hidden)
            if depth_index in self.input_skips:  # the encoded position re-enters behind this layer
                hidden = torch.cat((hidden, encoded), dim=-1)
        sigma = self.density_layer(hidden)
        if random_noise_density > 0.0:
            sigma = sigma + torch.randn_like(sigma) * random_noise_density
        view_dependent = torch.cat((self.feature_layer(hidden), self.encoding_direction(directions)), dim=-1)
        return self.density_activation(sigma), self.color_layers(view_dependent)


def generate_samples(n_rays: int, n_samples: int, near_plane: float, far_plane: float, randomize_samples: bool, dtype=torch.float32,
                     device=None) -> torch.Tensor:
    """Depths of the coarse pass (utils.py:57-75): `n_samples` evenly spaced values in [near, far] for every ray; with `randomize_samples` each
    one is redrawn uniformly inside its stratum (the strata meet at the mid points, the two outer ones end at near / far)."""
    t = torch.linspace(near_plane, far_plane, n_samples, dtype=dtype, device=device).expand(n_rays, n_samples)
    if not randomize_samples:
        return t
    cuts = (t[:, 1:] + t[:, :-1]) * 0.5
    lower, upper = torch.cat((t[:, :1], cuts), dim=1), torch.cat((cuts, t[:, -1:]), dim=1)
    return torch.lerp(lower, upper, torch.rand(t.shape, dtype=dtype, device=device))


def generate_samples_from_pdf(bins: torch.Tensor, values: torch.Tensor, n_samples: int, randomize_samples: bool) -> torch.Tensor:
    """Depths of the fine pass (utils.py:78-109): inverse-transform sampling of the piecewise-constant density whose K-2 inner masses
    (`values` without its ends, + 1e-5) sit between the K-1 mid points of `bins`; u is uniform (random or evenly spaced)."""
    knots = (bins[:, 1:] + bins[:, :-1]) * 0.5
    mass = values[:, 1:-1] + 1e-5
    cdf = torch.nn.functional.pad(torch.cumsum(mass / mass.sum(dim=1, keepdim=True), dim=1), (1, 0))  # cdf[:, 0] = 0
    rows = cdf.shape[0]
    if randomize_samples:
        u = torch.rand(rows, n_samples, device=bins.device)
    else:
        u = torch.linspace(0.0, 1.0, steps=n_samples, device=bins.device).expand(rows, n_samples).contiguous()
    right = torch.searchsorted(cdf, u, right=True)
    left, right = (right - 1).clamp(min=0), right.clamp(max=cdf.shape[1] - 1)
    c0, c1, k0, k1 = cdf.gather(1, left), cdf.gather(1, right), knots.gather(1, left), knots.gather(1, right)
    width = c1 - c0
    width = torch.where(width < 1e-5, torch.ones_like(width), width)  # empty interval: take its left knot
    return (k0 + (u - c0) / width * (k1 - k0)).detach()


def integrate_samples(depth_samples, ray_directions, densities, colors, background_color, final_delta: float = 1.0e10):
    """Emission-absorption quadrature along each ray (utils.py:112-136).  Segment i spans depth i .. i+1 (the last one `final_delta`), scaled
    by |direction| because depths are in units of the unnormalised ray; opacity_i = 1 - exp(-sigma_i len_i); a sample is weighted by its
    opacity times the transmittance of everything in front.  -> rgb (+ leftover transmittance * background), depth (weighted mean, 0 for
    rays that hit nothing), alpha, weights."""
    pad = torch.nn.functional.pad
    length = pad(torch.diff(depth_samples, dim=-1), (0, 1), value=final_delta) * torch.linalg.vector_norm(ray_directions, dim=-1, keepdim=True)
    opacity = 1.0 - torch.exp(-densities * length)
    through = torch.cumprod(pad(1.0 - opacity, (1, 0), value=1.0), dim=-1)      # through[:, i] = transmittance in front of sample i; [:, -1] = leftover
    weights = opacity * through[:, :-1]
    leftover = through[:, -1:]
    alpha = 1.0 - leftover
    weighted_depth = (weights * depth_samples).sum(dim=-1, keepdim=True)
    depth = torch.where(leftover < 1.0, weighted_depth / alpha, torch.zeros_like(alpha))
    rgb = (weights.unsqueeze(-1) * colors).sum(dim=-2)
    if background_color is not None:
        rgb = rgb + leftover * background_color
    return rgb, depth, alpha, weights


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
