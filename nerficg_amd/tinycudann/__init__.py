"""nerficg_amd.tinycudann -- the subset of the tiny-cuda-nn PyTorch API that nerficg imports through
src/Thirdparty/TinyCudaNN.py (`from tinycudann import *`) and uses in src/Methods/InstantNGP/Model.py:36,40-41,58-120:

    NetworkWithInputEncoding(n_input_dims, n_output_dims, encoding_config, network_config, seed)
        .params (flat f32 nn.Parameter: MLP weights first, encoding table after), .n_output_dims, .jit_fusion
        __call__((M, n_input_dims) CUDA tensor) -> (M, n_output_dims) fp16
    free_temporary_memory(), supports_jit_fusion()

backed by hand-written gfx950 kernels (libnerficg_hip.so: nrc_nwie_forward / nrc_nwie_backward / nrc_grid_backward).
Supported configurations = the family src/Methods/InstantNGP/Model.py:18-29,58-114 can instantiate from its yaml keys (anything else raises, by key name):
  encoding  {'otype':'Grid','type':'Hash', interpolation 'Linear', n_levels L, n_features_per_level F in {2, 4} with L x F <= 32, any
             log2_hashmap_size / base_resolution / per_level_scale}   with n_input_dims == 3
             (16 x 2, the shipped yaml, runs on the tuned kernels -- every other (L, F) on the general ones: nrc_nwie_forward's encoding bits)
            {'otype':'Composite','nested':[{'n_dims_to_encode':3,'otype':'SphericalHarmonics','degree': 1..4},
             {'otype':'Identity'}]}                                    with n_input_dims == 19
  network   {'otype':'FullyFusedMLP','activation':'ReLU','output_activation':'None'|'Sigmoid','n_neurons':64,
             'n_hidden_layers': 1|2}                                   with n_output_dims <= 16
"""
from __future__ import annotations

import ctypes
import math
import weakref

import torch

from .. import _lib

__all__ = ['NetworkWithInputEncoding', 'free_temporary_memory', 'supports_jit_fusion']

_WIDTH = 64
_PAD = 16
LOSS_SCALE = 128.0  # tiny-cuda-nn's internal loss scale for fp16 backward activations


def free_temporary_memory() -> None:
    """tiny-cuda-nn frees its own arena here; this build allocates through torch's caching allocator."""
    return None


def supports_jit_fusion() -> bool:
    """Encoding + MLP are always one fused kernel in this build (the flag `jit_fusion` is accepted and has no effect)."""
    return True


def _grid_offsets(cfg: dict) -> list[int]:
    offs = (ctypes.c_uint32 * (cfg['n_levels'] + 1))()
    _lib.check(_lib.load().nrc_grid_layout(cfg['n_levels'], cfg['log2_hashmap_size'], cfg['base_resolution'],
                                           float(cfg['per_level_scale']), ctypes.cast(offs, ctypes.c_void_p)), 'grid_layout')
    return list(offs)


class _NWIEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, module):
        m = x.shape[0]
        dev = x.device
        lib = _lib.load()
        w16 = module._half_params()
        need_grad = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])  # grad mode is off inside Function.forward
        out = torch.empty(m, module._out_ld, dtype=torch.float16, device=dev)
        rows = (m + 31) // 32 * 32   # nrc_nwie_save_rows: the saved state is laid out in whole 32-sample tiles
        save_in = torch.empty(rows, 32, dtype=torch.float16, device=dev) if need_grad else None
        save_acts = torch.empty(module.n_hidden, rows, _WIDTH, dtype=torch.float16, device=dev) if need_grad else None
        if module.encoding == 0:
            xin = x.detach().to(torch.float32).contiguous()
            in_ld = 3
        else:
            xin = x.detach().to(torch.float16).contiguous()
            in_ld = xin.shape[1]
        g = module.grid_cfg
        ws = torch.empty(int(lib.nrc_nwie_forward_ws_bytes(m)), dtype=torch.uint8, device=dev) if module.encoding == 0 else None
        _lib.check(lib.nrc_nwie_forward(
            module.encoding | (module.n_features << 8), _lib.ptr(xin), in_ld, m, _lib.ptr(w16), _lib.ptr(module._table16()), g['n_levels'],
            g['log2_hashmap_size'], g['base_resolution'], float(g['per_level_scale']), module.n_hidden, module.out_act,
            _PAD, _lib.ptr(out), module._out_ld, module._out_ld, _lib.ptr(save_in), _lib.ptr(save_acts), _lib.ptr(ws),
            _lib.stream_of(out)), 'nwie_forward')
        if need_grad:
            ctx.module = module
            ctx.x_dtype = x.dtype
            ctx.x_requires_grad = x.requires_grad
            ctx.save_for_backward(xin, out, save_in, save_acts, w16)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        xin, out, save_in, save_acts, w16 = ctx.saved_tensors
        module = ctx.module
        lib = _lib.load()
        m = xin.shape[0]
        dev = xin.device
        d_out = d_out.to(torch.float16).contiguous()
        general_grid = module.encoding == 0 and not (module.grid_cfg['n_levels'] == 16 and module.n_features == 2)
        grad_params = torch.zeros(module._n_kernel_mlp + (module.params.numel() - module.n_mlp_params), dtype=torch.float32, device=dev)   # kernel layout
        pair_major = 1 if module.encoding == 0 and not general_grid else 0  # the tuned grid backward reads [16 levels][m][2]
        d_in = torch.empty(m, 32, dtype=torch.float32, device=dev)
        st = _lib.stream_of(d_out)
        _lib.check(lib.nrc_nwie_backward(
            m, _lib.ptr(w16), module.n_hidden, module.out_act, _PAD, _lib.ptr(d_out), _lib.ptr(out), module._out_ld,
            _lib.ptr(save_in), _lib.ptr(save_acts), LOSS_SCALE, _lib.ptr(grad_params), _lib.ptr(d_in), pair_major, st), 'nwie_backward')
        grad_x = None
        if module.encoding == 0:
            g = module.grid_cfg
            table_grad = grad_params[module._n_kernel_mlp:]
            if general_grid:
                _lib.check(lib.nrc_grid_backward_general(_lib.ptr(xin), m, _lib.ptr(d_in), g['n_levels'], module.n_features, g['log2_hashmap_size'],
                                                         g['base_resolution'], float(g['per_level_scale']), _lib.ptr(table_grad), st), 'grid_backward_general')
                return grad_x, module._grad_to_master(grad_params), None
            gws = torch.empty(int(lib.nrc_grid_backward_ws_bytes(m, g['n_levels'], g['log2_hashmap_size'], g['base_resolution'], float(g['per_level_scale']))),
                              dtype=torch.uint8, device=dev)
            _lib.check(lib.nrc_grid_backward(_lib.ptr(xin), m, _lib.ptr(d_in), pair_major, g['n_levels'], g['log2_hashmap_size'],
                                             g['base_resolution'], float(g['per_level_scale']), _lib.ptr(table_grad), _lib.ptr(gws), st), 'grid_backward')
        elif ctx.x_requires_grad:
            # gradient w.r.t. the identity-encoded dims; the SH-encoded direction dims get zeros (view directions are data,
            # never optimised by the reference: Renderer.py:40 feeds rays.view_direction)
            grad_x = torch.zeros(m, xin.shape[1], dtype=ctx.x_dtype, device=dev)
            grad_x[:, 3:19] = d_in[:, 16:32].to(ctx.x_dtype)
        return grad_x, module._grad_to_master(grad_params), None


class NetworkWithInputEncoding(torch.nn.Module):
    def __init__(self, n_input_dims: int, n_output_dims: int, encoding_config: dict, network_config: dict, seed: int = 1337) -> None:
        super().__init__()
        self.n_input_dims = int(n_input_dims)
        self.n_output_dims = int(n_output_dims)
        self.encoding_config = encoding_config
        self.network_config = network_config
        self.seed = int(seed)
        self.jit_fusion = False
        self.loss_scale = LOSS_SCALE
        # ---- network
        if network_config.get('otype') not in ('FullyFusedMLP', 'CutlassMLP'):
            raise RuntimeError(f"nerficg_amd.tinycudann: unsupported network otype {network_config.get('otype')!r}")
        if str(network_config.get('activation', 'ReLU')).lower() != 'relu':
            raise RuntimeError('nerficg_amd.tinycudann: only ReLU hidden activations are implemented')
        if int(network_config.get('n_neurons', 64)) != _WIDTH:
            raise RuntimeError(f"nerficg_amd.tinycudann: only n_neurons = 64 is implemented (got {network_config.get('n_neurons')}): N_DENSITY_NEURONS / N_COLOR_NEURONS "
                               'of the InstantNGP yaml must keep their default')
        self.n_hidden = int(network_config.get('n_hidden_layers', 1))
        if self.n_hidden not in (1, 2):
            raise RuntimeError(f'nerficg_amd.tinycudann: n_hidden_layers must be 1 or 2 (got {self.n_hidden}): N_DENSITY_LAYERS / N_COLOR_LAYERS of the InstantNGP yaml')
        oa = str(network_config.get('output_activation', 'None')).lower()
        if oa not in ('none', 'sigmoid'):
            raise RuntimeError(f'nerficg_amd.tinycudann: unsupported output_activation {oa!r}')
        self.out_act = 1 if oa == 'sigmoid' else 0
        if not 1 <= self.n_output_dims <= _PAD:
            raise RuntimeError('nerficg_amd.tinycudann: n_output_dims must be in [1, 16]')
        self._out_ld = 4 * ((self.n_output_dims + 3) // 4)  # stored output columns (the rest of the padded 16 is never read)
        # ---- encoding
        ot = str(encoding_config.get('otype', ''))
        self.grid_cfg = dict(n_levels=1, log2_hashmap_size=4, base_resolution=2, per_level_scale=2.0)
        n_table = 0
        if ot in ('Grid', 'HashGrid'):
            if str(encoding_config.get('type', 'Hash')) != 'Hash' or str(encoding_config.get('interpolation', 'Linear')) != 'Linear':
                raise RuntimeError('nerficg_amd.tinycudann: only Hash grids with Linear interpolation are implemented')
            n_levels, n_feat = int(encoding_config.get('n_levels', 16)), int(encoding_config.get('n_features_per_level', 2))
            if self.n_input_dims != 3 or n_feat not in (2, 4) or n_levels < 1 or n_levels * n_feat > 32:
                raise RuntimeError(f"nerficg_amd.tinycudann: the grid encoding is implemented for n_input_dims = 3, n_features_per_level in (2, 4) and "
                                   f"n_levels * n_features_per_level <= 32 (got {self.n_input_dims}, {n_levels} levels, {n_feat} features): "
                                   'HASHGRID_N_LEVELS / HASHGRID_N_FEATURES_PER_LEVEL of the InstantNGP yaml (16 x 2 by default; e.g. 8 x 4 or 12 x 2 are fine, 16 x 4 is not)')
            self.encoding = 0
            self.n_features = n_feat
            self.grid_cfg = dict(n_levels=n_levels, log2_hashmap_size=int(encoding_config.get('log2_hashmap_size', 19)),
                                 base_resolution=int(encoding_config.get('base_resolution', 16)),
                                 per_level_scale=float(encoding_config.get('per_level_scale', 2.0)))
            self.grid_offsets = _grid_offsets(self.grid_cfg)
            n_table = self.grid_offsets[-1] * n_feat
            n_encoded = n_levels * n_feat
            cols = [k if k < n_encoded else -1 for k in range(32)]           # kernel input k = level * F + c, as tiny-cuda-nn orders them
        elif ot == 'Composite':
            nested = encoding_config.get('nested', [])
            degree = int(nested[0].get('degree', 0)) if len(nested) == 2 else 0
            ok = (len(nested) == 2 and nested[0].get('otype') == 'SphericalHarmonics' and 1 <= degree <= 4
                  and int(nested[0].get('n_dims_to_encode', 0)) == 3 and nested[1].get('otype') == 'Identity' and self.n_input_dims == 19)
            if not ok:
                raise RuntimeError('nerficg_amd.tinycudann: Composite encoding must be [SphericalHarmonics(degree = 1..4, n_dims_to_encode = 3), Identity] with n_input_dims = 19 '
                                   f'(got n_input_dims = {self.n_input_dims}, nested = {nested}): DIR_SH_ENCODING_DEGREE / N_DENSITY_OUTPUT_FEATURES of the InstantNGP yaml '
                                   '(1..4 / 16)')
            self.encoding = 1
            self.n_features = 0
            self.sh_degree = degree
            n_encoded = degree * degree + 16
            # the kernels evaluate all 16 degree-4 coefficients into inputs 0..15 and take the 16 identity dims as inputs 16..31; tiny-cuda-nn's columns are
            # [SH 0 .. d^2 - 1 | identity 0 .. 15 | padding]: coefficients the configuration does not have meet zero weights
            cols = [k if k < degree * degree else -1 for k in range(16)] + [degree * degree + j for j in range(16)]
        else:
            raise RuntimeError(f'nerficg_amd.tinycudann: unsupported encoding otype {ot!r}')
        self.n_in_padded = (n_encoded + 15) // 16 * 16     # tiny-cuda-nn pads the encoded width to a multiple of 16: the first layer is (64, n_in_padded)
        # first-layer columns: kernel input k reads master column _w0_cols[k] (-1: no such input, weight 0).  None = the master layout IS the kernel layout
        self._w0_cols = None if cols == list(range(32)) and self.n_in_padded == 32 else cols
        # ---- parameters: [W0 (64,32) | hidden (64,64)... | Wout (16,64)] then the grid table (Model.py:40 slices on this order)
        self.n_mlp_params = _WIDTH * self.n_in_padded + (self.n_hidden - 1) * _WIDTH * _WIDTH + _PAD * _WIDTH
        self._n_kernel_mlp = _WIDTH * 32 + (self.n_hidden - 1) * _WIDTH * _WIDTH + _PAD * _WIDTH      # what the kernels read: the first layer always (64, 32)
        gen = torch.Generator(device='cpu').manual_seed(self.seed)
        chunks = []
        shapes = [(_WIDTH, self.n_in_padded)] + [(_WIDTH, _WIDTH)] * (self.n_hidden - 1) + [(_PAD, _WIDTH)]
        for fan_out, fan_in in shapes:  # Xavier uniform, like tiny-cuda-nn's FullyFusedMLP::initialize_params
            s = math.sqrt(6.0 / (fan_in + fan_out))
            chunks.append((torch.rand(fan_out * fan_in, generator=gen) * 2 - 1) * s)
        if n_table:
            chunks.append((torch.rand(n_table, generator=gen) * 2 - 1) * 1e-4)  # grid init U(-1e-4, 1e-4)
        self.params = torch.nn.Parameter(torch.cat(chunks).to(torch.float32), requires_grad=True)
        self._half = None
        self._half_key = None
        # nerficg_amd.apex_optimizers.FusedAdam updates `params` through a raw pointer: it finds this module through the parameter and
        # has the step kernel rewrite the fp16 copy as well (_half_for_optimizer / _half_written_by_optimizer)
        self.params._nrc_half_owner = weakref.ref(self)

    @staticmethod
    def _key_of(p):
        return (p.data_ptr(), p._version, p.device)

    def _half_for_optimizer(self, p):
        """fp16 copy for the Adam kernel to rewrite (None if `p` is not this module's live CUDA parameter). The copy is brought up to date
        first, because an overflow-skipped step leaves it untouched."""
        if p is not self.params or not p.is_cuda or self._w0_cols is not None:   # (a remapped first layer: the copy is not a 1:1 mirror, rebuilt on use instead)
            return None
        self._refresh_half()
        return self._half

    def _half_written_by_optimizer(self, p) -> None:
        if p is self.params and self._half is not None:
            self._half_key = self._key_of(p)

    # fp16 compute copy of the fp32 master parameters, refreshed when the parameter tensor changes
    def _refresh_half(self) -> None:
        p = self.params
        key = self._key_of(p)
        if self._half is None or self._half_key != key:
            if not p.is_cuda:
                raise RuntimeError('nerficg_amd.tinycudann: parameters must live on the GPU (no CPU fallback)')
            if self._w0_cols is None:
                half = torch.empty(p.numel(), dtype=torch.float16, device=p.device)
                _lib.check(_lib.load().nrc_f32_to_f16(_lib.ptr(p.detach()), _lib.ptr(half), p.numel(), _lib.stream_of(half)), 'f32_to_f16')
            else:   # kernel layout: first layer (64, 32) gathered from the master's (64, n_in_padded) columns, everything behind it unchanged
                pd, n0 = p.detach(), _WIDTH * self.n_in_padded
                idx, live = self._w0_index(p.device)
                w0 = pd[:n0].view(_WIDTH, self.n_in_padded)[:, idx] * live
                half = torch.cat([w0.reshape(-1), pd[n0:]]).to(torch.float16)
            self._half, self._half_key = half, key

    def _w0_index(self, device):
        """(master column per kernel input, clamped; 0 / 1 mask of the inputs that exist) for a remapped first layer."""
        cols = torch.tensor(self._w0_cols, dtype=torch.long, device=device)
        return cols.clamp_min(0), (cols >= 0).to(torch.float32)

    def _grad_to_master(self, grad_kernel: torch.Tensor) -> torch.Tensor:
        """Gradient in the kernels' parameter layout -> the layout of `params` (identity unless the first layer is remapped)."""
        if self._w0_cols is None:
            return grad_kernel
        n0k = _WIDTH * 32
        idx, live = self._w0_index(grad_kernel.device)
        g0 = torch.zeros(_WIDTH, self.n_in_padded, dtype=grad_kernel.dtype, device=grad_kernel.device)
        g0.index_add_(1, idx, grad_kernel[:n0k].view(_WIDTH, 32) * live)
        return torch.cat([g0.reshape(-1), grad_kernel[n0k:]])

    def _half_params(self) -> torch.Tensor:
        self._refresh_half()
        return self._half

    def _table16(self):
        return self._half[self._n_kernel_mlp:] if self.encoding == 0 else None

    @property
    def default_layout(self) -> bool:
        """The configuration the fused InstantNGP paths are built for (16 x 2 grid / degree-4 SH): the general ones go through this module's forward."""
        return self._w0_cols is None and (self.encoding != 0 or (self.grid_cfg['n_levels'] == 16 and self.n_features == 2))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError('nerficg_amd.tinycudann: input must be a CUDA tensor (no CPU fallback)')
        if x.dim() != 2 or x.shape[1] != self.n_input_dims:
            raise RuntimeError(f'nerficg_amd.tinycudann: expected input of shape (M, {self.n_input_dims}), got {tuple(x.shape)}')
        out = _NWIEFunction.apply(x, self.params, self)
        return out[:, :self.n_output_dims]

    def extra_repr(self) -> str:
        return f'n_input_dims={self.n_input_dims}, n_output_dims={self.n_output_dims}, seed={self.seed}, encoding={self.encoding_config}, network={self.network_config}'
