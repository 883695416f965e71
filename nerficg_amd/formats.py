"""nerficg_amd.formats -- the reference's on-disk formats, so that checkpoints written by nerficg load into the
MI355X-native models and vice versa:

* `.pt` checkpoints of BaseModel.save / BaseModel.load (src/Methods/Base/Model.py:61-111): one pickled dict
  {'model_state_dict', 'model_name', 'creation_date', 'num_iterations_trained', 'output_directory' (pathlib.Path), <configurable parameters>}.
  InstantNGP (src/Methods/InstantNGP/Model.py:14-123): state keys occupancy_grid, occupancy_bitfield, encoding_xyz.params,
  color_mlp_with_encoding.params (flat f32, tiny-cuda-nn order), parameters = the @configure block.
  3DGS (src/Methods/GaussianSplatting/Model.py:18-35,248-273,318-333): state keys gaussians._positions, _features_dc, _features_rest,
  _scales, _rotations, _opacities, _baked_covariances; SH_DEGREE; a trained checkpoint (num_iterations_trained > 0) holds ACTIVATED
  values (bake_activations ran before the save) and is loaded with identity activations.
* `.ply` export (scripts/convert_to_ply.py:18-21 over Model.as_ply_dict / get_ply_dict): binary little-endian (or ascii) PLY, one element per
  dictionary entry, every property `float`, comment lines from the 'comments' entry -- the file plyfile's PlyData(...).write produces.
"""
from __future__ import annotations

import datetime
from collections import OrderedDict
from pathlib import Path

import numpy as np
import torch

__all__ = ['save_checkpoint', 'load_checkpoint', 'instant_ngp_to_checkpoint', 'instant_ngp_from_checkpoint', 'gaussians_to_checkpoint',
           'gaussians_from_checkpoint', 'write_ply', 'read_ply', 'gaussians_ply_dict']

_META = ('model_name', 'creation_date', 'num_iterations_trained', 'output_directory')
INSTANT_NGP_PARAMETERS = ('SCALE', 'RESOLUTION', 'CENTER', 'HASHGRID_N_LEVELS', 'HASHGRID_N_FEATURES_PER_LEVEL', 'HASHGRID_LOG2_SIZE',
                          'HASHGRID_BASE_RESOLUTION', 'HASHGRID_TARGET_RESOLUTION', 'N_DENSITY_OUTPUT_FEATURES', 'N_DENSITY_NEURONS',
                          'N_DENSITY_LAYERS', 'DIR_SH_ENCODING_DEGREE', 'N_COLOR_NEURONS', 'N_COLOR_LAYERS', 'ENABLE_JIT_FUSION')
_GS_KEYS = ('_positions', '_features_dc', '_features_rest', '_scales', '_rotations', '_opacities')


def save_checkpoint(path, state_dict: dict, parameters: dict, model_name: str = 'Default', creation_date: str | None = None,
                    num_iterations_trained: int = 0, output_directory: Path | str | None = None) -> None:
    """BaseModel.save (Base/Model.py:103-111)."""
    checkpoint = {'model_state_dict': state_dict, 'model_name': model_name,
                  'creation_date': creation_date or f'{datetime.datetime.now():%Y-%m-%d-%H-%M-%S}',
                  'num_iterations_trained': int(num_iterations_trained),
                  'output_directory': Path(output_directory) if output_directory is not None else Path('output') / model_name}
    checkpoint.update(parameters)
    torch.save(checkpoint, path)


def load_checkpoint(path, map_location='cpu') -> dict:
    """BaseModel.load's read (Base/Model.py:68): the dictionary holds a pathlib.Path, hence a full unpickle like the reference's."""
    path = Path(path)
    if path.suffix != '.pt':
        raise ValueError(f'Invalid model checkpoint: "{path}"')  # Framework.ModelError in the reference
    checkpoint = torch.load(path, map_location=map_location, weights_only=False)
    if 'model_state_dict' not in checkpoint:
        raise ValueError(f'"{path}" is not a nerficg model checkpoint (no model_state_dict)')
    return checkpoint


# ------------------------------------------------------------------------------------------------ InstantNGP
def instant_ngp_to_checkpoint(model, path, ENABLE_JIT_FUSION: bool = True, **meta) -> None:
    params = {k: getattr(model, k) for k in INSTANT_NGP_PARAMETERS if hasattr(model, k)}
    params['ENABLE_JIT_FUSION'] = ENABLE_JIT_FUSION
    save_checkpoint(path, model.state_dict(), params, **meta)


def instant_ngp_from_checkpoint(path, device='cuda', RANDOM_SEED: int = 1618033989):
    """Returns (model, metadata).  Unknown configurable parameters fall back to the defaults like BaseModel.load does; the flat parameter
    vectors must have the sizes the configuration implies (a mismatch is an error here, not a silent re-registration)."""
    from .instant_ngp import InstantNGPModel
    ck = load_checkpoint(path)
    kwargs = {k: ck[k] for k in INSTANT_NGP_PARAMETERS if k in ck and k != 'ENABLE_JIT_FUSION'}
    model = InstantNGPModel(RANDOM_SEED=RANDOM_SEED, device=device, **kwargs)
    state = ck['model_state_dict']
    own = model.state_dict()
    for key, value in state.items():
        if key not in own:
            raise ValueError(f'unexpected key in InstantNGP checkpoint: "{key}"')
        if tuple(own[key].shape) != tuple(value.shape):
            raise ValueError(f'checkpoint tensor "{key}" has shape {tuple(value.shape)}, the configuration implies {tuple(own[key].shape)}')
    model.load_state_dict(state, strict=False)
    return model, {k: ck.get(k) for k in _META}


# ------------------------------------------------------------------------------------------------ 3DGS
def gaussians_to_checkpoint(gaussians, path, SH_DEGREE: int | None = None, **meta) -> None:
    state = OrderedDict((f'gaussians.{k}', getattr(gaussians, k).detach()) for k in _GS_KEYS)  # registration order of Model.py:25-31
    if gaussians.get_baked_covariances is not None:
        state['gaussians._baked_covariances'] = gaussians.get_baked_covariances.detach()
    if gaussians.baked and int(meta.get('num_iterations_trained', 0)) <= 0:
        raise ValueError('a baked model holds activated values: save it with num_iterations_trained > 0 so that it is loaded without activations')
    save_checkpoint(path, state, {'SH_DEGREE': gaussians.max_sh_degree if SH_DEGREE is None else SH_DEGREE}, **meta)


def gaussians_from_checkpoint(path, device='cuda'):
    """Returns (Gaussians, metadata).  num_iterations_trained > 0 <=> pretrained: activations are identities and all SH degrees are active
    (Model.py:21-24,328-333)."""
    from .gaussian_splatting import Gaussians
    ck = load_checkpoint(path)
    state = ck['model_state_dict']
    missing = [k for k in _GS_KEYS if f'gaussians.{k}' not in state]
    if missing:
        raise ValueError(f'missing key(s) in 3DGS checkpoint: {missing}')
    t = {k: state[f'gaussians.{k}'].to(device=device, dtype=torch.float32).contiguous() for k in _GS_KEYS}
    sh_degree = int(ck.get('SH_DEGREE', 3))
    g = Gaussians(t['_positions'], t['_scales'], t['_rotations'], t['_opacities'], t['_features_dc'], t['_features_rest'], sh_degree=sh_degree)
    pretrained = int(ck.get('num_iterations_trained', 0)) > 0
    g.baked = pretrained
    g.active_sh_degree = sh_degree if pretrained else 0
    if 'gaussians._baked_covariances' in state:
        g._baked_covariances = torch.nn.Parameter(state['gaussians._baked_covariances'].to(device=device, dtype=torch.float32), requires_grad=False)
    return g, {k: ck.get(k) for k in _META}


def gaussians_ply_dict(gaussians) -> dict:
    """GaussianSplattingModel.get_ply_dict (Model.py:335-345): the vertex table + the two comment lines."""
    data = gaussians.as_ply_dict()
    if data:
        data['comments'] = ['SplatRenderMode: default', 'Generated with NeRFICG/GaussianSplatting']
    return data


# ------------------------------------------------------------------------------------------------ PLY
_PLY_TYPES = {'f4': 'float', 'f8': 'double', 'i1': 'char', 'u1': 'uchar', 'i2': 'short', 'u2': 'ushort', 'i4': 'int', 'u4': 'uint'}
_PLY_DTYPES = {v: k for k, v in _PLY_TYPES.items()}
_PLY_DTYPES.update({'float32': 'f4', 'float64': 'f8', 'int8': 'i1', 'uint8': 'u1', 'int16': 'i2', 'uint16': 'u2', 'int32': 'i4', 'uint32': 'u4'})


def write_ply(path, ply_data_dict: dict, use_ascii: bool = False) -> None:
    """save_as_ply (scripts/convert_to_ply.py:18-21): every entry except 'comments' is a structured numpy array = one element."""
    elements = [(name, np.asarray(data)) for name, data in ply_data_dict.items() if name != 'comments']
    lines = ['ply', f'format {"ascii" if use_ascii else "binary_little_endian"} 1.0']
    lines += [f'comment {c}' for c in ply_data_dict.get('comments', [])]
    for name, data in elements:
        if data.dtype.names is None:
            raise ValueError(f'ply element "{name}" must be a structured array')
        lines.append(f'element {name} {data.shape[0]}')
        for prop in data.dtype.names:
            lines.append(f'property {_PLY_TYPES[data.dtype[prop].str[1:]]} {prop}')
    lines.append('end_header')
    with open(path, 'wb') as f:
        f.write(('\n'.join(lines) + '\n').encode('ascii'))
        for _, data in elements:
            if use_ascii:
                for row in data:
                    f.write((' '.join(repr(v.item()) if np.issubdtype(v.dtype, np.integer) else f'{v.item():.9g}' for v in row) + '\n').encode('ascii'))
            else:
                f.write(np.ascontiguousarray(data.astype(data.dtype.newbyteorder('<'), copy=False)).tobytes())


def read_ply(path) -> dict:
    """Inverse of write_ply for scalar properties (the subset this framework writes): {'comments': [...], element: structured array}."""
    with open(path, 'rb') as f:
        if f.readline().strip() != b'ply':
            raise ValueError(f'"{path}" is not a ply file')
        fmt = f.readline().decode('ascii').split()
        if fmt[0] != 'format' or fmt[1] not in ('ascii', 'binary_little_endian'):
            raise ValueError(f'unsupported ply format line: {" ".join(fmt)}')
        comments, elements = [], []
        while True:
            line = f.readline()
            if not line:
                raise ValueError('ply header without end_header')
            words = line.decode('ascii').strip().split()
            if not words:
                continue
            if words[0] == 'end_header':
                break
            if words[0] == 'comment':
                comments.append(line.decode('ascii').strip()[len('comment '):])
            elif words[0] == 'element':
                elements.append((words[1], int(words[2]), []))
            elif words[0] == 'property':
                if words[1] == 'list':
                    raise ValueError('list properties are not supported')
                elements[-1][2].append((words[2], '<' + _PLY_DTYPES[words[1]]))
        out: dict = {'comments': comments}
        for name, count, props in elements:
            dtype = np.dtype(props)
            if fmt[1] == 'ascii':
                rows = [tuple(f.readline().decode('ascii').split()) for _ in range(count)]
                arr = np.array([tuple(np.dtype(t).type(v) for v, (_, t) in zip(r, props)) for r in rows], dtype=dtype)
            else:
                arr = np.frombuffer(f.read(count * dtype.itemsize), dtype=dtype, count=count).copy()
            out[name] = arr
    return out
