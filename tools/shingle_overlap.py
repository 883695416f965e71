#!/usr/bin/env python3
"""tools/shingle_overlap.py -- share of a reference file's 10-token shingles (comments / docstrings stripped) that reappear in a file of this
repository.  Development aid for keeping host code original; reads /root/reference, so it only runs in the build container."""
import io
import sys
import tokenize


def tokens(path):
    out = []
    src = open(path, 'rb').read()
    prev = None
    for tok in tokenize.tokenize(io.BytesIO(src).readline):
        if tok.type in (tokenize.COMMENT, tokenize.NL, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT, tokenize.ENCODING, tokenize.ENDMARKER):
            prev = tok.type
            continue
        if tok.type == tokenize.STRING and prev in (None, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT, tokenize.NL, tokenize.ENCODING):
            prev = tok.type  # docstring / bare string statement
            continue
        out.append(tok.string)
        prev = tok.type
    return out


def shingles(toks, k=10):
    return {tuple(toks[i:i + k]) for i in range(len(toks) - k + 1)}


PAIRS = [
    ('nerficg_amd/VolumeRenderingV2/__init__.py', 'src/Methods/InstantNGP/VolumeRenderingV2/custom_functions.py'),
    ('nerficg_amd/instant_ngp.py', 'src/Methods/InstantNGP/Renderer.py'),
    ('nerficg_amd/instant_ngp.py', 'src/Methods/InstantNGP/Model.py'),
    ('nerficg_amd/gaussian_splatting.py', 'src/Methods/GaussianSplatting/utils.py'),
    ('nerficg_amd/gaussian_splatting.py', 'src/Methods/GaussianSplatting/Model.py'),
    ('nerficg_amd/gaussian_splatting.py', 'src/Methods/GaussianSplatting/Renderer.py'),
    ('nerficg_amd/gaussian_splatting.py', 'src/Cameras/utils.py'),
    ('nerficg_amd/nerf.py', 'src/Methods/NeRF/utils.py'),
    ('nerficg_amd/nerf.py', 'src/Methods/NeRF/Model.py'),
    ('nerficg_amd/nerf.py', 'src/Methods/NeRF/Renderer.py'),
    ('nerficg_amd/adam_utils.py', 'src/Optim/adam_utils.py'),
    ('nerficg_amd/rays.py', 'src/Datasets/utils.py'),
    ('nerficg_amd/samplers.py', 'src/Optim/Samplers/utils.py'),
    ('nerficg_amd/samplers.py', 'src/Optim/Samplers/DatasetSamplers.py'),
    ('nerficg_amd/lr_utils.py', 'src/Optim/lr_utils.py'),
    ('nerficg_amd/formats.py', 'src/Methods/Base/Model.py'),
    ('nerficg_amd/raygen.py', 'src/Cameras/Perspective.py'),
    ('nerficg_amd/parallel.py', 'src/Methods/Base/Renderer.py'),
]

if __name__ == '__main__':
    for mine, ref in PAIRS:
        a, b = shingles(tokens(mine)), shingles(tokens('/root/reference/' + ref))
        if not b:
            continue
        print(f'{len(a & b) / len(b) * 100:5.1f} % of {ref:70s} in {mine}')


def show(mine, ref, k=10):
    """prints the token runs of `mine` that reappear in `ref` (development aid)"""
    tm, tr = tokens(mine), tokens('/root/reference/' + ref)
    sr = shingles(tr, k)
    hit = [tuple(tm[i:i + k]) in sr for i in range(len(tm) - k + 1)]
    i = 0
    while i < len(hit):
        if hit[i]:
            j = i
            while j < len(hit) and hit[j]:
                j += 1
            print('   ', ' '.join(tm[i:j + k - 1])[:400])
            i = j + k
        else:
            i += 1
