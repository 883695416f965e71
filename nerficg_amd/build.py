"""nerficg_amd/build.py -- compiles csrc/*.hip for gfx950 into nerficg_amd/lib/libnerficg_hip.so (in-tree).

hipcc cross-compiles without a GPU, so this runs in the build container as well as on the MI355X box.
Usage: python -m nerficg_amd.build [--force]
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / 'csrc'
LIBDIR = PKG / 'lib'
LIB = LIBDIR / 'libnerficg_hip.so'
OBJDIR = LIBDIR / 'obj'
ARCH = 'gfx950'

COMMON_FLAGS = ['-O3', '-fPIC', f'--offload-arch={ARCH}', '-std=c++17', '-Wall', '-Wno-unused-function',
                '-Wno-unused-result', '-Wno-unused-value', '-fno-gpu-rdc', '-DNDEBUG']
# translation units whose f32 arithmetic decides integer indices: no FMA contraction (bit-exact vs oracle/)
PER_FILE_FLAGS = {
    'ngp_march.hip': ['-ffp-contract=off'],
    'adam.hip': ['-ffp-contract=off'],
    'knn.hip': ['-ffp-contract=off'],
    'gs_densify.hip': ['-ffp-contract=off'],   # thresholds decide the row list: same f32 sequence as oracle/gs_densify.py   # squared distances decide the neighbour set: same f32 sequence as oracle/knn_oracle.c
  # HBM-bound anyway; keeps the update bit-identical to oracle/adam_oracle.c
    # + no atomic optimizer: it rewrites the one-lane LDS atomics of k_render_bw into 15-instruction wave-reduction loops
    # + no SLP vectoriser: it packs the blend arithmetic of two list entries into v_pk_* and pays in v_mov (the blend states its own pairs: blend_power)
    'gs_raster.hip': ['-ffp-contract=off', '-fno-slp-vectorize', '-mllvm', '-amdgpu-atomic-optimizer-strategy=None'],
    # no SLP vectoriser either: packed f32 VALU next to MFMAs is priced above its issue slot on this chip (MI355X_MICROARCH.md), and the packing
    # costs v_mov: k_nwie_bwd 2 641 -> 2 451 vector instructions, k_gb_split 3 015 -> 2 853; the fused training iteration -1 %, the image path unchanged
    'ngp_net.hip': ['-fno-slp-vectorize'],
}


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (Path(cand).exists() or cand == 'hipcc'):
            return cand
    raise RuntimeError('hipcc not found')


def _digest(paths: list[Path], extra: str) -> str:
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> Path:
    sources = sorted(CSRC.glob('*.hip'))
    headers = sorted(CSRC.glob('*.h')) + sorted((PKG.parent / 'include').glob('*.h'))
    if not sources:
        raise RuntimeError(f'no .hip sources in {CSRC}')
    OBJDIR.mkdir(parents=True, exist_ok=True)
    hipcc = _hipcc()
    hdr_digest = _digest(headers, ' '.join(COMMON_FLAGS))

    def compile_one(src: Path) -> tuple[Path, bool]:
        flags = COMMON_FLAGS + PER_FILE_FLAGS.get(src.name, [])
        obj = OBJDIR / (src.stem + '.o')
        stamp = OBJDIR / (src.stem + '.sha')
        digest = _digest([src], hdr_digest + ' '.join(flags))
        if not force and obj.exists() and stamp.exists() and stamp.read_text() == digest:
            return obj, False
        cmd = [hipcc, *flags, '-c', str(src), '-o', str(obj)]
        if verbose:
            print('[build]', ' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError(f'hipcc failed on {src.name}')
        if r.stderr.strip() and verbose:
            sys.stderr.write(r.stderr)
        stamp.write_text(digest)
        return obj, True

    with ThreadPoolExecutor(max_workers=min(4, len(sources))) as ex:
        results = list(ex.map(compile_one, sources))
    objs = [o for o, _ in results]
    if force or any(changed for _, changed in results) or not LIB.exists():
        cmd = [hipcc, '-shared', '-fPIC', f'--offload-arch={ARCH}', '-o', str(LIB), *map(str, objs)]
        if verbose:
            print('[build]', ' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
