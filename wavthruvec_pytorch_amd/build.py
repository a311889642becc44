"""Build libvec2wav_hip.so (gfx950) in-tree with hipcc.  hipcc cross-compiles without a GPU.

One object per source under csrc/_obj/ (compiled in parallel, re-compiled only when the source or a shared header is
newer), then one link.  The objects and the library are build artefacts (git-ignored); only the library is loaded.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, 'csrc')
OBJ_DIR = os.path.join(CSRC, '_obj')
LIB_PATH = os.path.join(PKG_DIR, 'libvec2wav_hip.so')
SOURCES = ['v2w_api.hip', 'v2w_conv_mfma.hip', 'v2w_conv_split.hip', 'v2w_conv_bf16.hip', 'v2w_stage_split.hip', 'v2w_stage_bf16.hip', 'v2w_resblock_fused.hip',
           'v2w_wgrad.hip', 'v2w_backward.hip', 'v2w_direct.hip', 'v2w_cbn.hip', 'v2w_fold.hip', 'v2w_mel.hip', 'v2w_disc.hip']
HEADERS = ['v2w_common.h', 'v2w_tile.h', os.path.join('..', '..', 'include', 'vec2wav_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']


def find_hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (set HIPCC or install ROCm)')


def _obj(src: str) -> str:
    return os.path.join(OBJ_DIR, src + '.o')


def _stale_sources():
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    out = []
    for s in SOURCES:
        o = _obj(s)
        if not os.path.exists(o) or os.path.getmtime(o) < max(hdr_t, os.path.getmtime(os.path.join(CSRC, s))):
            out.append(s)
    return out


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, jobs: int | None = None) -> str:
    """Compile every HIP source for gfx950 into one shared library next to the package."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = find_hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo = list(SOURCES) if force else _stale_sources()

    def compile_one(s):
        cmd = [hipcc] + FLAGS + ['-c', '-o', _obj(s), os.path.join(CSRC, s)]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed on {s}:\n' + r.stdout + r.stderr)

    jobs = jobs or max(1, min(len(todo) or 1, (os.cpu_count() or 2)))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', LIB_PATH] + [_obj(s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc link failed:\n' + r.stdout + r.stderr)
    return LIB_PATH


if __name__ == '__main__':
    import sys
    print(build(force='--force' in sys.argv, verbose=True))
