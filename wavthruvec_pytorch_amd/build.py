"""Build libvec2wav_hip.so (gfx950) in-tree with hipcc.  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, 'csrc')
LIB_PATH = os.path.join(PKG_DIR, 'libvec2wav_hip.so')
SOURCES = ['v2w_api.hip', 'v2w_conv_mfma.hip', 'v2w_conv_split.hip', 'v2w_stage_split.hip', 'v2w_resblock_fused.hip', 'v2w_wgrad.hip', 'v2w_backward.hip', 'v2w_direct.hip', 'v2w_cbn.hip', 'v2w_fold.hip', 'v2w_mel.hip', 'v2w_disc.hip']
HEADERS = ['v2w_common.h', 'v2w_tile.h', os.path.join('..', '..', 'include', 'vec2wav_hip.h')]


def find_hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (set HIPCC or install ROCm)')


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into one shared library next to the package."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [find_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           '-o', LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + r.stdout + r.stderr)
    return LIB_PATH


if __name__ == '__main__':
    print(build(force=True, verbose=True))
