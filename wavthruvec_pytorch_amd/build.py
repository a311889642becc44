"""Build libvec2wav_hip.so (gfx950) in-tree with hipcc.  hipcc cross-compiles without a GPU.

One object per source under csrc/_obj/ (compiled in parallel, re-compiled only when the source or a shared header is
newer), then one link.  The objects and the library are build artefacts (git-ignored); only the library is loaded.
"""
from __future__ import annotations

import contextlib
import fcntl
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, 'csrc')
OBJ_DIR = os.path.join(CSRC, '_obj')
LIB_PATH = os.path.join(PKG_DIR, 'libvec2wav_hip.so')
SOURCES = ['v2w_api.hip', 'v2w_conv_mfma.hip', 'v2w_conv_split.hip', 'v2w_conv_bf16.hip', 'v2w_conv_bf16_res.hip', 'v2w_convt_bf16_res.hip', 'v2w_stage_split.hip', 'v2w_stage_bf16.hip', 'v2w_stage_bf16_wide.hip', 'v2w_stage_bf16_n16.hip', 'v2w_stage_bf16_n16s.hip', 'v2w_stage_bf16_n32s.hip', 'v2w_resblock_fused.hip', 'v2w_stage_small.hip',
           'v2w_wgrad.hip', 'v2w_wgrad_bf16.hip', 'v2w_backward.hip', 'v2w_direct.hip', 'v2w_conv_post_bf16.hip', 'v2w_cbn.hip', 'v2w_fold.hip', 'v2w_mel.hip', 'v2w_disc.hip']
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


def sources_hash() -> str:
    """sha256 (16 hex digits) over every kernel source and header the library is built from: profiles record it so that a counter
    file collected from other sources is recognised as stale (bench.py `roofline.traffic`)."""
    hsh = hashlib.sha256()
    for f in sorted(SOURCES) + sorted(HEADERS):
        with open(os.path.join(CSRC, f), 'rb') as fh:
            hsh.update(os.path.basename(f).encode() + b'\0' + fh.read())
    return hsh.hexdigest()[:16]


@contextlib.contextmanager
def _build_lock():
    """One builder at a time per checkout: N ranks importing the package after a source edit (torchrun, bench.py --gpus N) would
    otherwise compile and link into the same paths at once."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(os.path.join(OBJ_DIR, '.lock'), 'w') as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def build(force: bool = False, verbose: bool = False, jobs: int | None = None) -> str:
    """Compile every HIP source for gfx950 into one shared library next to the package.  Objects and the library are written under a
    temporary name and renamed into place (an interrupted compile never leaves a truncated file that looks fresh)."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = find_hipcc()
    with _build_lock():
        if not force and not needs_build():       # another process built it while this one waited for the lock
            return LIB_PATH
        todo = list(SOURCES) if force else _stale_sources()

        def run(cmd, what):
            if verbose:
                print(' '.join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f'{what}:\n' + r.stdout + r.stderr)

        def compile_one(s):
            tmp = _obj(s) + f'.tmp{os.getpid()}'
            try:
                run([hipcc] + FLAGS + ['-c', '-o', tmp, os.path.join(CSRC, s)], f'hipcc failed on {s}')
                os.replace(tmp, _obj(s))
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)

        jobs = jobs or max(1, min(len(todo) or 1, (os.cpu_count() or 2)))
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(compile_one, todo))
        tmp = LIB_PATH + f'.tmp{os.getpid()}'
        try:
            run([hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', tmp] + [_obj(s) for s in SOURCES], 'hipcc link failed')
            os.replace(tmp, LIB_PATH)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return LIB_PATH


if __name__ == '__main__':
    import sys
    print(build(force='--force' in sys.argv, verbose=True))
