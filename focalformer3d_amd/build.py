"""Build recipe for libff3d_hip.so (gfx950 only).  hipcc cross-compiles without a GPU.

    python -m focalformer3d_amd.build          # rebuild what is stale
    python -m focalformer3d_amd.build --force  # rebuild everything

Incremental: every csrc/*.hip is compiled to its own object under lib/obj/ (in parallel, only when the source or a
header is newer than the object), then linked into focalformer3d_amd/lib/libff3d_hip.so (in-tree, so the built library
travels with the repo snapshot to the GPU box).
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB_DIR = os.path.join(PKG, 'lib')
OBJ_DIR = os.path.join(LIB_DIR, 'obj')
LIB_PATH = os.path.join(LIB_DIR, 'libff3d_hip.so')
SOURCES = sorted(glob.glob(os.path.join(PKG, 'csrc', '*.hip')))
HEADERS = sorted(glob.glob(os.path.join(PKG, 'csrc', '*.h'))) + [os.path.join(ROOT, 'include', 'ff3d.h')]
CFLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include')]


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + '.o')


def _newer(path, than):
    return (not os.path.exists(path)) or any(os.path.getmtime(f) > os.path.getmtime(path) for f in than)


def stale():
    return _newer(LIB_PATH, SOURCES + HEADERS)


def build(force=False, verbose=True):
    """Compile every HIP source into focalformer3d_amd/lib/libff3d_hip.so; returns its path."""
    if not force and not stale():
        return LIB_PATH
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo = [s for s in SOURCES if force or _newer(_obj(s), [s] + HEADERS)]

    def compile_one(src):
        cmd = [hipcc] + CFLAGS + ['-c', src, '-o', _obj(src)]
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 1))) as ex:
        list(ex.map(compile_one, todo))
    known = {_obj(s) for s in SOURCES}
    for stray in glob.glob(os.path.join(OBJ_DIR, '*.o')):       # objects of deleted sources must not be linked
        if stray not in known:
            os.remove(stray)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + sorted(known) + ['-o', LIB_PATH]
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == '__main__':
    build(force='--force' in sys.argv)
