"""Build recipe for libff3d_hip.so (gfx950 only).  hipcc cross-compiles without a GPU.

    python -m focalformer3d_amd.build          # rebuild if any source is newer than the .so
"""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB_DIR = os.path.join(PKG, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libff3d_hip.so')
SOURCES = sorted(glob.glob(os.path.join(PKG, 'csrc', '*.hip')))
HEADERS = sorted(glob.glob(os.path.join(PKG, 'csrc', '*.h'))) + [os.path.join(ROOT, 'include', 'ff3d.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-I' + os.path.join(ROOT, 'include')]


def stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=True):
    """Compile every HIP source into focalformer3d_amd/lib/libff3d_hip.so (in-tree, so the built
    library travels with the repo snapshot)."""
    if not force and not stale():
        return LIB_PATH
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc] + FLAGS + SOURCES + ['-o', LIB_PATH]
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == '__main__':
    build(force='--force' in sys.argv)
