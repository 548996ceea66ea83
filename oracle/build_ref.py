"""Build what can be built of the REFERENCE itself, from its sources where they lie under /root/reference, into oracle/_ref/
(git-ignored; travels to the GPU box with the snapshot).  Test infrastructure: only tests/ load the result.

  libref_bev_pool.so   projects/mmdet3d_plugin/models/utils/ops/bev_pool/src/bev_pool_cuda.cu compiled AS IT IS with
                       `hipcc --offload-arch=gfx950 -include hip/hip_runtime.h` (the file includes only <stdio.h> / <stdlib.h>;
                       the pre-include supplies the HIP spelling of the <<<...>>> launch it uses).  Exports the reference's own
                       launcher `bev_pool(b, d, h, w, n, c, n_intervals, x, geom_feats, interval_starts, interval_lengths,
                       out)` (C++-mangled) - the CUDA extension behind `bev_pool_ext.bev_pool_forward` (bev_pool.cpp:21-53).

Unbuildable here (and therefore still pinned by restatement only): locatt_ops (similar.cu / weighting.cu include utils.cuh ->
<cuda.h>, <cuda_runtime.h>, <ATen/cuda/CUDAContext.h>: CUDA toolkit headers this ROCm image does not have; no stand-ins are
written), and everything behind mmcv / mmdet / mmdet3d.

    python -m oracle.build_ref
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')
REF = '/root/reference'
BEV_POOL_SRC = os.path.join(REF, 'projects/mmdet3d_plugin/models/utils/ops/bev_pool/src/bev_pool_cuda.cu')
BEV_POOL_LIB = os.path.join(OUT, 'libref_bev_pool.so')
BEV_POOL_SYMBOL = '_Z8bev_pooliiiiiiiPKfPKiS2_S2_Pf'      # void bev_pool(int x7, const float*, const int* x3, float*)
BEV_POOL_GRAD_SYMBOL = '_Z13bev_pool_gradiiiiiiiPKfPKiS2_S2_Pf'     # void bev_pool_grad(...): the launcher of bev_pool_grad_kernel (:93-98)


def build(verbose=True):
    """Returns the list of built libraries ([] when /root/reference is absent - e.g. on the GPU box, which uses the
    prebuilt files of the snapshot)."""
    if not os.path.exists(BEV_POOL_SRC):
        return []
    os.makedirs(OUT, exist_ok=True)
    if not os.path.exists(BEV_POOL_LIB) or os.path.getmtime(BEV_POOL_LIB) < os.path.getmtime(BEV_POOL_SRC):
        cmd = [os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-O2', '-fPIC', '-shared',
               '-include', 'hip/hip_runtime.h', BEV_POOL_SRC, '-o', BEV_POOL_LIB]
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return [BEV_POOL_LIB]


if __name__ == '__main__':
    print(build())
