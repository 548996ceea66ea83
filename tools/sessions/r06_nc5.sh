#!/bin/bash
# round 6 experiment, final record: halo conv over the caller's NCHW fp32 map (experiments library) vs conversion pass + conv; product conv tests
O=$PWD/gpurun_out/r06_nc5; mkdir -p $O
for rep in 1 2 3; do
  FF3D_LIB=$PWD/focalformer3d_amd/lib/libff3d_hip_exp.so B=32 H=180 W=180 GEO=0 timeout 300 python tools/experiments/exp_halo_nchw.py 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
done
cat $O/ab.txt
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_round5_gpu.py tests/test_bench_shape_gpu.py -q -m gpu -k "conv or halo or heatmap or bench_shape or head" 2>&1 | tail -4 | tee $O/tests.txt
