#!/bin/bash
# round 4 session r: halo conv on v_mfma_f32_32x32x16_f16 (FF3D_HALO_M32=1, experiments library): parity + A/B
O=$PWD/gpurun_out/r04_r; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export FF3D_LIB=$R/focalformer3d_amd/lib/libff3d_hip_exp.so
for w in 0 1 0 1; do echo -n "FF3D_HALO_M32=$w: " | tee -a $O/halo_m32_ab.txt; FF3D_HALO_M32=$w timeout 120 python tools/experiments/exp_halo.py 2>&1 | tail -1 | tee -a $O/halo_m32_ab.txt; done
for w in 0 1; do echo -n "B=4 FF3D_HALO_M32=$w: " | tee -a $O/halo_m32_ab.txt; B=4 FF3D_HALO_M32=$w timeout 120 python tools/experiments/exp_halo.py 2>&1 | tail -1 | tee -a $O/halo_m32_ab.txt; done
FF3D_HALO_M32=1 timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_bench_shape_gpu.py -x -q -m gpu -k "conv or halo or dense or split" > $O/pytest_m32.log 2>&1; echo "tests (m32) rc=$?"; tail -2 $O/pytest_m32.log | cut -c1-300
