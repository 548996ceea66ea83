#!/bin/bash
# round 4 session e: halo conv with the weights loaded straight into registers (no LDS staging of weights, one barrier per chunk)
O=$PWD/gpurun_out/r04_e; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export FF3D_LIB=$R/focalformer3d_amd/lib/libff3d_hip_exp.so
for w in 0 1 0 1; do echo -n "FF3D_HALO_WREG=$w: " | tee -a $O/halo_wreg_ab.txt; FF3D_HALO_WREG=$w timeout 120 python tools/experiments/exp_halo.py 2>&1 | tail -1 | tee -a $O/halo_wreg_ab.txt; done
for w in 0 1; do echo -n "B=4 FF3D_HALO_WREG=$w: " | tee -a $O/halo_wreg_ab.txt; B=4 FF3D_HALO_WREG=$w timeout 120 python tools/experiments/exp_halo.py 2>&1 | tail -1 | tee -a $O/halo_wreg_ab.txt; done
FF3D_HALO_WREG=1 timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_bench_shape_gpu.py -x -q -m gpu -k "conv or halo or dense or split" > $O/pytest_wreg.log 2>&1; echo "tests (wreg) rc=$?"; tail -2 $O/pytest_wreg.log | cut -c1-300
