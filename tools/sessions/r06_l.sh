#!/bin/bash
# round 6 session l: timing ablations of locatt_mfma_kernel (experiments library)
E=$PWD/focalformer3d_amd/lib/libff3d_hip_exp.so
python tools/experiments/exp_locatt_mfma.py 2>&1 | grep "C=256" | sed "s/^/shipped /"
for a in 0 1 2 4 8 3 6 7 15; do FF3D_LIB=$E FF3D_LA_ABLATE=$a python tools/experiments/exp_locatt_mfma.py 2>&1 | grep "C=256" | sed "s/^/ABLATE=$a /"; done
