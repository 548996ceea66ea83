#!/bin/bash
# round 6: host-side profile of the training step
O=$PWD/gpurun_out/r06_tr4; mkdir -p $O
timeout 900 python tools/profile_host_train.py 256 6 > $O/host_profile.txt 2>&1
grep "^====" $O/host_profile.txt; head -75 $O/host_profile.txt | cut -c1-160
