#!/bin/bash
# round 6: host profile of the eager inference step at 1 and 4 frames (tools/profile_host.py)
O=$PWD/gpurun_out/r06_hp; mkdir -p $O
timeout 600 python tools/profile_host.py 1 40 > $O/host_b1.txt 2>&1
timeout 600 python tools/profile_host.py 4 40 > $O/host_b4.txt 2>&1
grep "host time per step" $O/host_b1.txt $O/host_b4.txt; sed -n 1,40p $O/host_b1.txt | cut -c1-160
