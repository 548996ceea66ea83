#!/bin/bash
O=gpurun_out/r02_j; mkdir -p $O
timeout 600 python tools/debug_waymo.py 64 468 > $O/waymo_c64.log 2>&1; tail -25 $O/waymo_c64.log | cut -c1-220
timeout 600 python tools/debug_waymo.py 64 180 > $O/waymo_c64_180.log 2>&1; tail -4 $O/waymo_c64_180.log | cut -c1-220
