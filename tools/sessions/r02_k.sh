#!/bin/bash
O=gpurun_out/r02_k; mkdir -p $O
timeout 600 python tools/debug_waymo2.py 64 180 > $O/c64.log 2>&1; tail -22 $O/c64.log | cut -c1-220
timeout 600 python tools/debug_waymo2.py 64 180 nofuse > $O/c64_nofuse.log 2>&1; tail -18 $O/c64_nofuse.log | cut -c1-220
