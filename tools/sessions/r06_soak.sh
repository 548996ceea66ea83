#!/bin/bash
# round 6, third session: [replay, eager launch, synchronise] soak on the final tree (32-frame pipelined head x 1000, lc neck + head x 300, one-frame graph x 2000)
O=$PWD/gpurun_out/r06_soak; mkdir -p $O
timeout 1200 python tools/stress_replay_sync.py pipelined --iters 1000 > $O/stress_pipelined.txt 2>&1; tail -2 $O/stress_pipelined.txt
timeout 900 python tools/stress_replay_sync.py lc --iters 300 --batch 8 > $O/stress_lc.txt 2>&1; tail -1 $O/stress_lc.txt
timeout 900 python tools/stress_replay_sync.py pipelined --iters 2000 --batch 1 > $O/stress_pipelined_b1.txt 2>&1; tail -1 $O/stress_pipelined_b1.txt
