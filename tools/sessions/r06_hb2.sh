#!/bin/bash
# round 6: the heatmap_box branch at full size (tests incl. the 180 x 180 head case) + its cost on the 32-frame step
O=$PWD/gpurun_out/r06_hb2; mkdir -p $O
timeout 900 python -m pytest tests/test_heatbox_gpu.py -q -m gpu 2>&1 | tail -8 > $O/heatbox.log; cat $O/heatbox.log
timeout 600 python tools/bench_heatbox.py > $O/bench_heatbox.json 2> $O/bench_heatbox.err; cat $O/bench_heatbox.json; tail -3 $O/bench_heatbox.err
