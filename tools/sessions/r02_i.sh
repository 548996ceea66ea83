#!/bin/bash
O=gpurun_out/r02_i; mkdir -p $O
timeout 600 python tools/debug_waymo.py > $O/waymo.log 2>&1; tail -5 $O/waymo.log | cut -c1-300
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "batch32 or any_magnitude or nhwc_pair or merge_aug or waymo" > $O/pytest.log 2>&1; tail -8 $O/pytest.log | cut -c1-200
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_b32.json 2>$O/bench_b32.err; cut -c1-200 $O/bench_b32.json
