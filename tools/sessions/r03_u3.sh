#!/bin/bash
# round 3 session u3 (the last tree of the round, after the grouped launches): final check of the tree (after the periodic ws kernel / opt-in switches): whole GPU suite + default bench
O=$PWD/gpurun_out/r03_u3; mkdir -p $O
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -3 $O/pytest_all.log | cut -c1-300
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-200 $O/bench_default.json
