#!/bin/bash
# round 3 session b: the new full-size BASELINE config tests (configs[2], configs[4]) with their error statistics,
# bench workloads lc / waymo, then the whole GPU suite
O=$PWD/gpurun_out/r03_b; mkdir -p $O
export TMPDIR=/tmp
FF3D_PARITY_STATS=$O/stats timeout 1200 python -m pytest tests/test_baseline_configs_gpu.py -q -m gpu --durations=12 > $O/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -30 $O/pytest_new.log | cut -c1-300
for f in $O/stats/*.json; do echo "== $f"; cat $f | tr -d '\n' | cut -c1-1500; echo; done
timeout 600 python bench.py --workload waymo --no-cpu-baseline > $O/bench_waymo.json 2> $O/bench_waymo.err; echo "waymo rc=$?"; cut -c1-200 $O/bench_waymo.json; tail -3 $O/bench_waymo.err
timeout 600 python bench.py --workload lc --no-cpu-baseline > $O/bench_lc.json 2> $O/bench_lc.err; echo "lc rc=$?"; cut -c1-200 $O/bench_lc.json; tail -3 $O/bench_lc.err
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_baseline_configs_gpu.py > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -5 $O/pytest_all.log | cut -c1-300
