#!/bin/bash
# round 3 session k: whole suite after the dropout-flag fix, host profile of the B=4 eager step, default bench
O=$PWD/gpurun_out/r03_k; mkdir -p $O
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -4 $O/pytest_all.log | cut -c1-300
timeout 300 python tools/profile_host.py 4 30 > $O/host_profile_b4.txt 2>&1; tail -3 $O/host_profile_b4.txt
timeout 300 python tools/profile_host.py 1 30 > $O/host_profile_b1.txt 2>&1; tail -3 $O/host_profile_b1.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-600 $O/bench_default.json
