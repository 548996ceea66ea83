#!/bin/bash
# round 3 session s: state check of the tree at its end - smoke(), whole GPU suite, default bench
O=$PWD/gpurun_out/r03_s; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 1700 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -3 $O/pytest_all.log | cut -c1-300
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-200 $O/bench_default.json
