#!/bin/bash
# round 2 final verification: the driver's three steps on a fresh box - GPU tests, smoke, default bench
O=$PWD/gpurun_out/r02_final; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_all.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench.json
