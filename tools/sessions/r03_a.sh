#!/bin/bash
# round 3 session a: root-cause probe of the suite-order gradient deviation (tools/debug_suite_order.py) + baseline bench
O=$PWD/gpurun_out/r03_a; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/debug_suite_order.py $O/fresh > $O/fresh.log 2>&1; echo "fresh rc=$?"; grep -E "^\[|^      |flags" $O/fresh.log | cut -c1-400
timeout 900 python tools/debug_suite_order.py $O/suite tests/test_bench_shape_gpu.py tests/test_head_gpu.py tests/test_ops_gpu.py > $O/suite.log 2>&1; echo "suite rc=$?"
grep -E "^\[|^      |flags|passed|failed|no deviation" $O/suite.log | cut -c1-400
timeout 600 python bench.py --no-cpu-baseline > $O/bench_b32.json 2> $O/bench_b32.err; cut -c1-160 $O/bench_b32.json
