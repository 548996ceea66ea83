#!/bin/bash
# round 6, third session: FF3D_LIN_MIN_ROWS = 0 as the default - the new no-vendor test, the small-batch tests, the 1-frame eager kernel table, the default line's latency record
O=$PWD/gpurun_out/r06_mr1; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_small_batch_gpu.py -x -q -m gpu -k "vendor or small or side_stream" 2>&1 | tail -5 | tee $O/tests.txt
prof() { name=$1; shift; ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o r -- python $R/bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions "$@" > $O/bench_under_rocprof_$name.json 2> $O/rocprof_$name.err ); DB=$(find $O/prof_$name -name '*_results.db' | head -1); python tools/rocprof_last_step.py $DB 80 > $O/bench_${name}_kernel_stats_last_step.txt 2>&1; rm -rf $O/prof_$name; head -8 $O/bench_${name}_kernel_stats_last_step.txt | cut -c1-150; }
prof b1_eager --graph off --batch 1 --steps 5 --warmup 3
grep -c Cijk $O/bench_b1_eager_kernel_stats_last_step.txt
timeout 300 python bench.py --latency-b1 --steps 5 --warmup 2 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(json.dumps(d.get('latency_b1_ms'))[:1500])" | tee $O/latency.txt
