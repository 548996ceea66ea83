#!/bin/bash
# round 5 session j: pairs handed from producer to consumer inside the bevfusion neck: parity (neck goldens, lc chain, graph unit) + lc bench A/B
O=$PWD/gpurun_out/r05_j; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_head_gpu.py tests/test_baseline_configs_gpu.py tests/test_round5_gpu.py tests/test_ops_gpu.py -x -q -k "neck or config2 or lc_chain or neck_and_head or local_context or lss" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -8 $O/tests.log
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b lc --workload lc --steps 12
FF3D_NECK_PAIR_CHAIN=0 b lc_chain_off --workload lc --steps 12
b lc2 --workload lc --steps 12
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_lc -o r -- python $R/bench.py --graph off --workload lc --steps 4 --warmup 2 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_under_rocprof_lc.json 2> $O/rocprof_lc.err )
DB=$(find $O/prof_lc -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/bench_lc_kernel_stats_last_step.txt 2>&1
rm -rf $O/prof_lc
head -16 $O/bench_lc_kernel_stats_last_step.txt | cut -c1-170
python - <<'PY'
import json
for n in ('lc', 'lc_chain_off', 'lc2'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_j/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), d['config'].get('single_stream_eager', {}).get('value'))
    except Exception as e:
        print(n, 'no line', e)
PY
tail -3 $O/bench_lc.err
