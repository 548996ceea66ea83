#!/bin/bash
# round 6 session k: MFMA local attention wired into the 'bevfusion' neck block: goldens, lc chain, lc bench A/B, kernel table of the lc step
O=$PWD/gpurun_out/r06_k; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_head_gpu.py tests/test_baseline_configs_gpu.py tests/test_round5_gpu.py tests/test_round6_gpu.py -x -q -k "encoder or neck or lc or chain or local_attention or i2p" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -6 $O/tests.log | cut -c1-300
b() { name=$1; shift; timeout 500 python bench.py --workload lc --steps 10 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b mfma_a; FF3D_LOCATT_MFMA=0 b scalar_a; b mfma_b; FF3D_LOCATT_MFMA=0 b scalar_b
python - <<'PY'
import json
for n in ('mfma_a', 'scalar_a', 'mfma_b', 'scalar_b'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r06_k/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'))
    except Exception as e:
        print(n, 'no line', e)
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_lc -o r -- python $R/bench.py --workload lc --graph off --steps 5 --warmup 3 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions > $O/bench_under_rocprof_lc.json 2> $O/rocprof_lc.err )
DB=$(find $O/prof_lc -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/bench_lc_kernel_stats_last_step.txt 2>&1
rm -rf $O/prof_lc
head -30 $O/bench_lc_kernel_stats_last_step.txt | cut -c1-170
