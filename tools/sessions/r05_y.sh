#!/bin/bash
# round 5 session y: BEV flatten walking 8 frames per block with the positional tiles in registers (FF3D_FLATTEN_FB=1: one block per frame)
O=$PWD/gpurun_out/r05_y; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_head_gpu.py tests/test_bench_shape_gpu.py tests/test_baseline_configs_gpu.py -x -q -k "flatten or golden or full_size or bench_shape or config4 or waymo or bf16" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log | cut -c1-200
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b l_fb8
FF3D_FLATTEN_FB=1 b l_fb1
FF3D_FLATTEN_FB=32 b l_fb32
b waymo_fb8 --workload waymo --steps 10
FF3D_FLATTEN_FB=1 b waymo_fb1 --workload waymo --steps 10
b l_fb8b
FF3D_FLATTEN_FB=1 b l_fb1b
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_b32 -o r -- python $R/bench.py --graph off --steps 5 --warmup 3 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_under_rocprof_b32.json 2> $O/rocprof_b32.err )
DB=$(find $O/prof_b32 -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/bench_b32_eager_kernel_stats_last_step.txt 2>&1
rm -rf $O/prof_b32
grep -n "flatten\|last step" $O/bench_b32_eager_kernel_stats_last_step.txt | cut -c1-170
python - <<'PY'
import json
for n in ('l_fb8', 'l_fb1', 'l_fb32', 'l_fb8b', 'l_fb1b', 'waymo_fb8', 'waymo_fb1'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_y/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'))
    except Exception as e:
        print(n, 'no line', e)
PY
