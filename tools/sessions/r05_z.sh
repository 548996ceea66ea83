#!/bin/bash
# round 5 session z: roi_mlp.0 with swapped operands on the 256 x 128 instance (FF3D_GEMM_SWAP=0: 128 x 128 tiles); flatten frames per block 4 / 16
O=$PWD/gpurun_out/r05_z; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_head_gpu.py tests/test_bench_shape_gpu.py -x -q -k "gemm or roi or golden or full_size or bench_shape" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log | cut -c1-200
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b swap
FF3D_GEMM_SWAP=0 b noswap
b swap2
FF3D_GEMM_SWAP=0 b noswap2
FF3D_FLATTEN_FB=4 b fb4
FF3D_FLATTEN_FB=16 b fb16
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_b32 -o r -- python $R/bench.py --graph off --steps 5 --warmup 3 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_under_rocprof_b32.json 2> $O/rocprof_b32.err )
DB=$(find $O/prof_b32 -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/bench_b32_eager_kernel_stats_last_step.txt 2>&1
rm -rf $O/prof_b32
grep -n "splitmm_kernel\|splitk\|last step" $O/bench_b32_eager_kernel_stats_last_step.txt | cut -c1-170
python - <<'PY'
import json
for n in ('swap', 'noswap', 'swap2', 'noswap2', 'fb4', 'fb16'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_z/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), {k: v for k, v in d['roofline_dense']['dense_launches_ms'].items() if '37632' in k})
    except Exception as e:
        print(n, 'no line', e)
PY
