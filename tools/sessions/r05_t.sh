#!/bin/bash
# round 5 session t: fused feed-forward kernel v2 (one accumulator, weights two steps ahead): parity, kernel time, bench A/B
O=$PWD/gpurun_out/r05_t; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "ffn_rows" > $O/tests_ffn.log 2>&1; echo "rc=$?" >> $O/tests_ffn.log
tail -25 $O/tests_ffn.log | cut -c1-220
for mt in auto 4 3; do
  if [ $mt = auto ]; then timeout 120 python tools/bench_ffn_rows.py 19200 1024; else FF3D_FFN_MT=$mt timeout 120 python tools/bench_ffn_rows.py 19200 1024; fi
done 2>&1 | grep -v amdgpu.ids | tee $O/ffn_rows_v2.txt
timeout 120 python tools/bench_ffn_rows.py 8000 1024 2>&1 | grep -v amdgpu.ids | tee -a $O/ffn_rows_v2.txt
timeout 1200 python -m pytest tests/test_bench_shape_gpu.py tests/test_round5_gpu.py tests/test_head_gpu.py -x -q -k "full_size or pipelined or bench_shape or golden" > $O/tests_head.log 2>&1; echo "rc=$?" >> $O/tests_head.log
tail -5 $O/tests_head.log | cut -c1-220
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b fused
FF3D_FFN_FUSED=0 b two_launch
b fused2
FF3D_FFN_FUSED=0 b two_launch2
python - <<'PY'
import json
for n in ('fused', 'two_launch', 'fused2', 'two_launch2'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_t/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), d['config'].get('single_stream_eager', {}).get('value'))
    except Exception as e:
        print(n, 'no line', e)
PY
