#!/bin/bash
# round 6 session h: value GEMM 256-column form with two accumulators (shipping form) + its ablations; RoI sampler with LDS-staged cells
O=$PWD/gpurun_out/r06_h; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_ops_gpu.py -x -q -k "gemm or roi" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -5 $O/tests.log | cut -c1-250
E=$PWD/focalformer3d_amd/lib/libff3d_hip_exp.so
for v in 1 0 1 0; do FF3D_GEMM_WS1=$v timeout 200 python tools/experiments/exp_valuegemm.py 2>&1 | grep -v amdgpu.ids | sed "s/^/WS1=$v /" >> $O/microbench.txt; done
for a in 1 2 4 8 6 7 9 10 12; do FF3D_LIB=$E FF3D_WS_ABLATE=$a timeout 200 python tools/experiments/exp_valuegemm.py 2>&1 | grep -v amdgpu.ids | sed "s/^/WS1=1 ABLATE=$a /" >> $O/microbench.txt; done
cat $O/microbench.txt
for sm in 0 1; do for v in 1 0 1 0; do SMALL=$sm FF3D_ROI_LDS=$v timeout 200 python tools/experiments/exp_roi.py 2>&1 | grep -v amdgpu.ids >> $O/roi.txt; done; done
cat $O/roi.txt
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b new_a; FF3D_ROI_LDS=0 b roi0_a; FF3D_GEMM_WS1=0 b ws0_a; b new_b; FF3D_ROI_LDS=0 b roi0_b; FF3D_GEMM_WS1=0 b ws0_b
python - <<'PY'
import json
for n in ('new_a', 'roi0_a', 'ws0_a', 'new_b', 'roi0_b', 'ws0_b'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r06_h/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), {k: v for k, v in d['roofline_dense']['dense_launches_ms'].items() if 'gemm 1360800' in k})
    except Exception as e:
        print(n, 'no line', e)
PY
