#!/bin/bash
# round 5 session aa: RoI sampler + roi_mlp.0 per chunk of frames (matrix read back from the memory-side cache?) - A/B
O=$PWD/gpurun_out/r05_aa; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bench_shape_gpu.py -x -q -k "full_size" > $O/tests_default.log 2>&1; echo "rc=$?" >> $O/tests_default.log
FF3D_ROI_CHUNK=2 FF3D_GEMM_SWAP_MINM=1024 timeout 600 python -m pytest tests/test_bench_shape_gpu.py -x -q -k "full_size" > $O/tests_chunk.log 2>&1; echo "rc=$?" >> $O/tests_chunk.log
tail -3 $O/tests_default.log $O/tests_chunk.log | cut -c1-200
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b whole
FF3D_ROI_CHUNK=2 FF3D_GEMM_SWAP_MINM=1024 b chunk2
FF3D_ROI_CHUNK=1 FF3D_GEMM_SWAP_MINM=512 b chunk1
FF3D_ROI_CHUNK=4 FF3D_GEMM_SWAP_MINM=1024 b chunk4
FF3D_ROI_CHUNK=8 b chunk8
b whole2
python - <<'PY'
import json
for n in ('whole', 'chunk2', 'chunk1', 'chunk4', 'chunk8', 'whole2'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_aa/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), {k: v for k, v in d['roofline_dense']['dense_launches_ms'].items() if '37632' in k})
    except Exception as e:
        print(n, 'no line', e)
PY
