#!/bin/bash
# round 3 session e: 8x64 halo conv (single accumulator, half-chunk ring): parity tests, A/B against the 4x64 kernel, benches
O=$PWD/gpurun_out/r03_e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_bench_shape_gpu.py -q -m gpu -k "conv or split or halo or head or dense" > $O/pytest_conv.log 2>&1; echo "conv tests rc=$?"; tail -6 $O/pytest_conv.log | cut -c1-300
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config']['execution'][:16], {k: v for k, v in d['roofline_dense']['dense_launches_ms'].items() if not k.startswith('linear')}, d.get('configs3_strong'))
PY
}
FF3D_CONV_HALO8=0 timeout 600 python bench.py --no-cpu-baseline --no-strong-probe > $O/bench_b32_halo4.json 2> $O/bench_b32_halo4.err; show $O/bench_b32_halo4.json
timeout 600 python bench.py --no-cpu-baseline --no-strong-probe > $O/bench_b32_halo8.json 2> $O/bench_b32_halo8.err; show $O/bench_b32_halo8.json
FF3D_CONV_HALO8=0 timeout 300 python bench.py --batch 4 --steps 30 --warmup 5 --no-cpu-baseline --no-strong-probe > $O/bench_b4_halo4.json 2> $O/bench_b4_halo4.err; show $O/bench_b4_halo4.json
timeout 300 python bench.py --batch 4 --steps 30 --warmup 5 --no-cpu-baseline --no-strong-probe > $O/bench_b4_halo8.json 2> $O/bench_b4_halo8.err; show $O/bench_b4_halo8.json
FF3D_BENCH_FORCE_DIST=1 timeout 300 python bench.py --batch 4 --steps 40 --warmup 5 --no-cpu-baseline --no-strong-probe > $O/bench_b4_rccl1_eager.json 2> $O/bench_b4_rccl1_eager.err; echo "rccl1 eager rc=$?"; show $O/bench_b4_rccl1_eager.json
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"; show $O/bench_default.json
