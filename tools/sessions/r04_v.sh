#!/bin/bash
# round 4 session v: I2P projections on the own linear kernel - parity tests, lc A/B
O=$PWD/gpurun_out/r04_v; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_head_gpu.py tests/test_baseline_configs_gpu.py tests/test_ops_gpu.py tests/test_training_gpu.py -x -q -m gpu -k "i2p or lc_chain or neck or encoder or cam" > $O/pytest_i2p.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest_i2p.log | cut -c1-400
show() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]).read().strip().splitlines():
    if line.startswith('{'):
        d = json.loads(line)
        print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'])
PY
}
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/$name.json 2> $O/$name.err; echo "rc=$?"; show $O/$name.json; }
b bench_lc_own --workload lc
FF3D_I2P_OWN_LINEAR=0 b bench_lc_vendor_i2p --workload lc
b bench_lc_own_2 --workload lc
