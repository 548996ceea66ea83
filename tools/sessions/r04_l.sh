#!/bin/bash
# round 4 session l: collective-capture preflight in disposable children - positive path (1-rank RCCL), negative path (2 ranks, gloo
# backend on one GPU -> eager), bench CLI tests, default line
O=$PWD/gpurun_out/r04_l; mkdir -p $O
export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]).read().strip().splitlines():
    if line.startswith('{'):
        d = json.loads(line)
        print(sys.argv[1].split('/')[-1], d['value'], d['n_gpus'], d['ms_per_step'], d['config']['execution'][:80], d['config']['ranks']['rccl_world'], [(r['rank'], r['pid'], r['ms_per_step']) for r in d['config']['ranks']['ranks']])
PY
}
FF3D_BENCH_FORCE_DIST=1 timeout 400 python bench.py --batch 4 --steps 40 --warmup 5 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_b4_rccl1.json 2> $O/bench_b4_rccl1.err; echo "rccl1 rc=$?"; show $O/bench_b4_rccl1.json; grep -v amdgpu.ids $O/bench_b4_rccl1.err | grep -i "bench.py" | head -3
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 1 --batch 4 --steps 10 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; echo "torchrun 1 rank rc=$?"; show $O/bench_torchrun1.json
FF3D_BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --batch 2 --channels 64 --steps 6 --warmup 2 --no-cpu-baseline --no-strong-probe > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo 2 ranks rc=$?"; show $O/bench_gloo2.json; grep -v amdgpu.ids $O/bench_gloo2.err | grep -i "bench.py:\|Error" | head -5
timeout 900 python -m pytest tests/test_bench_cli_gpu.py -q -m gpu > $O/pytest_cli.log 2>&1; echo "cli tests rc=$?"; tail -2 $O/pytest_cli.log | cut -c1-300
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_l/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['steps'], d['config']['execution'][:60], 'configs3', d['configs3_strong'].get('value'), d['configs3_strong'].get('execution', '')[:80], d['configs3_strong'].get('projected_speedup_8_vs_1'), {k:(v.get('value'),v.get('error')) for k,v in d['other_workloads'].items()})
PY
