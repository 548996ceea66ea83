#!/bin/bash
# round 4 session k: two ranks on ONE GPU with the gloo backend (RCCL refuses duplicate devices): the multi-rank control flow of
# bench.py - per-slot process groups, capture attempt with a collective that cannot be captured -> eager fallback, host barrier,
# per-rank records, max-over-ranks timing
O=$PWD/gpurun_out/r04_k; mkdir -p $O
export TMPDIR=/tmp
FF3D_BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --batch 2 --channels 64 --steps 6 --warmup 2 --no-cpu-baseline --no-strong-probe > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo 2 ranks rc=$?"; grep "^{" $O/bench_gloo2.json | cut -c1-300; grep -v amdgpu.ids $O/bench_gloo2.err | grep -i "bench.py\|error\|Traceback" | head -10
FF3D_BENCH_BACKEND=gloo FF3D_BENCH_DIST_MODE=eager timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --batch 2 --channels 64 --steps 6 --warmup 2 --no-cpu-baseline --no-strong-probe > $O/bench_gloo2_eager.json 2> $O/bench_gloo2_eager.err; echo "gloo 2 ranks eager rc=$?"; grep "^{" $O/bench_gloo2_eager.json | cut -c1-200
python - <<'PY'
import json
for n in ('bench_gloo2', 'bench_gloo2_eager'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r04_k/{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['n_gpus'], d['config']['execution'][:50], d['config']['ranks']['rccl_world'], d['config']['ranks']['backend'], [(r['rank'], r['pid'], r['ms_per_step']) for r in d['config']['ranks']['ranks']])
    except Exception as e:
        print(n, 'no line', e)
PY
